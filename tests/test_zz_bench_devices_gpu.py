"""GPU, last in the suite (written after GPU access had closed in round 4: it has not run yet, and a failure here must not hide the tests in front of it):
`python bench.py --gpus 2` launched PLAINLY -- no torch.distributed.run, one process -- drives two device contexts through the library's own multi-device engine
(mm_align_init spans the devices, stream_map deals the batches) and prints ONE JSON line with n_gpus = 2; on a one-GPU box MM_BENCH_SAME_DEVICE puts both contexts on cuda:0."""
import json, os, subprocess, sys
import pytest
import mmlib as M

pytestmark = pytest.mark.gpu

def test_bench_line_with_two_devices_in_one_process():
    env = dict(os.environ, MM_BENCH_SAME_DEVICE='1', MM_SLAB_GB='8', PYTHONPATH=M.ROOT)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'): env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(M.ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--workload', 'dm6', '--genome-len', '6000000', '--contigs', '40', '--depth', '8',
                        '--lanes', '2', '--check', '--check-reads', '300', '--baseline-reads', '600'], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=M.ROOT, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['devices_in_process'] == 2 and d['config']['processes'] == 1 and d['value'] > 0 and d['unit'] == 'Gbases/s'
    assert d['sam_identical'] is True, d.get('sam_check')
    assert d['roofline']['achieved'] > 0 and 'cpu_baseline' not in d

def test_deferred_rescue_rounds_give_the_same_bytes():
    """MM_K3_DEFER_RESCUE (experiment, off by default): reads the first occurrence threshold leaves without a result come back to the host and run their later rounds as
    launches of their own, their chains spread over the launch as chain jobs -- on a repeat-rich set with a high seed threshold, where many reads need those rounds"""
    import tempfile
    CLI = os.path.join(M.ROOT, 'minialign_amd', 'minialign')
    strip = lambda sam: b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@PG'))
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'rd.fa')
        M.gensim('genomehard', 7601, 12000000, 4, 0.5, out=ref); M.gensim('reads', 7602, ref, 0.5, 'pacbio', 'fa', 6000, 2500, out=rd)
        opts = ['-xpacbio', '-f0.2,0.05,0.002']
        want = strip(subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign')] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
        for thr in ('1', '64'):
            r = subprocess.run([CLI] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MM_SLAB_GB='32', MM_BATCH_BASES='3000000', MM_K3_DEFER_RESCUE=thr), timeout=600)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            assert strip(r.stdout) == want, thr
