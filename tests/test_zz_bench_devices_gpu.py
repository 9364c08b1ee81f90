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

def test_extension_trials_that_start_at_the_end_of_a_section():
    """libgaba level: trials that start 1 .. 47 bases in front of the end of the b section (a seed at the very end of a read: the init fetch cannot finish inside the section,
    the fill goes on into the N tail) with OTHER sequences -- all-A padding, random bases -- standing next to it in the arena: positions, fills and (traced) paths as the oracle's,
    whatever the neighbours hold.  (Written while chasing the one ONT-like read of round 4 whose records differed; the DP turned out to be innocent -- DESIGN.md 5 -- and this sweep
    is why that could be said.)"""
    import numpy as np
    import gabalib as G
    rng = np.random.default_rng(11)
    a = rng.integers(0, 4, 6000).astype(np.uint8)
    rd = a[1500:4500].copy(); flip = rng.random(len(rd)) < 0.1; rd[flip] = (rd[flip] + 1 + rng.integers(0, 3, int(flip.sum()))) % 4          # a read off a[1500:4500], 10 % substitutions
    rc = G.revcomp(rd)
    for P in (G.ONT1DSQ, G.PACBIO):
        hip = G.Hip(**P); ora = G.Oracle(**P)
        for filler in (np.zeros(64, np.uint8), rng.integers(0, 4, 64).astype(np.uint8)):
            jobs = []; keep = []
            for k in range(1, 48):
                for da in (0, 1, 3, 15):
                    for tr in (0, 1):
                        # forward read section ending at len(rd): the trial starts k bases before its end; the same through the reversed section of the reverse complement
                        jobs.append((a, 100, 0, filler, 0, 0, 0, 0)); keep.append(False)
                        jobs.append((a, 1500 + len(rd) - k + da, 0, rd, len(rd) - k, 0, 0, tr)); keep.append(True)
                        jobs.append((a, 100, 0, filler, 0, 0, 0, 0)); keep.append(False)
                        jobs.append((a, 1500 + len(rd) - k + da, 0, rc, len(rd) - k, 1, 0, tr)); keep.append(True)
            got = hip.extend_batch(jobs)
            bad = [(j[2:3] + j[4:8]) for j, g, kp in zip(jobs, got, keep) if kp and g != ora.extend(*j)]
            assert not bad, bad[:5]
