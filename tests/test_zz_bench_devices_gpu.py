"""GPU, last in the suite.  Written after GPU access had closed in round 4 and opt-in until they had run; all four passed on a box at the start of round 5
(profiles/round5_unrun_tests.txt) and have been part of the default run since.
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

def test_rescue_rounds_on_a_hard_repeat_set_give_the_oracles_bytes():
    """a reference with mammalian repeat structure and a high seed threshold: many reads find their chains only in the later occurrence-threshold rounds, which run inside the
    extension launch on the wave that holds the read, the chains they find spread over the launch as jobs (default) or walked by that wave (MM_K3_NO_ROUND_JOBS)"""
    import tempfile
    CLI = os.path.join(M.ROOT, 'minialign_amd', 'minialign')
    strip = lambda sam: b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@PG'))
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'rd.fa')
        M.gensim('genomehard', 7601, 12000000, 4, 0.5, out=ref); M.gensim('reads', 7602, ref, 0.5, 'pacbio', 'fa', 6000, 2500, out=rd)
        opts = ['-xpacbio', '-f0.2,0.05,0.002']
        want = strip(subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign')] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
        for thr in ({}, dict(MM_K3_NO_ROUND_JOBS='1')):
            r = subprocess.run([CLI] + opts + [ref, rd], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MM_SLAB_GB='32', MM_BATCH_BASES='3000000', **thr), timeout=600)
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

def test_every_read_runs_with_the_value_the_reference_would_carry():
    """the carried reference length read by read: what the device handed every read (MM_DUMP_CARRY, after verification and re-runs) against what the oracle's
    thread buffer holds when it takes the read (OM_DEBUG_CARRY) -- on a many-contig, repeat-rich set with a high seed threshold, where reads go on to the later occurrence
    thresholds and are chained again (the case in which round 3's bookkeeping went wrong: a read that is chained again changes the prediction behind its neighbours' backs)"""
    import tempfile
    CLI = os.path.join(M.ROOT, 'minialign_amd', 'minialign')
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'rd.fa'); dump = os.path.join(d, 'carry.txt')
        M.gensim('genomehard', 7701, 16000000, 60, 0.5, out=ref); M.gensim('reads', 7702, ref, 0.4, 'pacbio', 'fa', 5000, 2000, out=rd)
        opts = ['-xpacbio', '-f0.2,0.05,0.002']
        o = subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign')] + opts + [ref, rd], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, OM_DEBUG_CARRY='1'), check=True)
        want = [(f[1], int(f[2])) for f in (l.split('\t') for l in o.stderr.decode().splitlines()) if f[0] == 'carry']
        lens = [int(l.split(b'len=')[1]) for l in open(ref, 'rb') if l.startswith(b'>')]
        NIL = 0xffffffff
        for env in (dict(MM_BATCH_BASES='400000', MM_LANES='3'), dict(MM_BATCH_BASES='3000000', MM_LANES='1')):
            r = subprocess.run([CLI] + opts + [ref, rd], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, MM_SLAB_GB='8', MM_DUMP_CARRY=dump, **env), timeout=600)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            got = [l.split('\t') for l in open(dump).read().splitlines()]          # name, value the read ran with, its first apos, (bpos >= qlen), last reference it loaded, results
            assert [g[0] for g in got] == [n for n, _ in want]
            bad = []; flips = 0
            for i, (g, (n, truth)) in enumerate(zip(got, want)):
                used, apos0, cond0, rid_last = int(g[1]), int(g[2]), int(g[3]), int(g[4])
                # the decision the value feeds (minialign.c:3823: apos >= rlen) must be the one the reference's value gives -- the value itself may differ where it decides nothing
                if apos0 != NIL and not cond0 and (apos0 >= used) != (apos0 >= truth): bad.append((i, n, used, truth, apos0))
                if apos0 != NIL and not cond0 and apos0 >= truth: flips += 1
                # ... and what the read leaves behind is what the reference carries on to the next one
                if i + 1 < len(want): assert want[i + 1][1] == (lens[rid_last] if rid_last >= 0 else truth), (i, n, rid_last, truth, want[i + 1])
            assert not bad, bad[:5]
            assert flips > 0, 'no read of the set has its first seed beyond the carried length: the test has no teeth'
