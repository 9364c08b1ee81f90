"""GPU parity: the HIP banded extension (gaba_dp_extend_batch, include/gaba.h) against the CPU oracle
(oracle/ora_gaba.c) on seeded random jobs -- bit-exact on every observable: fills (max/status/positions),
max position, path bits, segments, gap counts, identity bits."""
import numpy as np, pytest
import gabalib as G

pytestmark = pytest.mark.gpu

def _diff(x, y):
    return [k for k in y if x.get(k) != y[k]]

@pytest.mark.parametrize("name,P", [("pacbio", G.PACBIO), ("ont1dsq", G.ONT1DSQ), ("affine", G.AFFINE_DEFAULT)])
def test_extend_batch_matches_oracle(name, P):
    hip = G.Hip(**P); ora = G.Oracle(**P)
    jobs = G.random_jobs(1234, 300)
    got = hip.extend_batch(jobs)
    bad = []
    for i, j in enumerate(jobs):
        want = ora.extend(*j)
        if got[i] != want:
            bad.append((i, _diff(got[i], want)))
    assert not bad, "mismatching jobs (index, fields): %r" % bad[:5]

def test_long_reads_match_oracle():
    P = G.PACBIO
    hip = G.Hip(**P); ora = G.Oracle(**P)
    jobs = G.random_jobs(77, 24, max_len=30000, bw_choices=(0,))
    got = hip.extend_batch(jobs)
    for i, j in enumerate(jobs):
        want = ora.extend(*j)
        assert got[i] == want, (i, _diff(got[i], want))

def test_golden_fixture():
    """committed golden vectors generated from the compiled reference (tests/golden/make_gaba_golden.py)"""
    import json, os
    path = os.path.join(os.path.dirname(__file__), 'golden', 'gaba_extend.json')
    gold = json.load(open(path))
    for grp in gold['groups']:
        hip = G.Hip(**grp['params'])
        jobs = [(np.array(j['a'], dtype=np.uint8), j['apos'], j['arev'], np.array(j['b'], dtype=np.uint8), j['bpos'], j['brev'], j['bw'], 1)
                for j in grp['jobs']]
        got = hip.extend_batch(jobs)
        for i, j in enumerate(grp['jobs']):
            want = j['expect']; want['fills'] = [tuple(f) for f in want['fills']]; want['pos'] = tuple(want['pos'])
            if 'segs' in want: want['segs'] = [tuple(s) for s in want['segs']]
            assert got[i] == want, (grp['name'], i, _diff(got[i], want))
