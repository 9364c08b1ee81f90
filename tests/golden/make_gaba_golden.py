"""Generates tests/golden/gaba_extend.json from the *compiled reference* (oracle/_ref/libgaba_ref.so, built from
/root/reference by oracle/Makefile).  Inputs come from this repo's own seeded generator (tests/gabalib.py);
only inputs and the reference's outputs are stored -- no reference source.  Run in the build container:
    python tests/golden/make_gaba_golden.py"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import gabalib as G

def main():
    out = {'generator': 'tests/golden/make_gaba_golden.py', 'reference': 'ocxtal/minialign 0.6.0-devel libgaba (AVX2 build)', 'groups': []}
    for name, P, seed in (('pacbio', G.PACBIO, 9001), ('ont1dsq', G.ONT1DSQ, 9002), ('affine', G.AFFINE_DEFAULT, 9003),
                          ('linear', G.LINEAR_AVA, 9004)):         # gi == 0: the reference's linear-gap build (the `ava' preset)
        ref = G.Reference(**P)
        jobs = G.random_jobs(seed, 24, max_len=700)
        grp = {'name': name, 'params': dict(P), 'jobs': []}
        for (a, apos, arev, b, bpos, brev, bw, tr) in jobs:
            exp = ref.extend(a, apos, arev, b, bpos, brev, bw, tr)
            grp['jobs'].append({'a': a.tolist(), 'apos': apos, 'arev': bool(arev), 'b': b.tolist(), 'bpos': bpos, 'brev': bool(brev),
                                'bw': bw, 'expect': exp})
        out['groups'].append(grp)
    p = os.path.join(os.path.dirname(__file__), 'gaba_extend.json')
    json.dump(out, open(p, 'w'), separators=(',', ':'))
    print('wrote', p, os.path.getsize(p), 'bytes')

if __name__ == '__main__':
    main()
