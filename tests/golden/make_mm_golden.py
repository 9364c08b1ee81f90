"""Generates the mapper-level golden fixtures from the *compiled reference* (oracle/_ref/minialign + libmm_ref.so):
for each seeded synthetic set (inputs are regenerated on the fly by tools/gensim.c, so only the parameters are stored)
the reference's SAM (gzip) and a few per-stage vectors (minimizers, sorted seeds, chain roots of the first reads).
Run in the build container:  python tests/golden/make_mm_golden.py"""
import gzip, hashlib, json, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mmlib as M

SETS = [
    dict(name='pb_small', preset='pacbio', genome=(101, 200000, 3, 0.08), reads=(102, 0.6, 'pacbio', 'fa', 3000, 1000)),
    dict(name='pb_rep', preset='pacbio', genome=(111, 120000, 2, 0.45), reads=(112, 0.5, 'pacbio', 'fa', 2500, 800)),
    dict(name='ont_small', preset='ont.1dsq', genome=(121, 200000, 4, 0.08), reads=(122, 0.5, 'ont', 'fa')),
]

def make_inputs(s, d):
    ref = os.path.join(d, s['name'] + '.ref.fa'); rd = os.path.join(d, s['name'] + '.reads.fa')
    M.gensim('genome', *s['genome'], out=ref)
    M.gensim('reads', s['reads'][0], ref, *s['reads'][1:], out=rd)
    return ref, rd

def main():
    here = os.path.dirname(os.path.abspath(__file__))
    meta = {'generator': 'tests/golden/make_mm_golden.py', 'reference': 'ocxtal/minialign 0.6.0-devel (gcc 11.4 -O3 -mavx2 build, oracle/_ref)', 'sets': []}
    with tempfile.TemporaryDirectory() as d:
        for s in SETS:
            ref, rd = make_inputs(s, d)
            sam = subprocess.run([os.path.join(M.ROOT, 'oracle', '_ref', 'minialign'), '-x' + s['preset'], ref, rd],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
            sam = b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@PG'))
            with gzip.GzipFile(os.path.join(here, s['name'] + '.sam.gz'), 'wb', mtime=0) as f: f.write(sam)
            r = M.RefMM(['-x' + s['preset']], ref)
            reads = M.read_fasta(rd)
            stages = []
            for name, q in reads[:6]:
                sk = r.sketch(q); sd = r.seed(q); ch = r.chain()
                stages.append({'read': name, 'sketch_n': int(len(sk)), 'sketch_md5': hashlib.md5(sk.tobytes()).hexdigest(), 'sketch_head': [int(x) for x in sk[:8]],
                               'seed_n': int(len(sd)), 'seed_md5': hashlib.md5(sd.tobytes()).hexdigest(),
                               'chain': [int(x) for x in ch]})
            meta['sets'].append(dict(s, occ=r.occ(), n_records=sam.count(b'\n'), sam_md5=hashlib.md5(sam).hexdigest(), stages=stages,
                                     ref_md5=hashlib.md5(open(ref, 'rb').read()).hexdigest(), reads_md5=hashlib.md5(open(rd, 'rb').read()).hexdigest()))
    json.dump(meta, open(os.path.join(here, 'mm_golden.json'), 'w'), indent=1)
    print('wrote', [s['name'] for s in SETS])

if __name__ == '__main__':
    main()
