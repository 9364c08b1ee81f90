"""Circular references (-c, minialign.c:2795-2799, 3632-3696, 3753): three contigs (a 60 kb and a 9 kb circle, one linear), reads that cross the origins on both
strands at many offsets, reads longer than the small circle, plain reads; the expected SAM comes from the *compiled reference* (oracle/_ref/minialign -t1).
Run in the build container:  python tests/golden/make_circ_golden.py"""
import gzip, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mmlib as M

HERE = os.path.dirname(os.path.abspath(__file__))
CIRC_LINES = [
    ('all',   ['-xpacbio', '-c*']),                   # every sequence circular
    ('named', ['-xpacbio', '-cchrA,plasmid']),        # by name: the linear contig keeps its ends
    ('ont',   ['-xont.1dsq', '-c', 'plasmid', '-TSA']),   # separate argument word; SA tag over the two segments of a wrapped alignment
    ('none',  ['-xpacbio']),                          # the same reads without -c
]

def make_circ_inputs(d):
    ref = os.path.join(d, 'circ.ref.fa'); rd = os.path.join(d, 'circ.reads.fa')
    rng = np.random.default_rng(971)
    gen = lambda n: bytes(b'ACGT'[int(c)] for c in rng.integers(0, 4, n))
    contigs = [('chrA', gen(60000)), ('plasmid', gen(9000)), ('lin', gen(40000))]
    with open(ref, 'wb') as f:
        for n, s in contigs: f.write(b'>' + n.encode() + b'\n' + s + b'\n')
    def noisy(t, e):
        out = bytearray()
        for ch in t:
            r = rng.random()
            if r < e * 0.2: out.append(b'ACGT'[int(rng.integers(0, 4))])
            elif r < e * 0.5: continue
            elif r < e: out += bytes([ch, b'ACGT'[int(rng.integers(0, 4))]])
            else: out.append(ch)
        return bytes(out)
    rc = lambda t: t.translate(bytes.maketrans(b'ACGTN', b'TGCAN'))[::-1]
    reads = []
    for ci, (n, s) in enumerate(contigs):
        L = len(s); dbl = s + s
        for j in range(14):                                   # reads across the origin: start 100 .. 5000 before the end
            back = [100, 300, 700, 1200, 2000, 3000, 5000][j % 7]; ln = back + [150, 900, 2500, 4000][j % 4]
            t = noisy(dbl[L - back:L - back + ln], [0.03, 0.08, 0.12][j % 3])
            reads.append(('%s_wrap%d' % (n, j), rc(t) if j & 1 else t))
        for j in range(6):                                    # plain reads inside the contig
            st = int(rng.integers(0, L - 3000)); t = noisy(s[st:st + int(rng.integers(1500, 3000))], 0.1)
            reads.append(('%s_in%d' % (n, j), rc(t) if j & 1 else t))
    p = contigs[1][1]
    reads.append(('plasmid_twice', noisy((p + p + p)[4000:4000 + 15000], 0.05)))       # longer than the circle
    reads.append(('plasmid_exact', p[8000:] + p[:1000]))
    with open(rd, 'wb') as f:
        for n, s in reads: f.write(b'>' + n.encode() + b'\n' + s + b'\n')
    return ref, rd

def strip_pg(sam):
    return b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@PG'))

def main():
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        ref, rd = make_circ_inputs(d)
        for name, opts in CIRC_LINES:
            sam = strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', '_ref', 'minialign')] + opts + ['-t1', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
            with gzip.GzipFile(os.path.join(HERE, 'circ_%s.sam.gz' % name), 'wb', mtime=0) as f: f.write(sam)
            flags = {}
            for l in sam.splitlines():
                if not l.startswith(b'@'): flags[l.split(b'\t')[1]] = flags.get(l.split(b'\t')[1], 0) + 1
            print(name, sam.count(b'\n'), 'lines', flags)

if __name__ == '__main__':
    main()
