"""Command-line options beyond the presets (-k -w -B -f -L -a -b -e -p -q -r -Y -s -m -W -G, minialign.c:5990-6099): one seeded, repeat-rich input,
several option lines, the expected SAM from the *compiled reference* (oracle/_ref/minialign -t1).
Run in the build container:  python tests/golden/make_opt_golden.py"""
import gzip, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import mmlib as M

HERE = os.path.dirname(os.path.abspath(__file__))
OPTION_LINES = [
    ('kw',      ['-k13', '-w8']),                                                      # no preset: the bare defaults with another sketch
    ('affine',  ['-xpacbio', '-a1', '-b2', '-p2', '-q1', '-r0']),                     # -r0: plain affine gaps
    ('index',   ['-xpacbio', '-f0.1,0.02,0.002', '-B12', '-L3000']),                  # thresholds, bucket bits, length filter (both sides)
    ('frq5',    ['-xpacbio', '-f', '0.2,0.1,0.05,0.01,0.001']),                       # five rescue rounds, space-separated argument
    ('chain',   ['-xont.1dsq', '-s200', '-m0.5', '-W3000', '-G2000', '-Y30']),
    ('mod',     ['-xpacbio', '-eAC1,GT-1']),                                          # asymmetric score matrix
    ('spaced',  ['-k', '14', '-w', '5', '-a2', '-b3', '-p3', '-q1', '-r2,2']),
]

def make_opt_inputs(d):
    ref = os.path.join(d, 'opt.ref.fa'); rd = os.path.join(d, 'opt.reads.fa')
    M.gensim('genome', 951, 400000, 5, 0.30, out=ref)
    M.gensim('reads', 952, ref, 0.6, 'pacbio', 'fa', 3000, 1200, out=rd)
    return ref, rd

def strip_pg(sam):
    return b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@PG'))

def main():
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        ref, rd = make_opt_inputs(d)
        for name, opts in OPTION_LINES:
            sam = strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', '_ref', 'minialign')] + opts + ['-t1', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
            with gzip.GzipFile(os.path.join(HERE, 'opt_%s.sam.gz' % name, ), 'wb', mtime=0) as f: f.write(sam)
            print(name, sam.count(b'\n'), 'lines', sum(1 for l in sam.splitlines() if not l.startswith(b'@') and l.split(b'\t')[2] != b'*'), 'mapped')

if __name__ == '__main__':
    main()
