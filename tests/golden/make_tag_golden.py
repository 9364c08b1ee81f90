"""Optional SAM fields (-T tags, -R read group, -Q qualities, -P; minialign.c:5204-5426, 5880-5967): a FASTQ read set with header comments, qualities and
a few chimeric reads over a repeat-rich reference, several option lines, the expected SAM from the *compiled reference* (oracle/_ref/minialign -t1).
Run in the build container:  python tests/golden/make_tag_golden.py"""
import gzip, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import mmlib as M

HERE = os.path.dirname(os.path.abspath(__file__))
RG = '@RG\\tID:g1\\tSM:x'
TAG_LINES = [
    ('all',   ['-xpacbio', '-TAS,NM,XS,MD,NH,IH,CO,RG', '-R', RG, '-Q']),
    ('sa',    ['-xpacbio', '-TSA,MD,NM']),                        # SA swallows the supplementary / secondary records
    ('p',     ['-xpacbio', '-P']),                                # QUIRK: also prints IH (flag and tag bits share a word)
    ('ih',    ['-xont.1dsq', '-TIH,AS', '-Q']),                   # QUIRK: also omits the secondary records
    ('rg',    ['-xpacbio', '-TSA', '-R' + RG]),                   # -R alone switches RG:Z on
    ('paf',   ['-xpacbio', '-Opaf']),                             # the other output formats (-O): no header, nothing for unmapped reads
    ('paftag', ['-xpacbio', '-O', 'paf', '-TAS,ID,NM,CG,SQ', '-P']),
    ('blast6', ['-xpacbio', '-Oblast6']),
    ('maf',   ['-xont.1dsq', '-Omaf']),
    ('edge_maf', ['-xpacbio', '-Omaf']),
    ('ava',   ['-xava', '-Opaf']),                                # the `ava' preset: linear gaps (gi = 0)
    ('avasam', ['-xava', '-A']),                                  # QUIRK: -A shares its bit with the AS tag
    ('avax',  ['-xava', '-X', '-C', '3,4', '-Opaf', '-TAS,NM']),  # -X: every file onto every file (here: reference and reads alike), one index per file
    ('edge_sa', ['-xpacbio', '-TSA,NM,MD,XS', '-Q']),             # on the edge-case reads of make_edge_golden.py (FASTQ): chimeras, N runs, both strands
]

def inputs_for(name, d):
    if name.startswith('edge_'):
        sys.path.insert(0, HERE)
        from make_edge_golden import make_edge_inputs
        return make_edge_inputs(d, 'fq')
    return make_tag_inputs(d)


def make_tag_inputs(d):
    ref = os.path.join(d, 'tag.ref.fa'); fa = os.path.join(d, 'tag.reads.fa'); fq = os.path.join(d, 'tag.reads.fq')
    M.gensim('genome', 961, 300000, 4, 0.10, out=ref)
    M.gensim('reads', 962, ref, 0.5, 'pacbio', 'fa', 3000, 1200, out=fa)
    recs = [(l.split(b'\n', 1)[0], l.split(b'\n', 1)[1].replace(b'\n', b'')) for l in open(fa, 'rb').read().split(b'>')[1:]]
    import numpy as np
    rng = np.random.default_rng(963); contigs = [bytes(b'ACGTN'[int(c)] for c in q) for _, q in M.read_fasta(ref)]
    def noisy(t, e=0.06):
        out = bytearray()
        for ch in t:
            r = rng.random()
            if r < e * 0.3: out.append(b'ACGT'[int(rng.integers(0, 4))])
            elif r < e * 0.6: continue
            elif r < e: out += bytes([ch, b'ACGT'[int(rng.integers(0, 4))]])
            else: out.append(ch)
        return bytes(out)
    rc = lambda t: t.translate(bytes.maketrans(b'ACGTN', b'TGCAN'))[::-1]
    for i in range(4):                                                                          # two / three unrelated pieces: primary + supplementary records
        a = contigs[i % 4][20000 + 7000 * i:23000 + 7000 * i]; b = contigs[(i + 1) % 4][41000 + 3000 * i:44500 + 3000 * i]; c = contigs[(i + 2) % 4][5000:7000]
        recs.append((b'chimera%d' % i, noisy(a) + (rc(noisy(b)) if i & 1 else noisy(b)) + (noisy(c) if i >= 2 else b'')))
    recs += [(b'short', b'ACGTACGTAC'), (b'allN', b'N' * 500)]
    with open(fq, 'wb') as f:
        for i, (name, seq) in enumerate(recs):
            cmt = [b'', b' a comment', b'\tkey=val\twith tabs  ', b' '][i % 4]
            qual = bytes(33 + ((j * 7 + i) % 41) for j in range(len(seq)))
            f.write(b'@' + name + cmt + b'\n' + seq + b'\n+\n' + qual + b'\n')
    return ref, fq

def strip_pg(sam):
    return b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@PG'))

def main():
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for name, opts in TAG_LINES:
            ref, fq = inputs_for(name, d)
            sam = strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', '_ref', 'minialign')] + opts + ['-t1', ref, fq], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
            with gzip.GzipFile(os.path.join(HERE, 'tag_%s.sam.gz' % name), 'wb', mtime=0) as f: f.write(sam)
            flags = {}
            for l in sam.splitlines():
                if '-O' not in ' '.join(opts) and not l.startswith(b'@'): flags[l.split(b'\t')[1]] = flags.get(l.split(b'\t')[1], 0) + 1
            print(name, sam.count(b'\n'), 'lines', flags, 'SA' if b'SA:Z' in sam else '')

if __name__ == '__main__':
    main()
