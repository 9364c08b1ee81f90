"""Edge-case read set for the mapper (empty / ragged inputs, N runs, unmappable reads, both strands, FASTQ):
inputs are built deterministically here, the expected SAM comes from the *compiled reference* (oracle/_ref/minialign).
Run in the build container:  python tests/golden/make_edge_golden.py"""
import gzip, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, mmlib as M

HERE = os.path.dirname(os.path.abspath(__file__))
COMP = bytes.maketrans(b'ACGTN', b'TGCAN')

def make_edge_inputs(d, fmt='fa'):
    """returns (ref.fa, reads.<fmt>); fmt 'fq' writes the same reads as FASTQ with dummy qualities"""
    ref = os.path.join(d, 'edge.ref.fa'); rd = os.path.join(d, 'edge.reads.' + fmt)
    M.gensim('genome', 901, 150000, 3, 0.10, out=ref)
    contigs = M.read_fasta(ref)
    rng = np.random.default_rng(4242)
    txt = lambda a: bytes(b'ACGTN'[int(c)] for c in a)
    c0 = txt(contigs[0][1]); c1 = txt(contigs[1][1]); c2 = txt(contigs[2][1])
    def noisy(s, e=0.08):
        out = bytearray()
        for ch in s:
            r = rng.random()
            if r < e * 0.3: out += bytes([b'ACGT'[int(rng.integers(0, 4))]])          # substitution
            elif r < e * 0.6: continue                                                # deletion
            elif r < e: out += bytes([ch, b'ACGT'[int(rng.integers(0, 4))]])          # insertion
            else: out.append(ch)
        return bytes(out)
    rc = lambda s: s.translate(COMP)[::-1]
    reads = [
        ('one_base', b'A'),
        ('shorter_than_k', c0[100:110]),
        ('exactly_k', c0[200:215]),
        ('short_exact', c0[1000:1060]),
        ('all_n', b'N' * 500),
        ('random_unmappable', bytes(b'ACGT'[int(x)] for x in rng.integers(0, 4, 3000))),
        ('fwd_clean', c0[5000:9000]),
        ('rev_clean', rc(c0[12000:16000])),
        ('fwd_noisy', noisy(c1[3000:11000])),
        ('rev_noisy', rc(noisy(c1[15000:22000]))),
        ('n_run_inside', c0[30000:32000] + b'N' * 150 + c0[32150:35000]),
        ('n_sprinkled', bytes((78 if rng.random() < 0.02 else ch) for ch in c2[2000:8000])),
        ('chimera_two_contigs', noisy(c0[40000:43000]) + noisy(c2[10000:13500])),
        ('big_deletion', c1[25000:28000] + c1[28400:31500]),
        ('big_insertion', c1[33000:35000] + bytes(b'ACGT'[int(x)] for x in rng.integers(0, 4, 300)) + c1[35000:37500]),
        ('contig_start', c2[0:2500]),
        ('contig_end', c2[len(c2) - 2500:]),
        ('overhang_past_end', c0[len(c0) - 1500:] + bytes(b'ACGT'[int(x)] for x in rng.integers(0, 4, 1200))),
        ('duplicate_a', c0[50000:53000]),
        ('duplicate_b', c0[50000:53000]),
        ('long_read', noisy(c0[60000:120000], 0.12)),
        ('lowercase_is_same', c0[20000:23000].lower()),
        ('tandem', (c1[40000:40120] * 30)),
        ('last_one_base', b'C'),
    ]
    with open(rd, 'wb') as f:
        for name, s in reads:
            if fmt == 'fq': f.write(b'@' + name.encode() + b'\n' + s + b'\n+\n' + b'I' * len(s) + b'\n')
            else: f.write(b'>' + name.encode() + b'\n' + s + b'\n')
    return ref, rd

def strip_pg(sam):
    return b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@PG'))

def main():
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        for fmt in ('fa', 'fq'):
            ref, rd = make_edge_inputs(d, fmt)
            for preset in ('pacbio', 'ont.1dsq'):
                sam = strip_pg(subprocess.run([os.path.join(M.ROOT, 'oracle', '_ref', 'minialign'), '-x' + preset, ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
                with gzip.GzipFile(os.path.join(HERE, 'edge_%s_%s.sam.gz' % (preset.replace('.', ''), fmt)), 'wb', mtime=0) as f: f.write(sam)
                print(fmt, preset, sam.count(b'\n'), 'lines')

if __name__ == '__main__':
    main()
