"""Reader parity (bseq_read_fasta, minialign.c:1996-2090): oddly formatted FASTA / FASTQ files given as *reads* over a tiny reference -- almost all of them
stay unmapped, so the SAM shows what the reader made of each record (name, bases, qualities, comment).  Expected SAM from the *compiled reference*
(oracle/_ref/minialign -t1, under a time limit).  Run in the build container:  python tests/golden/make_parse_golden.py"""
import gzip, os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import mmlib as M

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {
    'fa_wrapped':  b'>r1 wrapped at 10\nACGTACGTAC\nGGGTTTAAAC\nCC\n>r2\nAC\nGT\n',
    'fa_crlf':     b'>r1 crlf\r\nACGTACGTAC\r\nGGGT\r\n>r2\r\nTTTT\r\n',
    'fa_lower':    b'>lower\nacgtnacgtuACGTU\n>iupac\nRYKMSWBDHVN-.*\n',
    'fa_blank':    b'\n\n>r1\nACGT\n\nACGT\n\n>r2\n\nGG\n',
    'fa_noeol':    b'>r1\nACGTACGT\n>last\nGGGCCC',
    'fa_spaces':   b'>  padded name  and comment \nACGT\n>\tlead_tab\tx y\nACGTT\n>name_only_trailing_space \nAC\n',
    'fa_empty':    b'>empty1\n>empty2\n\n>full\nACGT\n>empty3\n',
    'fa_gt':       b'>r1\nACGT>ACGT\nAC\n>r2\nGG\n',
    'fq_plain':    b'@q1 c1\nACGTACGT\n+\nIIIIHHHH\n@q2\nGGCC\n+q2\n!!!!\n',
    'fq_at_qual':  b'@q1\nACGTACGT\n+\n@@@@IIII\n@q2\nGGCC\n+\n@+@+\n',
    'fq_wrapped':  b'@q1\nACGT\nACGT\n+\nIIII\nHHHH\n@q2\nGG\n+\nII\n',
    'fq_crlf':     b'@q1 x\r\nACGTAC\r\n+\r\nIIIIII\r\n@q2\r\nGG\r\n+\r\nII\r\n',
    'fq_noeol':    b'@q1\nACGTAC\n+\nIIIIII\n@q2\nGGA\n+\nIII',
    'fq_short_q':  b'@q1\nACGTAC\n+\nIII\n@q2\nGGA\n+\nIII\n',
}
OPTS = ['-xpacbio', '-Q', '-TCO']

def make_parse_inputs(d):
    ref = os.path.join(d, 'parse.ref.fa')
    M.gensim('genome', 981, 30000, 1, 0.0, out=ref)
    out = {}
    for name, blob in CASES.items():
        p = os.path.join(d, 'parse.%s.%s' % (name, 'fq' if name.startswith('fq') else 'fa')); open(p, 'wb').write(blob); out[name] = p
    return ref, out

def strip_pg(sam):
    return b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@PG'))

def main():
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as d:
        ref, files = make_parse_inputs(d)
        for name, p in files.items():
            try:
                r = subprocess.run([os.path.join(M.ROOT, 'oracle', '_ref', 'minialign')] + OPTS + ['-t1', ref, p], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=20)
                res[name] = (r.returncode, strip_pg(r.stdout))
            except subprocess.TimeoutExpired:
                res[name] = ('timeout', b'')
            print(name, res[name][0], [l.split(b'\t')[0].decode() + ':' + l.split(b'\t')[9].decode()[:24] + ':' + l.split(b'\t')[10].decode()[:12] + ':' + b' '.join(l.split(b'\t')[11:]).decode() for l in res[name][1].splitlines() if not l.startswith(b'@')])
    import json
    with gzip.GzipFile(os.path.join(HERE, 'parse_cases.json.gz'), 'wb', mtime=0) as f:
        f.write(json.dumps({k: [v[0], v[1].decode('latin1')] for k, v in res.items()}).encode())

if __name__ == '__main__':
    main()
