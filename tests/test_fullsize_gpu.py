"""GPU, full size (BASELINE.json configs[1]: E.coli-size reference, x100 read set, 464 Mb): size-independent properties of the
SAM the HIP pipeline writes, plus byte parity against the CPU oracle on a sample of the same reads."""
import os, re, subprocess, tempfile
import numpy as np, pytest
import mmlib as M

pytestmark = pytest.mark.gpu
CLI = os.path.join(M.ROOT, 'minialign_amd', 'minialign')
CIG = re.compile(rb'(\d+)([MIDNSHP=X])')

def test_full_size_properties_and_sample_parity():
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'reads.fa'); out = os.path.join(d, 'out.sam')
        M.gensim('genome', 0x5eed0001, 4641652, 1, 0.05, out=ref)                 # the bench.py workload
        M.gensim('reads', 0x5eed0002, ref, 100.0, 'pacbio', 'fa', 20000, 2000, out=rd)
        with open(out, 'wb') as f:
            r = subprocess.run([CLI, '-xpacbio', ref, rd], stdout=f, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        # read names / lengths in input order
        names, lens = [], []
        with open(rd, 'rb') as f:
            for line in f:
                if line.startswith(b'>'): names.append(line[1:].split()[0]); lens.append(0)
                else: lens[-1] += len(line) - 1
        qlen = dict(zip(names, lens))
        contig = {}
        order, n_primary, n_mapped, bases_mapped = [], 0, 0, 0
        sample = set(names[:100]) | set(names[i] for i in np.random.default_rng(7).choice(len(names), 100, replace=False))
        kept = {n: [] for n in sample}
        with open(out, 'rb') as f:
            for line in f:
                if line.startswith(b'@'):
                    if line.startswith(b'@SQ'):
                        sn = re.search(rb'SN:(\S+)', line).group(1); contig[sn] = int(re.search(rb'LN:(\d+)', line).group(1))
                    continue
                c = line.split(b'\t', 10)
                name, flag, rname, pos, cigar, seq = c[0], int(c[1]), c[2], int(c[3]), c[5], c[9]
                if name in kept: kept[name].append(line)
                if not (flag & (256 | 2048)):
                    n_primary += 1; order.append(name)
                if flag & 4:
                    assert cigar == b'*' and rname == b'*'
                    continue
                ops = [(int(n), o) for n, o in CIG.findall(cigar)]
                assert b''.join(b'%d%s' % (n, o) for n, o in ops) == cigar
                q_all = sum(n for n, o in ops if o in b'MIS=XH'); q_seq = sum(n for n, o in ops if o in b'MIS=X'); r_span = sum(n for n, o in ops if o in b'MDN=X')
                assert q_all == qlen[name], (name, q_all, qlen[name])               # clips included, every base of the read accounted for
                assert q_seq == len(seq)
                assert 1 <= pos and pos - 1 + r_span <= contig[rname]
                if not (flag & (256 | 2048)): n_mapped += 1; bases_mapped += qlen[name]
        assert order == names                                                        # one primary line per read, input order
        assert n_primary == len(names)
        assert n_mapped > 0.98 * len(names)                                          # simulated reads of this reference: practically all map
        # sample parity: the same reads through the CPU oracle (single-contig reference: no state is carried between reads)
        sub = os.path.join(d, 'sample.fa'); want_names = [n for n in names if n in sample]
        with open(rd, 'rb') as f, open(sub, 'wb') as g:
            keep = False
            for line in f:
                if line.startswith(b'>'): keep = line[1:].split()[0] in sample
                if keep: g.write(line)
        o = subprocess.run([os.path.join(M.ROOT, 'oracle', 'ora_minialign'), '-xpacbio', ref, sub], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        want = {}
        for line in o.splitlines(True):
            if not line.startswith(b'@'): want.setdefault(line.split(b'\t', 1)[0], []).append(line)
        for n in want_names:
            assert kept[n] == want[n], 'read %r differs from the oracle' % n
        # and, where the compiled reference travelled with the snapshot, the whole 680 MB of SAM against it (-t1: with several
        # threads the reference's own output depends on which thread buffer a read lands in)
        refbin = os.path.join(M.ROOT, 'oracle', '_ref', 'minialign')
        if os.path.exists(refbin):
            import hashlib
            def digest(cmd):
                h = hashlib.md5(); p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                for line in p.stdout:
                    if not line.startswith(b'@PG'): h.update(line)
                assert p.wait() == 0
                return h.hexdigest()
            h_ref = digest([refbin, '-xpacbio', '-t1', ref, rd])
            h_got = hashlib.md5()
            with open(out, 'rb') as f:
                for line in f:
                    if not line.startswith(b'@PG'): h_got.update(line)
            assert h_got.hexdigest() == h_ref, 'full-size SAM differs from the compiled reference'


@pytest.mark.parametrize('gseed,rseed,glen,contigs,rep,depth,preset,prof', [
    (1, 2, 4641652, 1, 0.05, 100.0, 'pacbio', 'pacbio'),        # a read of this set keeps more than 64 candidate seeds in mm_search_load_next (radix passes of its sort)
    (11, 12, 12000000, 16, 0.20, 20.0, 'pacbio', 'pacbio'),     # repeat-rich, several contigs
    (21, 22, 4641652, 1, 0.05, 40.0, 'ont.1dsq', 'ont'),
    (114, 115, 2000000, 6, 0.25, 25.0, 'ava', 'pacbio'),        # linear gaps, many weak hits per read; one traceback leaves the band
])
def test_more_full_size_sets_against_the_compiled_reference(gseed, rseed, glen, contigs, rep, depth, preset, prof):
    """whole SAM (md5) of further full-size read sets against oracle/_ref/minialign -t1; skipped where the compiled reference did not travel"""
    import hashlib
    refbin = os.path.join(M.ROOT, 'oracle', '_ref', 'minialign')
    if not os.path.exists(refbin): pytest.skip('oracle/_ref not built')
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); rd = os.path.join(d, 'rd.fa')
        M.gensim('genome', gseed, glen, contigs, rep, out=ref)
        M.gensim('reads', rseed, ref, depth, prof, 'fa', *((15000, 4000) if preset == 'ava' else (20000, 2000)), out=rd)
        def digest(cmd):
            h = hashlib.md5(); n = 0; p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            for line in p.stdout:
                if not line.startswith(b'@PG'): h.update(line); n += 1
            assert p.wait() == 0, cmd
            return h.hexdigest(), n
        got = digest([os.path.join(M.ROOT, 'minialign_amd', 'minialign'), '-x' + preset, ref, rd])
        want = digest([refbin, '-x' + preset, '-t1', ref, rd])
        assert got == want
