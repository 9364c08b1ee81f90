"""GPU: the mm_* C entry points of include/minialign.h driven through ctypes (what bench.py and a host program bind), checked against
the command-line program and the oracle: index queries, the in-memory batch entry, and two batches in flight on two lanes."""
import ctypes, os, subprocess, tempfile
import numpy as np, pytest
import mmlib as M
from golden.make_mm_golden import make_inputs

pytestmark = pytest.mark.gpu
CLI = os.path.join(M.ROOT, 'minialign_amd', 'minialign')

def _lib():
    L = ctypes.CDLL(os.path.join(M.ROOT, 'minialign_amd', 'libminialign_amd.so'))
    for f in ('mm_opt_init', 'mm_idx_gen', 'mm_align_init', 'mm_reads_load', 'mm_batch_upload', 'mm_batch_upload_lane'): getattr(L, f).restype = ctypes.c_void_p
    return L

def _open(L, preset, ref, rd):
    o = ctypes.c_void_p(L.mm_opt_init())
    argv = (ctypes.c_char_p * 4)(b'minialign', ('-x' + preset).encode(), ref.encode(), rd.encode())
    files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
    assert L.mm_opt_parse(o, 4, argv, files, 8, ctypes.byref(nf)) == 0 and nf.value == 2
    mi = ctypes.c_void_p(L.mm_idx_gen(o, ref.encode())); assert mi
    al = ctypes.c_void_p(L.mm_align_init(o, mi)); assert al, 'mm_align_init failed (no GPU?)'
    return o, mi, al

def _body(sam):
    return b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@'))

def test_index_queries_and_batch_entries_match_cli_and_oracle():
    s = dict(name='g_api', preset='pacbio', genome=(361, 300000, 4, 0.15), reads=(362, 1.0, 'pacbio', 'fa', 4000, 1500))
    with tempfile.TemporaryDirectory() as d:
        ref, rd = make_inputs(s, d)
        want = _body(subprocess.run([CLI, '-xpacbio', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
        L = _lib(); o, mi, al = _open(L, 'pacbio', ref, rd)
        # index: occurrence thresholds and value lists against the oracle's index over the same reference
        refseq = M.read_fasta(ref); ora = M.OracleMM('pacbio', refseq)
        assert L.mm_idx_n_seq(mi) == len(refseq)
        assert [L.mm_idx_occ(mi, i) for i in range(3)] == ora.occ()[:3]
        reads = M.read_fasta(rd)
        buf = (ctypes.c_uint64 * 4096)()
        for m in ora.sketch(reads[0][1])[:200]:
            key = int(m) >> 8
            n = L.mm_idx_get(mi, ctypes.c_uint64(key), buf, 4096)
            assert [int(buf[i]) for i in range(n)] == [int(v) for v in ora.idx_get(key)]
        # in-memory batch entry: one byte per base, concatenated
        lens = (ctypes.c_uint32 * len(reads))(*[len(q) for _, q in reads])
        cat = np.concatenate([q for _, q in reads]).astype(np.uint8)
        names = (ctypes.c_char_p * len(reads))(*[n.encode() for n, _ in reads])
        sam = ctypes.c_char_p(); slen = ctypes.c_uint64(0)
        assert L.mm_align_batch(al, cat.ctypes.data_as(ctypes.c_void_p), lens, names, len(reads), ctypes.byref(sam), ctypes.byref(slen)) == 0
        assert ctypes.string_at(sam, slen.value) == want
        # the same reads as two batches in flight on two lanes (single-contig order does not matter here: 4 contigs, so run them in order
        # and hand the carried state over through finish order, as mm_align_file would)
        L.mm_align_destroy(al); al = ctypes.c_void_p(L.mm_align_init(o, mi))
        rs = ctypes.c_void_p(L.mm_reads_load(rd.encode())); n = L.mm_reads_count(rs); h = n // 2
        b0 = ctypes.c_void_p(L.mm_batch_upload_lane(al, rs, 0, h, 0)); assert b0
        sam2 = ctypes.c_char_p(); slen2 = ctypes.c_uint64(0)
        assert L.mm_batch_run(al, b0) == 0 and L.mm_batch_finish(al, b0, ctypes.byref(sam2), ctypes.byref(slen2)) == 0
        b1 = ctypes.c_void_p(L.mm_batch_upload(al, rs, h, n - h)); assert b1
        assert L.mm_batch_run_async(al, b1) == 0 and L.mm_batch_wait(al, b1) == 0
        assert L.mm_batch_finish(al, b1, ctypes.byref(sam2), ctypes.byref(slen2)) == 0
        assert ctypes.string_at(sam2, slen2.value) == want
        # two different batches truly in flight together (single-contig reference: nothing is carried between reads)
        s1 = dict(name='g_api1', preset='pacbio', genome=(371, 400000, 1, 0.10), reads=(372, 1.0, 'pacbio', 'fa', 3000, 1000))
        ref1, rd1 = make_inputs(s1, d)
        want1 = _body(subprocess.run([CLI, '-xpacbio', ref1, rd1], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout)
        o1, mi1, al1 = _open(L, 'pacbio', ref1, rd1)
        rs1 = ctypes.c_void_p(L.mm_reads_load(rd1.encode())); n1 = L.mm_reads_count(rs1); h1 = n1 // 2
        c0 = ctypes.c_void_p(L.mm_batch_upload_lane(al1, rs1, 0, h1, 0)); c1 = ctypes.c_void_p(L.mm_batch_upload_lane(al1, rs1, h1, n1 - h1, 1)); assert c0 and c1
        assert L.mm_batch_run_async(al1, c0) == 0 and L.mm_batch_run_async(al1, c1) == 0
        assert L.mm_batch_wait(al1, c0) == 0 and L.mm_batch_wait(al1, c1) == 0
        sam3 = ctypes.c_char_p(); slen3 = ctypes.c_uint64(0)
        assert L.mm_batch_finish(al1, c0, ctypes.byref(sam3), ctypes.byref(slen3)) == 0 and L.mm_batch_finish(al1, c1, ctypes.byref(sam3), ctypes.byref(slen3)) == 0
        assert ctypes.string_at(sam3, slen3.value) == want1
        for b in (b0, b1, c0, c1): L.mm_batch_free(b)
        L.mm_reads_free(rs); L.mm_reads_free(rs1)
        L.mm_align_destroy(al); L.mm_align_destroy(al1); L.mm_idx_destroy(mi); L.mm_idx_destroy(mi1)


def test_structured_results_match_the_sam_records():
    """mm_align_batch_regs: what mm_align_seq returns per read (mm_reg_t / mm_aln_t / gaba_alignment_t).  Flag, position, mapping quality and CIGAR rebuilt from
    the structures must equal fields 2-6 of the SAM records of the command-line program"""
    import gabalib as G
    s = dict(name='g_regs', preset='pacbio', genome=(381, 300000, 4, 0.25), reads=(382, 1.0, 'pacbio', 'fa', 4000, 1500))
    with tempfile.TemporaryDirectory() as d:
        ref, rd = make_inputs(s, d)
        sam = subprocess.run([CLI, '-xpacbio', ref, rd], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        want = {}
        for l in sam.splitlines():
            if not l.startswith(b'@'): f = l.split(b'\t'); want.setdefault(f[0].decode(), []).append((int(f[1]), f[2].decode(), int(f[3]), int(f[4]), f[5].decode()))
        L = _lib(); o, mi, al = _open(L, 'pacbio', ref, rd)
        L.gaba_dump_cigar_reverse.restype = ctypes.c_uint64
        refseq = M.read_fasta(ref); reads = M.read_fasta(rd)
        lens = (ctypes.c_uint32 * len(reads))(*[len(q) for _, q in reads]); cat = np.concatenate([q for _, q in reads]).astype(np.uint8)
        regs = (ctypes.c_void_p * len(reads))()
        assert L.mm_align_batch_regs(al, cat.ctypes.data_as(ctypes.c_void_p), lens, len(reads), regs) == 0
        class Aln(ctypes.Structure):          # mm_aln_t { aid, mapq } + gaba_alignment_t header
            _fields_ = [('aid', ctypes.c_uint32), ('mapq', ctypes.c_uint32), ('res', ctypes.c_void_p * 2), ('score', ctypes.c_int64), ('identity', ctypes.c_double),
                        ('agcnt', ctypes.c_uint32), ('bgcnt', ctypes.c_uint32), ('dcnt', ctypes.c_uint32), ('slen', ctypes.c_uint32), ('seg', ctypes.POINTER(G.Seg)),
                        ('plen', ctypes.c_uint32), ('padding', ctypes.c_uint32)]
        buf = ctypes.create_string_buffer(1 << 18); n_checked = 0
        for i, (name, q) in enumerate(reads):
            if not regs[i]:
                assert want[name][0][0] == 4; continue
            n_all, n_uniq = (ctypes.c_uint32 * 2).from_address(regs[i])
            tab = (ctypes.c_void_p * n_all).from_address(regs[i] + 8)
            got = []; flag = 0
            for k in range(n_all):
                if k >= n_uniq: flag = 0x100
                a = Aln.from_address(tab[k]); path = tab[k] + ctypes.sizeof(Aln)
                assert a.aid == k and a.padding == 0x40000000
                for j in range(a.slen, 0, -1):
                    sg = a.seg[j - 1]; rlen = len(refseq[sg.aid >> 1][1])
                    hl = len(q) - sg.bpos - sg.blen; tl = sg.bpos; clip = 'H' if flag & 0x900 else 'S'
                    L.gaba_dump_cigar_reverse(buf, ctypes.c_uint64(len(buf)), ctypes.c_void_p(path), ctypes.c_uint64(sg.ppos), ctypes.c_uint64(sg.alen + sg.blen))
                    cig = ('%d%s' % (hl, clip) if hl else '') + buf.value.decode() + ('%d%s' % (tl, clip) if tl else '')
                    got.append((flag | ((~sg.bid & 1) << 4), refseq[sg.aid >> 1][0], rlen - sg.apos - sg.alen + 1, a.mapq >> 4, cig))
                    if k == 0 and j == a.slen: flag = 0x800
                flag = 0x800
            assert got == want[name], name
            n_checked += len(got); L.mm_reg_free(ctypes.c_void_p(regs[i]))
        assert n_checked > len(reads)
        L.mm_align_destroy(al); L.mm_idx_destroy(mi)
