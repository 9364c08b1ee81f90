"""The text dumpers of include/gaba.h (extended CIGAR with = / X, gapped sequence rows) and gaba_dp_calc_score against golden vectors produced by the
compiled reference's own functions (tests/golden/make_dumper_golden.py).  The dumpers are host code: no GPU needed; calc_score needs a gaba_dp_t."""
import ctypes, gzip, json, os
import numpy as np, pytest
import gabalib as G

HERE = os.path.dirname(os.path.abspath(__file__))
EOU = 0x800000000000

class Section(ctypes.Structure):
    _fields_ = [('id', ctypes.c_uint32), ('len', ctypes.c_uint32), ('base', ctypes.c_uint64)]

def _vectors():
    return json.loads(gzip.open(os.path.join(HERE, 'golden', 'gaba_dumpers.json.gz')).read())

def _setup(v):
    a = np.array([int(c) for c in v['a']], dtype=np.uint8); b = np.array([int(c) for c in v['b']], dtype=np.uint8)
    pa = np.full(len(a) + 128, 4, dtype=np.uint8); pa[64:64 + len(a)] = a
    pb = np.full(len(b) + 128, 4, dtype=np.uint8); pb[64:64 + len(b)] = b
    mir = lambda ptr, n: 2 * EOU - ptr - n                       # gaba_mirror (gaba.h:151-155)
    sa = Section(1 if v['arev'] else 0, len(a), mir(pa.ctypes.data + 64, len(a)) if v['arev'] else pa.ctypes.data + 64)
    sb = Section(3 if v['brev'] else 2, len(b), mir(pb.ctypes.data + 64, len(b)) if v['brev'] else pb.ctypes.data + 64)
    arr = (ctypes.c_uint32 * (len(v['path']) + 10))(); arr[0] = v['plen']; arr[1] = 0x40000000
    for i, w in enumerate(v['path']): arr[2 + i] = w
    return pa, pb, sa, sb, arr, ctypes.c_void_p(ctypes.addressof(arr) + 8), G.Seg(*v['seg'])

def test_xcigar_and_sequence_rows_match_the_reference():
    L = G.load_product()
    for f in ('gaba_dump_xcigar_forward', 'gaba_dump_xcigar_reverse', 'gaba_dump_seq_ref', 'gaba_dump_seq_query', 'gaba_dump_seq_reverse', 'gaba_print_xcigar_forward'):
        getattr(L, f).restype = ctypes.c_uint64
    PR = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char)
    vs = _vectors(); assert len(vs) > 100
    buf = ctypes.create_string_buffer(1 << 15)
    for v in vs:
        pa, pb, sa, sb, arr, path, seg = _setup(v)
        n = L.gaba_dump_xcigar_forward(buf, ctypes.c_uint64(len(buf)), path, ctypes.byref(seg), ctypes.byref(sa), ctypes.byref(sb))
        assert buf.value.decode() == v['xcigar_f'] and n == len(v['xcigar_f'])
        L.gaba_dump_xcigar_reverse(buf, ctypes.c_uint64(len(buf)), path, ctypes.byref(seg), ctypes.byref(sa), ctypes.byref(sb))
        assert buf.value.decode() == v['xcigar_r']
        L.gaba_dump_seq_ref(buf, ctypes.c_uint64(len(buf)), path, ctypes.byref(seg), ctypes.byref(sa)); assert buf.value.decode() == v['row_a']
        L.gaba_dump_seq_query(buf, ctypes.c_uint64(len(buf)), path, ctypes.byref(seg), ctypes.byref(sb)); assert buf.value.decode() == v['row_b']
        if not v['arev']:
            L.gaba_dump_seq_reverse(buf, ctypes.c_uint64(len(buf)), 0, path, ctypes.c_uint64(seg.ppos), ctypes.c_uint64(seg.alen + seg.blen), ctypes.c_void_p(pa.ctypes.data + 64 + seg.apos), ctypes.c_char(b'-'))
            assert buf.value.decode() == v['row_a_rev']
        got = []
        L.gaba_print_xcigar_forward(PR(lambda fp, n, c: (got.append('%d%s' % (n, c.decode())), 1)[1]), None, path, ctypes.byref(seg), ctypes.byref(sa), ctypes.byref(sb))
        assert ''.join(got) == v['xcigar_f']

@pytest.mark.gpu
def test_calc_score_matches_the_reference():
    L = G.load_product()
    L.gaba_dp_init.restype = ctypes.c_void_p; L.gaba_dp_calc_score.restype = ctypes.c_void_p
    class Score(ctypes.Structure):
        _fields_ = [('score', ctypes.c_int64), ('identity', ctypes.c_double)] + [(k, ctypes.c_uint32) for k in ('agcnt', 'bgcnt', 'mcnt', 'xcnt', 'aicnt', 'bicnt', 'afgcnt', 'bfgcnt', 'aficnt', 'bficnt')] + [('adj', ctypes.c_int32), ('reserved', ctypes.c_uint32)]
    models = dict(pacbio=G.PACBIO, ont1dsq=G.ONT1DSQ, affine=G.AFFINE_DEFAULT); dps = {}
    for v in _vectors():
        if v['model'] not in dps:
            p = G.Params(); m = models[v['model']]; p.sm = G.score_matrix(m['m'], m['x']); p.gi, p.ge, p.gfa, p.gfb, p.xdrop = m['gi'], m['ge'], m['gfa'], m['gfb'], m['xdrop']
            ctx = ctypes.c_void_p(L.gaba_init(ctypes.byref(p))); assert ctx
            dps[v['model']] = ctypes.c_void_p(L.gaba_dp_init(ctx)); assert dps[v['model']]
        pa, pb, sa, sb, arr, path, seg = _setup(v)
        r = L.gaba_dp_calc_score(dps[v['model']], path, ctypes.byref(seg), ctypes.byref(sa), ctypes.byref(sb)); assert r
        s = Score.from_address(r)
        got = [s.score, s.mcnt, s.xcnt, s.agcnt, s.bgcnt, s.aicnt, s.bicnt, s.afgcnt, s.bfgcnt, s.aficnt, s.bficnt, s.adj]
        assert got == v['score'], (v['model'], v['seg'])
        assert np.float64(s.identity).tobytes().hex() == v['identity']
        L.gaba_dp_flush(dps[v['model']])

def test_the_device_side_cigar_walker_prints_what_the_reverse_dumper_prints():
    """K4 (csrc/mm_cigar.hpp) restates gaba_dp_print_cigar_reverse for one lane per segment; the same code compiled for the host (mm_cigar_walk) against the library's
    dumper (gaba_dump_cigar_reverse, the function the SAM goldens and the vectors above pin) on the golden paths and on random bit strings -- runs of every length,
    stretches that start and end anywhere in a word, paths at odd and even word offsets of a pool (the dumper aligns its pointer down to 8 bytes, the walker counts bits
    from the start of the pool)"""
    L = G.load_product()
    L.mm_cigar_walk.restype = ctypes.c_uint64; L.gaba_dump_cigar_reverse.restype = ctypes.c_uint64
    rng = np.random.default_rng(20260929)
    cases = []
    for v in _vectors()[:60]:
        cases.append((np.array(v['path'], dtype=np.uint32), v['plen'], 0, v['plen']))
    for _ in range(300):
        # alignment-like bit strings: 01 pairs (a match: the 1 in the lower bit) with runs of 0s (deletions) and 1s (insertions) of many lengths in between -- never a
        # deletion right above an insertion, which IS a match pair; stretches are cut right above a run of pairs (a lone 0 above a 1 at the bottom of a stretch is not a path)
        ops = []; n_bits = 0; target = int(rng.integers(40, 4000)); last = 'M'
        while n_bits < target:
            r = rng.random()
            if r < 0.6 or (last == 'I'): k = int(rng.integers(1, 90)); ops.append(('M', k)); n_bits += 2 * k; last = 'M'
            elif r < 0.8: k = int(rng.integers(1, 140)); ops.append(('D', k)); n_bits += k; last = 'D'
            else: k = int(rng.integers(1, 140)); ops.append(('I', k)); n_bits += k; last = 'I'
        bits = []; cuts = [0]
        for op, k in ops:
            if op == 'M': bits += [1, 0] * k; cuts.append(len(bits))
            elif op == 'D': bits += [0] * k
            else: bits += [1] * k
        words = np.zeros(len(bits) // 32 + 2, dtype=np.uint32)
        for i, b in enumerate(bits):
            if b: words[i >> 5] |= np.uint32(1 << (i & 31))
        lo = int(cuts[int(rng.integers(0, len(cuts)))]) if rng.random() < 0.7 else 0
        if lo >= len(bits): lo = 0
        ln = int(rng.integers(1, len(bits) - lo + 1))
        cases.append((words, len(bits), lo, ln))
    buf = ctypes.create_string_buffer(1 << 16); out = ctypes.create_string_buffer(1 << 16)
    for words, plen, ppos, ln in cases:
        for lead in (2, 3):          # the path starts at an even / odd word of the pool (header words right in front of it, filler in front of those)
            pool = np.zeros(lead + len(words) + 8, dtype=np.uint32)
            pool[:lead - 2] = 0xdeadbeef; pool[lead - 2] = plen & 0xffffffff; pool[lead - 1] = 0x40000000; pool[lead:lead + len(words)] = words
            path = ctypes.c_void_p(pool.ctypes.data + 4 * lead)
            n = L.gaba_dump_cigar_reverse(buf, ctypes.c_uint64(len(buf)), path, ctypes.c_uint64(ppos), ctypes.c_uint64(ln))
            want = buf.raw[:n]
            cnt = L.mm_cigar_walk(pool.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(lead), ctypes.c_uint64(ppos), ctypes.c_uint64(ln), None)
            m = L.mm_cigar_walk(pool.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(lead), ctypes.c_uint64(ppos), ctypes.c_uint64(ln), out)
            assert cnt == m == n and out.raw[:m] == want, (plen, ppos, ln, lead, want[:80], out.raw[:80])
