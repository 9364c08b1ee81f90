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
