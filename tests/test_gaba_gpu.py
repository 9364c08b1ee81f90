"""GPU parity: the HIP banded extension (gaba_dp_extend_batch, include/gaba.h) against the CPU oracle
(oracle/ora_gaba.c) on seeded random jobs -- bit-exact on every observable: fills (max/status/positions),
max position, path bits, segments, gap counts, identity bits."""
import ctypes
import numpy as np, pytest
import gabalib as G

pytestmark = pytest.mark.gpu

def _diff(x, y):
    return [k for k in y if x.get(k) != y[k]]

@pytest.mark.parametrize("name,P", [("pacbio", G.PACBIO), ("ont1dsq", G.ONT1DSQ), ("affine", G.AFFINE_DEFAULT), ("linear", G.LINEAR_AVA)])
def test_extend_batch_matches_oracle(name, P):
    hip = G.Hip(**P); ora = G.Oracle(**P)
    jobs = G.random_jobs(1234, 300) + (G.random_jobs(4321, 300, max_len=12000) if name == "linear" else [])       # gi = 0: ties everywhere, the traceback's tie-breaks matter
    got = hip.extend_batch(jobs)
    bad = []
    for i, j in enumerate(jobs):
        want = ora.extend(*j)
        if got[i] != want:
            bad.append((i, _diff(got[i], want)))
    assert not bad, "mismatching jobs (index, fields): %r" % bad[:5]

def test_long_reads_match_oracle():
    P = G.PACBIO
    hip = G.Hip(**P); ora = G.Oracle(**P)
    jobs = G.random_jobs(77, 24, max_len=30000, bw_choices=(0,))
    got = hip.extend_batch(jobs)
    for i, j in enumerate(jobs):
        want = ora.extend(*j)
        assert got[i] == want, (i, _diff(got[i], want))

def test_golden_fixture():
    """committed golden vectors generated from the compiled reference (tests/golden/make_gaba_golden.py)"""
    import json, os
    path = os.path.join(os.path.dirname(__file__), 'golden', 'gaba_extend.json')
    gold = json.load(open(path))
    for grp in gold['groups']:
        hip = G.Hip(**grp['params'])
        jobs = [(np.array(j['a'], dtype=np.uint8), j['apos'], j['arev'], np.array(j['b'], dtype=np.uint8), j['bpos'], j['brev'], j['bw'], 1)
                for j in grp['jobs']]
        got = hip.extend_batch(jobs)
        for i, j in enumerate(grp['jobs']):
            want = j['expect']; want['fills'] = [tuple(f) for f in want['fills']]; want['pos'] = tuple(want['pos'])
            if 'segs' in want: want['segs'] = [tuple(s) for s in want['segs']]
            assert got[i] == want, (grp['name'], i, _diff(got[i], want))


# ---- the per-call API (gaba_dp_fill_root / fill / search_max / trace, gaba.h:266-357) against the oracle's same calls ----
class _Sec(ctypes.Structure):
    _fields_ = [('id', ctypes.c_uint32), ('len', ctypes.c_uint32), ('base', ctypes.c_void_p)]

class _Fill(ctypes.Structure):
    _fields_ = [('aid', ctypes.c_uint32), ('bid', ctypes.c_uint32), ('ascnt', ctypes.c_uint32), ('bscnt', ctypes.c_uint32),
                ('apos', ctypes.c_uint64), ('bpos', ctypes.c_uint64), ('max', ctypes.c_int64), ('status', ctypes.c_uint32), ('reserved', ctypes.c_uint32 * 5)]

class _Pos(ctypes.Structure):
    _fields_ = [('aid', ctypes.c_uint32), ('bid', ctypes.c_uint32), ('apos', ctypes.c_uint32), ('bpos', ctypes.c_uint32), ('plen', ctypes.c_uint64)]

class _Seg(ctypes.Structure):
    _fields_ = [('aid', ctypes.c_uint32), ('bid', ctypes.c_uint32), ('apos', ctypes.c_uint32), ('bpos', ctypes.c_uint32),
                ('alen', ctypes.c_uint32), ('blen', ctypes.c_uint32), ('ppos', ctypes.c_uint64)]

class _OraAln(ctypes.Structure):          # oracle/ora_gaba.h og_alignment_t
    _fields_ = [('score', ctypes.c_int64), ('identity', ctypes.c_double), ('agcnt', ctypes.c_uint32), ('bgcnt', ctypes.c_uint32), ('dcnt', ctypes.c_uint32),
                ('slen', ctypes.c_uint32), ('seg', ctypes.POINTER(_Seg)), ('plen', ctypes.c_uint32), ('path', ctypes.POINTER(ctypes.c_uint32))]

class _DevAln(ctypes.Structure):          # include/gaba.h gaba_alignment_s (path[] follows)
    _fields_ = [('reserved', ctypes.c_void_p * 2), ('score', ctypes.c_int64), ('identity', ctypes.c_double), ('agcnt', ctypes.c_uint32), ('bgcnt', ctypes.c_uint32),
                ('dcnt', ctypes.c_uint32), ('slen', ctypes.c_uint32), ('seg', ctypes.POINTER(_Seg)), ('plen', ctypes.c_uint32), ('padding', ctypes.c_uint32)]

def _fill_tuple(f):
    return (f.aid, f.bid, f.ascnt, f.bscnt, f.apos, f.bpos, f.max, f.status)

def _drive(fill_root, fill, search, trace, asecs, bsecs, tail_a, tail_b, apos, bpos):
    """the call pattern of the reference's own multi-section unit tests (gaba.c:5590-5765): feed the next section of a side
    whenever the fill reports UPDATE_A / UPDATE_B, the N tail once a side is exhausted, until TERM or both tails are in"""
    ai = bi = 0; fills = []
    f = fill_root(asecs[0], apos, bsecs[0], bpos)
    fills.append(f)
    ta = tb = False
    for _ in range(64):
        st = f.contents.status
        if st & 0x8000: break
        if st & 0x000f:
            ai += 1
            if ai >= len(asecs):
                if ta: break
                ta = True
        if st & 0x00f0:
            bi += 1
            if bi >= len(bsecs):
                if tb: break
                tb = True
        f = fill(f, asecs[ai] if ai < len(asecs) else tail_a, bsecs[bi] if bi < len(bsecs) else tail_b)
        fills.append(f)
    best = max(fills, key=lambda q: q.contents.max)
    return fills, best, search(best), trace(best)

@pytest.mark.gpu
@pytest.mark.parametrize('bw', [0, 1, 2])
def test_per_call_api_matches_oracle(bw):
    P = G.PACBIO
    ora = G.Oracle(**P); OL = ora.L
    hip = G.Hip(**P); HL = hip.L
    for fn in ('og_dp_fill_root', 'og_dp_fill'): getattr(OL, fn).restype = ctypes.POINTER(_Fill)
    OL.og_dp_search_max.restype = ctypes.POINTER(_Pos); OL.og_dp_trace.restype = ctypes.POINTER(_OraAln)
    for fn in ('gaba_dp_fill_root', 'gaba_dp_fill'): getattr(HL, fn).restype = ctypes.POINTER(_Fill)
    HL.gaba_dp_search_max.restype = ctypes.POINTER(_Pos); HL.gaba_dp_trace.restype = ctypes.POINTER(_DevAln)
    HL.gaba_dp_init_bw.restype = ctypes.c_void_p; HL.gaba_arena_upload.restype = ctypes.c_void_p
    rng = np.random.default_rng(1234 + bw)
    # every other trial traces through a caller-supplied allocator (gaba_alloc_t, gaba.h:61-75): one lmalloc per alignment, handed back to lfree
    libc = ctypes.CDLL(None); libc.malloc.restype = ctypes.c_void_p; libc.malloc.argtypes = [ctypes.c_size_t]; libc.free.argtypes = [ctypes.c_void_p]
    LM = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t); LF = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p)
    class _Alloc(ctypes.Structure): _fields_ = [('opaque', ctypes.c_void_p), ('lmalloc', LM), ('lfree', LF)]
    live = {}
    def _lm(opaque, size): p = libc.malloc(size); live[p] = opaque; return p
    def _lf(opaque, ptr): assert live.pop(ptr) == opaque; libc.free(ptr)
    alloc = _Alloc(0x1234, LM(_lm), LF(_lf)); n_custom = 0
    for trial in range(12):
        L = int(rng.integers(40, 1500))
        a = rng.integers(0, 4, L, dtype=np.uint8)
        b = G.mutate(rng, a, 0.02, 0.08, 0.04)
        if len(b) < 8: continue
        # host arrays: [64 pad | sections back to back | 96 x N tail | 64 pad]
        ha = np.concatenate([np.full(64, 4, np.uint8), a, np.full(96 + 64, 4, np.uint8)])
        hb = np.concatenate([np.full(64, 4, np.uint8), b, np.full(96 + 64, 4, np.uint8)])
        def cut(n, k):
            pts = sorted(set([0, n] + [int(x) for x in rng.integers(1, n, k)]))
            return [(pts[i], pts[i + 1] - pts[i]) for i in range(len(pts) - 1)]
        ca = cut(len(a), int(rng.integers(0, 3))); cb = cut(len(b), int(rng.integers(0, 3)))
        mk = lambda h, off, ln, sid: _Sec(sid, ln, h.ctypes.data + 64 + off)
        asecs = [mk(ha, o, n, 2 * i) for i, (o, n) in enumerate(ca)]; bsecs = [mk(hb, o, n, 2 * i + 1) for i, (o, n) in enumerate(cb)]
        tail_a = _Sec(0xfffffffe, 96, ha.ctypes.data + 64 + len(a)); tail_b = _Sec(0xfffffffe, 96, hb.ctypes.data + 64 + len(b))
        apos = int(rng.integers(0, max(1, asecs[0].len // 2))); bpos = int(rng.integers(0, max(1, bsecs[0].len // 2)))
        # oracle
        OL.og_dp_flush(ora.dp)
        o_f, o_best, o_pp, o_aln = _drive(
            lambda x, ap, y, bp: OL.og_dp_fill_root(ora.dp, bw, ctypes.byref(x), ap, ctypes.byref(y), bp, 0),
            lambda f, x, y: OL.og_dp_fill(ora.dp, f, ctypes.byref(x), ctypes.byref(y), 0),
            lambda f: OL.og_dp_search_max(ora.dp, f), lambda f: OL.og_dp_trace(ora.dp, f), asecs, bsecs, tail_a, tail_b, apos, bpos)
        # device
        ara = ctypes.c_void_p(HL.gaba_arena_upload(ha.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(ha))))
        arb = ctypes.c_void_p(HL.gaba_arena_upload(hb.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(hb))))
        dp = ctypes.c_void_p(HL.gaba_dp_init_bw(ctypes.c_void_p(hip.ctx), bw)); assert dp
        d_f, d_best, d_pp, d_aln = _drive(
            lambda x, ap, y, bp: HL.gaba_dp_fill_root(dp, ctypes.byref(x), ap, ctypes.byref(y), bp, 0),
            lambda f, x, y: HL.gaba_dp_fill(dp, f, ctypes.byref(x), ctypes.byref(y), 0),
            lambda f: HL.gaba_dp_search_max(dp, f), lambda f: HL.gaba_dp_trace(dp, f, ctypes.byref(alloc) if trial & 1 else None), asecs, bsecs, tail_a, tail_b, apos, bpos)
        assert [_fill_tuple(f.contents) for f in d_f] == [_fill_tuple(f.contents) for f in o_f], 'fills differ (trial %d)' % trial
        op, dpp = o_pp.contents, d_pp.contents
        assert (op.aid, op.bid, op.apos, op.bpos, op.plen) == (dpp.aid, dpp.bid, dpp.apos, dpp.bpos, dpp.plen)
        assert bool(o_aln) == bool(d_aln)
        if o_aln:
            oa, da = o_aln.contents, d_aln.contents
            assert (oa.score, oa.agcnt, oa.bgcnt, oa.dcnt, oa.slen, oa.plen) == (da.score, da.agcnt, da.bgcnt, da.dcnt, da.slen, da.plen)
            assert np.float64(oa.identity).tobytes() == np.float64(da.identity).tobytes()
            nw = (oa.plen + 31) // 32
            dpath = ctypes.cast(ctypes.addressof(da) + ctypes.sizeof(_DevAln), ctypes.POINTER(ctypes.c_uint32))
            assert [oa.path[i] for i in range(nw)] == [dpath[i] for i in range(nw)]
            seg = lambda s: (s.aid, s.bid, s.apos, s.bpos, s.alen, s.blen, s.ppos)
            assert [seg(oa.seg[i]) for i in range(oa.slen)] == [seg(da.seg[i]) for i in range(da.slen)]
            if trial & 1: assert len(live) == 1; n_custom += 1
            OL.og_aln_free(o_aln); HL.gaba_dp_res_free(dp, d_aln)
            assert len(live) == 0
        HL.gaba_dp_clean(dp); HL.gaba_arena_free(ara); HL.gaba_arena_free(arb)
    assert n_custom > 0
