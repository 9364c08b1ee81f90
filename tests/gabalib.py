"""ctypes bindings shared by the tests: the CPU oracle (oracle/liboracle.so), the compiled reference
(oracle/_ref/libgaba_ref.so, when present) -- identical record layouts (oracle/ora_gaba.h og_xresult_t,
oracle/ref_harness/gaba_ref_shim.c shim_result_t)."""
import ctypes, os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

class Fill(ctypes.Structure):
    _fields_ = [('max', ctypes.c_int64), ('status', ctypes.c_uint32), ('aid', ctypes.c_uint32), ('bid', ctypes.c_uint32),
                ('ascnt', ctypes.c_uint32), ('bscnt', ctypes.c_uint32), ('apos', ctypes.c_uint64), ('bpos', ctypes.c_uint64)]
    def tup(self):
        return (self.max, self.status, self.aid, self.bid, self.ascnt, self.bscnt, self.apos, self.bpos)

class Seg(ctypes.Structure):
    _fields_ = [('aid', ctypes.c_uint32), ('bid', ctypes.c_uint32), ('apos', ctypes.c_uint32), ('bpos', ctypes.c_uint32),
                ('alen', ctypes.c_uint32), ('blen', ctypes.c_uint32), ('ppos', ctypes.c_uint64)]
    def tup(self):
        return (self.aid, self.bid, self.apos, self.bpos, self.alen, self.blen, self.ppos)

class XResult(ctypes.Structure):
    _fields_ = [('n_fill', ctypes.c_uint32), ('max_fill_idx', ctypes.c_uint32), ('fill', Fill * 8),
                ('p_aid', ctypes.c_uint32), ('p_bid', ctypes.c_uint32), ('p_apos', ctypes.c_uint32), ('p_bpos', ctypes.c_uint32),
                ('p_plen', ctypes.c_uint64),
                ('traced', ctypes.c_int32), ('score', ctypes.c_int64), ('identity', ctypes.c_double),
                ('agcnt', ctypes.c_uint32), ('bgcnt', ctypes.c_uint32), ('dcnt', ctypes.c_uint32), ('slen', ctypes.c_uint32),
                ('plen', ctypes.c_uint32), ('seg', Seg * 16), ('n_path_words', ctypes.c_uint32)]

    def as_dict(self, path):
        d = dict(n_fill=self.n_fill, max_fill_idx=self.max_fill_idx,
                 fills=[self.fill[i].tup() for i in range(min(self.n_fill, 8))],
                 pos=(self.p_aid, self.p_bid, self.p_apos, self.p_bpos, self.p_plen), traced=self.traced)
        if self.traced == 1:
            d.update(score=self.score, identity=np.float64(self.identity).tobytes().hex(), agcnt=self.agcnt, bgcnt=self.bgcnt,
                     dcnt=self.dcnt, slen=self.slen, plen=self.plen,
                     segs=[self.seg[i].tup() for i in range(min(self.slen, 16))],
                     path=[int(x) for x in path[:self.n_path_words]])
        return d

PACBIO = dict(m=2, x=4, gi=4, ge=2, gfa=3, gfb=3, xdrop=50)      # minialign.c:5854
ONT1DSQ = dict(m=2, x=6, gi=6, ge=2, gfa=4, gfb=4, xdrop=50)     # minialign.c:5857-5877 (-a2 -b6 -p6 -q2 -r4,4)
AFFINE_DEFAULT = dict(m=1, x=1, gi=1, ge=1, gfa=0, gfb=0, xdrop=50)  # minialign.c:6158-6161
LINEAR_AVA = dict(m=2, x=3, gi=0, ge=2, gfa=0, gfb=0, xdrop=50)      # minialign.c:5879 (-xava: -a2 -b3 -p0 -q2); gi == 0 selects the linear-gap build

def score_matrix(m, x):
    return (ctypes.c_int8 * 16)(*[m if (i & 3) == (i >> 2) else -x for i in range(16)])

class _Engine:
    """common driver: ext(a, apos, arev, b, bpos, brev, bw_idx, trace) -> dict"""
    def extend(self, a, apos, arev, b, bpos, brev, bw_idx=0, trace=1):
        a = np.ascontiguousarray(a, dtype=np.uint8); b = np.ascontiguousarray(b, dtype=np.uint8)
        # keep 64 bytes of slack on both sides: the reference's 32-byte loads overrun the sequences
        pa = np.full(len(a) + 128, 4, dtype=np.uint8); pa[64:64 + len(a)] = a
        pb = np.full(len(b) + 128, 4, dtype=np.uint8); pb[64:64 + len(b)] = b
        res = XResult(); path = (ctypes.c_uint32 * ((len(a) + len(b) + 512) // 32 + 32))()
        r = self._extend(self.dp, bw_idx, ctypes.c_void_p(pa.ctypes.data + 64), len(a), apos, int(arev),
                         ctypes.c_void_p(pb.ctypes.data + 64), len(b), bpos, int(brev), int(trace), ctypes.byref(res), path)
        assert r == 0
        return res.as_dict(path)

class Oracle(_Engine):
    def __init__(self, m, x, gi, ge, gfa, gfb, xdrop):
        L = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'liboracle.so'))
        self.L = L
        class P(ctypes.Structure):
            _fields_ = [('sm', ctypes.c_int8 * 16), ('gi', ctypes.c_int8), ('ge', ctypes.c_int8), ('gfa', ctypes.c_int8),
                        ('gfb', ctypes.c_int8), ('xdrop', ctypes.c_int8), ('ft', ctypes.c_uint8), ('reserved', ctypes.c_void_p),
                        ('_pad', ctypes.c_uint64)]
        p = P(); p.sm = score_matrix(m, x); p.gi, p.ge, p.gfa, p.gfb, p.xdrop = gi, ge, gfa, gfb, xdrop
        L.og_init.restype = ctypes.c_void_p; L.og_dp_init.restype = ctypes.c_void_p
        self.ctx = L.og_init(ctypes.byref(p))
        assert self.ctx, 'og_init rejected the scores'
        self.dp = ctypes.c_void_p(L.og_dp_init(ctypes.c_void_p(self.ctx)))
        self._extend = L.og_extend

class Reference(_Engine):
    PATH = os.path.join(ROOT, 'oracle', '_ref', 'libgaba_ref.so')
    @staticmethod
    def available():
        return os.path.exists(Reference.PATH)
    def __init__(self, m, x, gi, ge, gfa, gfb, xdrop):
        L = ctypes.CDLL(self.PATH)
        self.L = L
        L.shim_init.restype = ctypes.c_void_p; L.shim_dp_init.restype = ctypes.c_void_p
        self.ctx = L.shim_init(score_matrix(m, x), gi, ge, gfa, gfb, xdrop)
        assert self.ctx
        self.dp = ctypes.c_void_p(L.shim_dp_init(ctypes.c_void_p(self.ctx)))
        self._extend = L.shim_extend

def mutate(rng, seq, sub, ins, dele):
    """PBSIM-like error model on a uint8 0..3 array (own generator)"""
    out = []
    r = rng.random(len(seq) * 2 + 16)
    k = 0
    for c in seq:
        x = r[k]; k += 1
        if x < dele:
            continue
        if x < dele + sub:
            out.append((c + 1 + int(r[k] * 3)) & 3); k += 1
            continue
        out.append(c)
        if x > 1.0 - ins:
            out.append(int(r[k] * 4) & 3); k += 1
    return np.array(out, dtype=np.uint8)


# ---------------------------------------------------------------------------------------------
# product library (HIP): include/gaba.h through ctypes
class Job(ctypes.Structure):
    _fields_ = [('a_off', ctypes.c_uint64), ('alen', ctypes.c_uint32), ('apos', ctypes.c_uint32),
                ('b_off', ctypes.c_uint64), ('blen', ctypes.c_uint32), ('bpos', ctypes.c_uint32),
                ('arev', ctypes.c_uint8), ('brev', ctypes.c_uint8), ('bw_idx', ctypes.c_uint8), ('do_trace', ctypes.c_uint8)]

class Params(ctypes.Structure):
    _fields_ = [('sm', ctypes.c_int8 * 16), ('gi', ctypes.c_int8), ('ge', ctypes.c_int8), ('gfa', ctypes.c_int8),
                ('gfb', ctypes.c_int8), ('xdrop', ctypes.c_int8), ('ft', ctypes.c_uint8), ('reserved', ctypes.c_void_p),
                ('_pad', ctypes.c_uint64)]

class BatchStats(ctypes.Structure):
    _fields_ = [('kernel_ms', ctypes.c_double), ('vectors', ctypes.c_uint64), ('blocks', ctypes.c_uint64), ('trace_steps', ctypes.c_uint64)]

def load_product():
    """load libminialign_amd.so; raises if it was not built (no silent fallback)"""
    p = os.environ.get('MM_LIB_OVERRIDE') or os.path.join(ROOT, 'minialign_amd', 'libminialign_amd.so')     # override: kernel experiments only
    if not os.path.exists(p):
        raise RuntimeError('minialign_amd/libminialign_amd.so is missing: run __graft_entry__.build()')
    L = ctypes.CDLL(p)
    L.gaba_init.restype = ctypes.c_void_p
    L.gaba_arena_upload.restype = ctypes.c_void_p
    return L

class Hip:
    """batched driver over gaba_dp_extend_batch; jobs: list of (a, apos, arev, b, bpos, brev, bw_idx, trace)"""
    def __init__(self, m, x, gi, ge, gfa, gfb, xdrop):
        self.L = load_product()
        p = Params(); p.sm = score_matrix(m, x); p.gi, p.ge, p.gfa, p.gfb, p.xdrop = gi, ge, gfa, gfb, xdrop
        self.ctx = self.L.gaba_init(ctypes.byref(p))
        if not self.ctx:
            raise RuntimeError('gaba_init failed (no HIP device?)')

    def extend_batch(self, jobs):
        n = len(jobs)
        aa, bb, J = [], [], (Job * n)()
        ao = bo = 0; maxp = 0
        for i, (a, apos, arev, b, bpos, brev, bw, tr) in enumerate(jobs):
            a = np.ascontiguousarray(a, dtype=np.uint8); b = np.ascontiguousarray(b, dtype=np.uint8)
            J[i].a_off, J[i].alen, J[i].apos = ao, len(a), apos
            J[i].b_off, J[i].blen, J[i].bpos = bo, len(b), bpos
            J[i].arev, J[i].brev, J[i].bw_idx, J[i].do_trace = int(arev), int(brev), bw, int(tr)
            aa.append(a); bb.append(b); ao += len(a); bo += len(b)
            maxp = max(maxp, len(a) + len(b) + 512)
        A = np.concatenate(aa); B = np.concatenate(bb)
        ha = self.L.gaba_arena_upload(A.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(A)))
        hb = self.L.gaba_arena_upload(B.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(B)))
        assert ha and hb
        stride = maxp // 32 + 32
        res = (XResult * n)(); paths = np.zeros(n * stride, dtype=np.uint32)
        rc = self.L.gaba_dp_extend_batch(ctypes.c_void_p(self.ctx), ctypes.c_void_p(ha), ctypes.c_void_p(hb), J, n, res,
                                         paths.ctypes.data_as(ctypes.c_void_p), stride)
        self.L.gaba_arena_free(ctypes.c_void_p(ha)); self.L.gaba_arena_free(ctypes.c_void_p(hb))
        assert rc == 0, 'gaba_dp_extend_batch rc=%d' % rc
        return [res[i].as_dict(paths[i * stride:(i + 1) * stride]) for i in range(n)]

    def stats(self):
        s = BatchStats(); self.L.gaba_last_stats(ctypes.c_void_p(self.ctx), ctypes.byref(s)); return s


def revcomp(s):
    r = s[::-1].copy(); m = r < 4; r[m] = 3 - r[m]; return r

def random_jobs(seed, n, max_len=6000, bw_choices=(0, 0, 0, 1, 2)):
    """seeded random extension jobs covering the edge cases the reference's unit tests probe (gaba.c:5351-5765):
    short / empty-ish inputs, tandem repeats, long indels, N runs, unequal lengths, start offsets, all strands"""
    rng = np.random.default_rng(seed)
    jobs = []
    for it in range(n):
        mode = int(rng.integers(0, 6))
        L = int(rng.integers(1, max_len)) if mode else int(rng.integers(1, 200))
        a = rng.integers(0, 4, L, dtype=np.uint8)
        if mode == 2:
            u = rng.integers(0, 4, int(rng.integers(1, 40)), dtype=np.uint8); a = np.tile(u, L // len(u) + 1)[:L]
        err = rng.uniform(0.0, 0.35)
        b = mutate(rng, a, err * 0.1, err * 0.6, err * 0.3)
        if len(b) == 0:
            b = rng.integers(0, 4, 5, dtype=np.uint8)
        if mode == 3:
            k = int(rng.integers(0, len(b)))
            b = np.concatenate([b[:k], rng.integers(0, 4, int(rng.integers(1, 120)), dtype=np.uint8), b[k:]])
        if mode == 4 and len(a) > 10:
            k = int(rng.integers(0, len(a) - 5)); a = a.copy(); a[k:k + int(rng.integers(1, 30))] = 4
        if rng.random() < 0.3:
            b = np.concatenate([b, rng.integers(0, 4, int(rng.integers(0, 300)), dtype=np.uint8)])
        if rng.random() < 0.2:
            a = np.concatenate([a, rng.integers(0, 4, int(rng.integers(0, 300)), dtype=np.uint8)])
        apos = int(rng.integers(0, max(1, len(a) // 3))) if rng.random() < 0.7 else int(rng.integers(0, len(a)))
        bpos = min(len(b) - 1, apos) if rng.random() < 0.8 else int(rng.integers(0, len(b)))
        arev = bool(rng.random() < 0.4); brev = bool(rng.random() < 0.4)
        aa = revcomp(a) if arev else a; bb = revcomp(b) if brev else b
        bw = int(rng.choice(bw_choices))
        jobs.append((aa, apos, arev, bb, bpos, brev, bw, 1))
    return jobs
