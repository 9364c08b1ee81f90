"""ctypes bindings for the mapper-level tests: the CPU oracle (oracle/liboracle.so, ora_mm.h) and -- when present --
the stage harness of the compiled reference (oracle/_ref/libmm_ref.so, oracle/ref_harness/mm_ref_shim.c)."""
import ctypes, os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENC = np.zeros(256, dtype=np.uint8)
for ch, v in (('A', 0), ('C', 1), ('G', 2), ('T', 3), ('U', 3), ('N', 4)):
    for c in (ch, ch.lower()):
        ENC[ord(c)] = v

def read_fasta(fn):
    """-> list of (name, uint8 array 0..4); same letter handling as minialign.c:223-229 (low nibble table)"""
    tbl = np.zeros(16, dtype=np.uint8)
    for ch, v in (('A', 0), ('C', 1), ('G', 2), ('T', 3), ('U', 3), ('N', 4)):
        tbl[ord(ch) & 15] = v
    out = []; name = None; chunks = []
    fq = False; state = 0
    with open(fn, 'rb') as f:
        for line in f:
            line = line.rstrip(b'\r\n')
            if state == 2:
                state = 0; continue
            if not line: continue
            if line[:1] in (b'>', b'@') and state == 0 or (line[:1] == b'>'):
                if name is not None: out.append((name, np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)))
                fq = line[:1] == b'@'
                name = line[1:].split()[0].decode(); chunks = []; state = 1 if fq else 0
                continue
            if fq and line[:1] == b'+':
                state = 2; continue
            chunks.append(tbl[np.frombuffer(line, dtype=np.uint8) & 15])
            if fq: pass
    if name is not None: out.append((name, np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)))
    return out

class OmOpt(ctypes.Structure):
    _fields_ = [('k', ctypes.c_uint32), ('w', ctypes.c_uint32), ('b', ctypes.c_uint32), ('n_frq', ctypes.c_uint32), ('frq', ctypes.c_float * 16),
                ('wlen', ctypes.c_uint32), ('glen', ctypes.c_uint32), ('min_score', ctypes.c_uint32), ('min_ratio', ctypes.c_float),
                ('p_sm', ctypes.c_int8 * 16), ('p_gi', ctypes.c_int8), ('p_ge', ctypes.c_int8), ('p_gfa', ctypes.c_int8), ('p_gfb', ctypes.c_int8),
                ('p_xdrop', ctypes.c_int8), ('p_ft', ctypes.c_uint8), ('p_reserved', ctypes.c_void_p), ('p_pad', ctypes.c_uint64),
                ('arg_line', ctypes.c_char_p), ('min_len', ctypes.c_uint32),
                ('flag', ctypes.c_uint64), ('tags', ctypes.c_uint64), ('rg_line', ctypes.c_char_p), ('rg_id', ctypes.c_char_p), ('keep_qual', ctypes.c_uint32), ('format', ctypes.c_uint32),
                ('ava', ctypes.c_uint32), ('circ_set', ctypes.c_uint32), ('circ_names', ctypes.c_char_p)]

class OmSeq(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char_p), ('l_name', ctypes.c_uint32), ('seq', ctypes.c_void_p), ('l_seq', ctypes.c_uint32), ('qual', ctypes.c_char_p), ('comment', ctypes.c_char_p)]

class OracleMM:
    def __init__(self, preset, ref, circ=None):
        """ref: list of (name, uint8 array); circ: None, or a comma list of circular sequence names (b'' = all)"""
        L = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'liboracle.so')); self.L = L
        self.opt = OmOpt(); assert L.om_opt_init(ctypes.byref(self.opt), preset.encode()) == 0
        if circ is not None: self.opt.circ_set = 1; self.opt.circ_names = circ
        self.ref = ref
        self._keep = [np.ascontiguousarray(s) for _, s in ref]
        self.seqs = (OmSeq * len(ref))()
        for i, (n, s) in enumerate(ref):
            self.seqs[i].name = n.encode(); self.seqs[i].l_name = len(n); self.seqs[i].seq = self._keep[i].ctypes.data; self.seqs[i].l_seq = len(s)
        L.om_idx_build.restype = ctypes.c_void_p; L.om_align_init.restype = ctypes.c_void_p
        self.mi = ctypes.c_void_p(L.om_idx_build(ctypes.byref(self.opt), self.seqs, len(ref)))
        self.al = ctypes.c_void_p(L.om_align_init(ctypes.byref(self.opt), self.mi))
        L.om_sketch.restype = ctypes.c_uint64; L.om_stage_seed.restype = ctypes.c_uint64; L.om_stage_chain.restype = ctypes.c_uint64
        L.om_idx_get.restype = ctypes.POINTER(ctypes.c_uint64)
    def occ(self): return [self.L.om_idx_occ(self.mi, i) for i in range(self.opt.n_frq)]
    def sketch(self, seq):
        out = np.zeros(4 * len(seq) // self.opt.w + 512, dtype=np.uint64)
        n = self.L.om_sketch(self.opt.w, self.opt.k, seq.ctypes.data_as(ctypes.c_void_p), len(seq), out.ctypes.data_as(ctypes.c_void_p))
        return out[:n].copy()
    def idx_get(self, minier):
        n = ctypes.c_uint32(0)
        p = self.L.om_idx_get(self.mi, ctypes.c_uint64(int(minier)), ctypes.byref(n))
        return [p[i] for i in range(n.value)]
    def seed(self, seq, it=0):
        p = ctypes.c_void_p()
        n = self.L.om_stage_seed(self.al, len(seq), seq.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(it), ctypes.byref(p))
        if n == 0: return np.zeros((0, 4), dtype=np.uint32)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint32)), shape=(n, 4)).copy()
    def chain(self):
        p = ctypes.c_void_p()
        n = self.L.om_stage_chain(self.al, ctypes.byref(p))
        if n == 0: return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint64)), shape=(n,)).copy()

class RefMM:
    PATH = os.path.join(ROOT, 'oracle', '_ref', 'libmm_ref.so')
    @staticmethod
    def available(): return os.path.exists(RefMM.PATH)
    def __init__(self, args, ref_fa):
        L = ctypes.CDLL(self.PATH); self.L = L
        argv = (ctypes.c_char_p * (len(args) + 2))(*([b'minialign'] + [a.encode() for a in args] + [None]))
        L.mmref_open.restype = ctypes.c_void_p
        self.h = ctypes.c_void_p(L.mmref_open(argv, ref_fa.encode()))
        assert self.h
        L.mmref_sketch.restype = ctypes.c_uint64; L.mmref_seed.restype = ctypes.c_uint64; L.mmref_chain.restype = ctypes.c_uint64
        self.w = L.mmref_kwb(self.h, 1)
    def occ(self, n=3): return [self.L.mmref_occ(self.h, i) for i in range(n)]
    def sketch(self, seq):
        pad = np.concatenate([seq, np.zeros(64, np.uint8)])
        out = np.zeros(4 * len(seq) // self.w + 512, dtype=np.uint64)
        n = self.L.mmref_sketch(self.h, pad.ctypes.data_as(ctypes.c_void_p), len(seq), out.ctypes.data_as(ctypes.c_void_p))
        return out[:n].copy()
    def idx_get(self, minier):
        out = (ctypes.c_uint64 * 4096)()
        n = self.L.mmref_idx_get(self.h, ctypes.c_uint64(int(minier)), out, 4096)
        return [out[i] for i in range(min(n, 4096))]
    def seed(self, seq, it=0):
        self._pad = np.concatenate([np.zeros(64, np.uint8), seq, np.zeros(64, np.uint8)])
        cap = 1 << 20
        out = np.zeros((cap, 4), dtype=np.uint32)
        n = self.L.mmref_seed(self.h, ctypes.c_void_p(self._pad.ctypes.data + 64), len(seq), ctypes.c_uint64(it), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(cap))
        return out[:n].copy()
    def chain(self):
        cap = 1 << 18
        roots = np.zeros(cap, dtype=np.uint64); leaves = np.zeros((cap, 4), dtype=np.uint32); nl = ctypes.c_uint64(0)
        n = self.L.mmref_chain(self.h, roots.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(cap), leaves.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nl))
        return roots[:n].copy()

def gensim(*args, out):
    exe = os.path.join(ROOT, 'tools', 'gensim')
    if not os.path.exists(exe):
        subprocess.check_call(['gcc', '-O2', '-o', exe, os.path.join(ROOT, 'tools', 'gensim.c'), '-lm'])
    with open(out, 'wb') as f:
        subprocess.check_call([exe] + [str(a) for a in args], stdout=f)
