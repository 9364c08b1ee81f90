"""GPU: stage taps.  The outputs of the device stages in front of the extension -- K1 (sketch + index lookup + seed expansion), K2s (the unstable radix sort as a
permutation), K2p / K2c (window scans + chain sweep, root sort) -- compared directly with what the compiled reference's own mm_sketch / mm_seed / mm_chain gave
on the same reads (tests/golden/mm_golden.json, written by make_mm_golden.py from oracle/_ref's stage harness): number of minimizers, the sorted seed array
(md5 of its bytes, sentinel included) and the chain roots, per read -- not only through the final SAM."""
import ctypes, hashlib, json, os, tempfile
import numpy as np, pytest
import mmlib as M
from golden.make_mm_golden import make_inputs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
META = json.load(open(os.path.join(HERE, 'golden', 'mm_golden.json')))

@pytest.mark.parametrize('s', META['sets'], ids=[s['name'] for s in META['sets']])
def test_device_stages_match_the_reference_stage_vectors(s):
    from minialign_amd import multi
    os.environ.setdefault('MM_SLAB_GB', '6')
    L = multi.load_library(); assert L.mm_set_device(0) == 0
    L.mm_batch_tap_sketch.restype = ctypes.c_int64; L.mm_batch_tap_sketch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    L.mm_batch_tap.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    with tempfile.TemporaryDirectory() as d:
        ref, rd = make_inputs(s, d)
        o = ctypes.c_void_p(L.mm_opt_init()); argv = (ctypes.c_char_p * 4)(b'minialign', ('-x' + s['preset']).encode(), ref.encode(), rd.encode()); files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
        assert L.mm_opt_parse(o, 4, argv, files, 8, ctypes.byref(nf)) == 0
        mi = ctypes.c_void_p(L.mm_idx_gen(o, ref.encode())); al = ctypes.c_void_p(L.mm_align_init(o, mi)); assert al
        assert [L.mm_idx_occ(mi, i) for i in range(len(s['occ']))] == s['occ']
        reads = ctypes.c_void_p(L.mm_reads_load(rd.encode())); n = L.mm_reads_count(reads)
        names = [L.mm_reads_name(reads, i).decode() for i in range(n)]
        h = ctypes.c_void_p(L.mm_batch_pack(reads, 0, n)); assert h
        for st in s['stages']:
            i = names.index(st['read'])
            cap = 1 << 18
            seeds = np.zeros((cap, 4), dtype=np.uint32); roots = np.zeros(cap, dtype=np.uint64)
            n_min = ctypes.c_uint32(0); n_seeds = ctypes.c_uint32(0); n_roots = ctypes.c_uint32(0)
            assert L.mm_batch_tap(al, h, i, ctypes.byref(n_min), seeds.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(n_seeds), roots.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(n_roots)) == 0
            assert n_min.value == st['sketch_n'], (st['read'], n_min.value, st['sketch_n'])
            words = np.zeros(n_min.value + 8, dtype=np.uint64)
            assert L.mm_batch_tap_sketch(al, h, i, words.ctypes.data_as(ctypes.c_void_p), len(words)) == st['sketch_n']
            assert [int(x) for x in words[:8]] == st['sketch_head'], st['read']
            assert hashlib.md5(words[:n_min.value].tobytes()).hexdigest() == st['sketch_md5'], st['read']          # K1's minimizer stream, word for word
            assert n_seeds.value == st['seed_n']
            assert hashlib.md5(seeds[:n_seeds.value].tobytes()).hexdigest() == st['seed_md5'], st['read']
            assert [int(x) for x in roots[:n_roots.value]] == st['chain'], st['read']
        L.mm_batch_free(h)
