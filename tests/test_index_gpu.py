"""GPU: the index built on the device (minialign_amd/csrc/mm_index.hpp: reference sketch, stable partition into buckets in reference order, the unstable per-bucket
radix sort replayed, occurrence thresholds, table fill with the reference's cursor quirk) against the ORACLE's index (oracle/ora_mm.c om_idx_build, pinned on the
compiled reference): the occurrence thresholds occ[] and, for every minimizer of the reference, the value list in the reference's order -- on multi-contig,
repeat-rich (keys above the last threshold: the rest of their bucket is dropped, minialign.c:2927-2931) and circular references, and for k / w / bucket-bit
settings that give one, two and three non-trivial digit levels in the per-bucket sort and buckets from a handful to tens of thousands of elements.  Every SAM golden
of the suite runs through the device-built index as well (it is the default of mm_idx_gen)."""
import ctypes, os, sys, tempfile
import numpy as np, pytest
import mmlib as M

pytestmark = pytest.mark.gpu

@pytest.fixture(scope='module')
def lib():
    from minialign_amd import multi
    os.environ.pop('MM_HOST_INDEX', None)
    L = multi.load_library(); assert L.mm_set_device(0) == 0
    for f in ('mm_opt_init', 'mm_idx_gen', 'mm_idx_load'): getattr(L, f).restype = ctypes.c_void_p
    L.mm_idx_get.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32]
    return L

def _build(L, opts, ref, host=False):
    if host: os.environ['MM_HOST_INDEX'] = '1'
    try:
        o = ctypes.c_void_p(L.mm_opt_init()); args = [b'minialign'] + [x.encode() for x in opts] + [ref.encode()]
        argv = (ctypes.c_char_p * len(args))(*args); files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
        assert L.mm_opt_parse(o, len(args), argv, files, 8, ctypes.byref(nf)) == 0 and nf.value == 1
        mi = ctypes.c_void_p(L.mm_idx_gen(o, ref.encode())); assert mi
        return mi
    finally:
        os.environ.pop('MM_HOST_INDEX', None)

def _lists(L, mi, keys):
    buf = (ctypes.c_uint64 * (1 << 18))()
    return [[int(buf[i]) for i in range(min(L.mm_idx_get(mi, k, buf, 1 << 18), 1 << 18))] for k in keys]

@pytest.mark.parametrize('name,genome,opts,preset', [
    ('three_contigs', (901, 200000, 3, 0.15), ['-xpacbio'], 'pacbio'),                        # buckets of a handful of elements: the insertion-sort path only
    ('repeat_rich', (911, 6000000, 6, 0.45), ['-xpacbio'], 'pacbio'),                         # 64 .. 100 elements per bucket, keys above the last threshold (cursor quirk)
    ('many_contigs', (921, 20000000, 300, 0.10), ['-xpacbio'], 'pacbio'),                     # 200+ per bucket: one radix level and more
    ('ont', (931, 3000000, 4, 0.20), ['-xont.1dsq'], 'ont.1dsq'),
])
def test_device_index_matches_the_oracle_index(lib, name, genome, opts, preset):
    L = lib
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); M.gensim('genome', *genome, out=ref)
        mi = _build(L, opts, ref)
        refseq = M.read_fasta(ref); ora = M.OracleMM(preset, refseq)
        n_occ = len(ora.occ())
        assert [L.mm_idx_occ(mi, i) for i in range(n_occ)] == ora.occ()
        step = max(1, sum(len(q) for _, q in refseq) // 1500000)          # every minimizer of the small references, every few of the larger ones
        keys = sorted(set(int(m) >> 8 for _, q in refseq for m in ora.sketch(q)[::step])) + [12345, 1 << 29]
        got = _lists(L, mi, keys); want = [[int(v) for v in ora.idx_get(k)] for k in keys]
        assert got == want, name
        assert sum(len(v) > 1 for v in want) > 0 and sum(len(v) == 0 for v in want) >= 2
        L.mm_idx_destroy(mi)

@pytest.mark.parametrize('opts', [['-xpacbio', '-k19', '-w7', '-B12'], ['-xpacbio', '-k12', '-w5', '-B6'], ['-xpacbio', '-k15', '-w10', '-B10', '-f0.2,0.05,0.002'], ['-xava']],
                         ids=['three-digit-levels', 'large-buckets', 'low-thresholds', 'ava'])
def test_device_index_equals_the_host_build_for_other_settings(lib, opts):
    """k / w / bucket bits / thresholds beyond the presets: the device build against the library's own host build (which the CPU suite pins on the oracle), list by list"""
    L = lib
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); M.gensim('genome', 941, 8000000, 12, 0.35, out=ref)
        dev = _build(L, opts, ref); host = _build(L, opts, ref, host=True)
        assert [L.mm_idx_occ(dev, i) for i in range(3)] == [L.mm_idx_occ(host, i) for i in range(3)]
        # keys: the minimizers of a stretch of every contig, through the library's own sketch entry
        L.mm_sketch.restype = ctypes.c_uint32
        kk = int([x for x in opts if x.startswith('-k')][0][2:]) if any(x.startswith('-k') for x in opts) else 15
        ww = int([x for x in opts if x.startswith('-w')][0][2:]) if any(x.startswith('-w') for x in opts) else (10 if opts[0] != '-xava' else 5)
        keys = set()
        for _, q in M.read_fasta(ref):
            q = np.ascontiguousarray(q[:200000]); words = np.zeros(len(q), dtype=np.uint64)
            n = L.mm_sketch(q.ctypes.data_as(ctypes.c_void_p), len(q), ww, kk, words.ctypes.data_as(ctypes.c_void_p), None, len(words))
            keys.update(int(x) >> 8 for x in words[:n])
        keys = sorted(keys)
        a = _lists(L, dev, keys); b = _lists(L, host, keys)
        assert a == b
        assert sum(len(v) > 0 for v in a) > 100          # (with k = 12 most keys of this reference are above the last threshold and dropped)
        L.mm_idx_destroy(dev); L.mm_idx_destroy(host)

def test_device_index_of_circular_references_matches_the_oracle(lib):
    L = lib
    sys.path.insert(0, os.path.join(M.ROOT, 'tests', 'golden'))
    from make_circ_golden import make_circ_inputs
    with tempfile.TemporaryDirectory() as d:
        ref, _ = make_circ_inputs(d)
        refseq = M.read_fasta(ref)
        for copt, names in (('-c*', None), ('-cplasmid', b'plasmid')):
            mi = _build(L, ['-xpacbio', copt], ref)
            ora = M.OracleMM('pacbio', refseq, circ=(names if names else b''))
            assert [L.mm_idx_occ(mi, i) for i in range(3)] == ora.occ()[:3]
            keys = sorted(set(int(m) >> 8 for _, q in refseq for m in ora.sketch(np.concatenate([q, q[:40]]))))
            assert _lists(L, mi, keys) == [[int(v) for v in ora.idx_get(k)] for k in keys]
            L.mm_idx_destroy(mi)

def test_index_file_from_a_device_built_index_round_trips(lib):
    """-d: the host copy of a device-built index is fetched for mm_idx_dump; the block read back answers as the index in HBM does"""
    L = lib
    libc = ctypes.CDLL(None); libc.fopen.restype = ctypes.c_void_p; libc.fclose.argtypes = [ctypes.c_void_p]
    with tempfile.TemporaryDirectory() as d:
        ref = os.path.join(d, 'ref.fa'); mai = os.path.join(d, 'x.mai'); M.gensim('genome', 951, 1500000, 5, 0.3, out=ref)
        mi = _build(L, ['-xpacbio'], ref)
        fp = ctypes.c_void_p(libc.fopen(mai.encode(), b'wb')); assert L.mm_idx_dump(mi, fp) == 0; libc.fclose(fp)
        fp = ctypes.c_void_p(libc.fopen(mai.encode(), b'rb')); eof = ctypes.c_int(0); m2 = ctypes.c_void_p(L.mm_idx_load(fp, ctypes.byref(eof))); libc.fclose(fp); assert m2
        ora = M.OracleMM('pacbio', M.read_fasta(ref))
        keys = sorted(set(int(m) >> 8 for _, q in M.read_fasta(ref) for m in ora.sketch(q)[::5]))
        assert _lists(L, mi, keys) == _lists(L, m2, keys) == [[int(v) for v in ora.idx_get(k)] for k in keys]
        L.mm_idx_destroy(mi); L.mm_idx_destroy(m2)
