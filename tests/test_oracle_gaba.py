"""CPU tests (no GPU): the oracle (oracle/ora_gaba.c) against (1) the committed golden vectors generated from the
compiled reference and (2), when oracle/_ref is present, the compiled reference itself on fresh random jobs; plus
the CIGAR known-answer values the reference's own unit tests hold (gaba.c:4297-4522)."""
import ctypes, json, os
import numpy as np, pytest
import gabalib as G

def _norm(want):
    want = dict(want)
    want['fills'] = [tuple(f) for f in want['fills']]; want['pos'] = tuple(want['pos'])
    if 'segs' in want:
        want['segs'] = [tuple(s) for s in want['segs']]
    return want

def test_oracle_matches_golden():
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'gaba_extend.json')))
    n = 0
    for grp in gold['groups']:
        ora = G.Oracle(**grp['params'])
        for j in grp['jobs']:
            got = ora.extend(np.array(j['a'], dtype=np.uint8), j['apos'], j['arev'], np.array(j['b'], dtype=np.uint8), j['bpos'], j['brev'], j['bw'], 1)
            assert got == _norm(j['expect']), (grp['name'], n)
            n += 1
    assert n >= 60

LINEAR_SETS = [dict(m=1, x=1, gi=0, ge=1, gfa=0, gfb=0, xdrop=50), dict(m=1, x=2, gi=0, ge=1, gfa=0, gfb=0, xdrop=50), dict(m=2, x=4, gi=0, ge=3, gfa=0, gfb=0, xdrop=50),
               dict(m=1, x=3, gi=0, ge=2, gfa=0, gfb=0, xdrop=50)]
@pytest.mark.skipif(not G.Reference.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("P,seed", [(G.PACBIO, 21), (G.ONT1DSQ, 22), (G.AFFINE_DEFAULT, 23), (G.LINEAR_AVA, 24)] + [(p, 25 + i) for i, p in enumerate(LINEAR_SETS)])
def test_oracle_matches_compiled_reference(P, seed):
    """gi == 0 runs the reference's linear-gap build on its side and the affine recurrences with gi = 0 on the oracle's: same fills, positions, paths, counts"""
    ora = G.Oracle(**P); ref = G.Reference(**P)
    for j in G.random_jobs(seed, 120, max_len=3000):
        assert ora.extend(*j) == ref.extend(*j)

def _cigar(L, fn, words, offset, length):
    # the parsers peek below path[0]: keep the {plen, 0x40000000} header of gaba_alignment_s (gaba.h:217) in front
    arr = (ctypes.c_uint32 * (len(words) + 8))(); arr[0] = length; arr[1] = 0x40000000
    for i, w in enumerate(words): arr[2 + i] = w
    base = ctypes.addressof(arr) + 8
    buf = ctypes.create_string_buffer(4096)
    getattr(L, fn).restype = ctypes.c_uint64
    n = getattr(L, fn)(buf, ctypes.c_uint64(4096), ctypes.c_void_p(base), ctypes.c_uint64(offset), ctypes.c_uint64(length))
    return buf.value.decode(), n

# known-answer values held by the reference's in-tree unit tests (gaba.c:4297-4522), restated as data
CIGAR_KAT_FW = [([0x55555555], 0, 32, "16M"), ([0x55555555, 0x55555555], 0, 64, "32M"),
                ([0x55555555] * 4, 0, 128, "64M"), ([0x55550000, 0x55555555, 0x55555555, 0x55555555], 16, 112, "56M"),
                ([0x55550555], 0, 32, "6M4D8M"), ([0x5555f555], 0, 32, "6M4I8M"),
                ([0xaaaa0555], 0, 33, "6M5D8M"), ([0xaaabf555], 0, 33, "6M5I8M")]
CIGAR_KAT_RV = [([0x55555555], 0, 32, "16M"), ([0x55555555, 0x55555555], 0, 64, "32M"),
                ([0x55555000, 0x55555555, 0x55555555, 0x55555555], 12, 116, "58M"), ([0x55], 0, 8, "4M"),
                ([0x55555000, 0x55555555, 0x55555555, 0x55], 12, 92, "46M"),
                ([0x55550555], 0, 32, "8M4D6M"), ([0x5555f555], 0, 32, "8M4I6M"),
                ([0xaaaa0555], 0, 33, "8M5D6M"), ([0xaaabf555], 0, 33, "8M5I6M"),
                ([0xaaabf555, 0xaaaa0556], 0, 65, "8M5D5M1I8M5I6M"),
                ([0xaaabf555, 0xaaaa0556, 0xaaaaaaaa], 0, 65, "8M5D5M1I8M5I6M"),
                ([0xaaabf554, 0xaaaa0556, 0xaaaaaaaa], 0, 65, "8M5D5M1I8M5I5M2D")]

def test_cigar_known_answers():
    L = ctypes.CDLL(os.path.join(G.ROOT, 'oracle', 'liboracle.so'))
    for words, ofs, ln, want in CIGAR_KAT_FW:
        assert _cigar(L, 'og_dump_cigar_forward', words, ofs, ln)[0] == want
    for words, ofs, ln, want in CIGAR_KAT_RV:
        assert _cigar(L, 'og_dump_cigar_reverse', words, ofs, ln)[0] == want

@pytest.mark.skipif(not G.Reference.available(), reason="oracle/_ref not built")
def test_cigar_matches_reference_on_random_paths():
    L = ctypes.CDLL(os.path.join(G.ROOT, 'oracle', 'liboracle.so')); R = ctypes.CDLL(G.Reference.PATH)
    rng = np.random.default_rng(5)
    ora = G.Oracle(**G.PACBIO)
    for j in G.random_jobs(31, 60, max_len=1500):
        d = ora.extend(*j)
        if d['traced'] != 1 or d['plen'] == 0: continue
        for s in d['segs']:
            ln = s[4] + s[5]
            a = _cigar(L, 'og_dump_cigar_reverse', d['path'] + [0, 0], s[6], ln)
            b = _cigar(R, 'shim_dump_cigar_reverse', d['path'] + [0, 0], s[6], ln)
            assert a == b
