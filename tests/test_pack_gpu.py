"""GPU: reads packed on the device from the text of their file (K0: mm_text_codes_kernel + mm_codes_pack_kernel -- bseq_read_fasta's base conversion with the table
of minialign.c:223-229 and the 2-bit / N-mask packing) against the host packing over the parser's base codes, arena word by arena word: on the oddly formatted
files of the reader tests (wrapped lines, CR LF, lower case, IUPAC letters and punctuation, blank lines, no final newline, a delimiter inside a line, FASTQ with
'@' in the qualities ...), on generated FASTA / FASTQ, wrapped and gzip-compressed.  The command-line program packs this way, so every SAM parity test runs
through it as well."""
import ctypes, gzip, os, tempfile
import numpy as np, pytest
import mmlib as M
from golden.make_parse_golden import make_parse_inputs, CASES

pytestmark = pytest.mark.gpu

@pytest.fixture(scope='module')
def ctx():
    from minialign_amd import multi
    os.environ.setdefault('MM_SLAB_GB', '4')
    L = multi.load_library(); assert L.mm_set_device(0) == 0
    L.mm_reads_load_text.restype = ctypes.c_void_p; L.mm_pack_check.restype = ctypes.c_int64; L.mm_pack_check.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.mm_pack_fetch.restype = ctypes.c_int64; L.mm_pack_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32]
    with tempfile.TemporaryDirectory() as d:
        ref, files = make_parse_inputs(d)
        o = ctypes.c_void_p(L.mm_opt_init()); argv = (ctypes.c_char_p * 3)(b'minialign', b'-xpacbio', ref.encode()); fl = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
        assert L.mm_opt_parse(o, 3, argv, fl, 8, ctypes.byref(nf)) == 0
        mi = ctypes.c_void_p(L.mm_idx_gen(o, ref.encode())); al = ctypes.c_void_p(L.mm_align_init(o, mi)); assert al
        yield L, al, d, files

def _oracle_reads(path):
    """names and base codes from the oracle's reader (oracle/ora_mm.c om_read_fasta_ex: bseq_read_fasta restated, pinned on the compiled reference's output for
    these files, tests/golden/parse_cases.json.gz); None when it rejects the file"""
    OL = ctypes.CDLL(os.path.join(M.ROOT, 'oracle', 'liboracle.so'))
    class Seqs(ctypes.Structure): _fields_ = [('a', ctypes.POINTER(M.OmSeq)), ('n', ctypes.c_uint64)]
    OL.om_read_fasta_ex.restype = Seqs
    if path.endswith('.gz'):
        import tempfile as T
        with T.NamedTemporaryFile(suffix='.txt') as f:
            f.write(gzip.decompress(open(path, 'rb').read())); f.flush()
            return _oracle_reads(f.name)
    want = OL.om_read_fasta_ex(path.encode(), 0, 0)
    if ctypes.c_int.in_dll(OL, 'om_read_error').value: return None
    return [(want.a[i].name[:want.a[i].l_name], np.frombuffer(ctypes.string_at(want.a[i].seq, want.a[i].l_seq), dtype=np.uint8)) for i in range(want.n) if want.a[i].l_seq >= 1]

def _check(L, al, path):
    """(reads, differing arena words device vs the library's own host packing, device arena == the oracle reader's codes)"""
    r = ctypes.c_void_p(L.mm_reads_load_text(path.encode()))
    if not r: return None
    n = L.mm_reads_count(r); rc = L.mm_pack_check(al, r)
    want = _oracle_reads(path)
    total = int(L.mm_reads_bases(r, 0, n)); codes = np.zeros(total + 64, dtype=np.uint8); lens = np.zeros(n + 1, dtype=np.uint32)
    got_n = L.mm_pack_fetch(al, r, codes.ctypes.data_as(ctypes.c_void_p), len(codes), lens.ctypes.data_as(ctypes.c_void_p), n)
    same = want is not None and got_n == len(want) == n and [int(x) for x in lens[:n]] == [len(sq) for _, sq in want] and bytes(codes[:total]) == b''.join(sq.tobytes() for _, sq in want) \
        and [L.mm_reads_name(r, i) for i in range(n)] == [nm for nm, _ in want]
    L.mm_reads_free(r)
    return n, rc, same

@pytest.mark.parametrize('name', sorted(CASES))
def test_oddly_formatted_files_pack_the_same_on_the_device(ctx, name):
    L, al, d, files = ctx
    res = _check(L, al, files[name])
    if res is None: pytest.skip('the reader gives up on this file (as the reference does)')
    n, rc, same = res
    assert n > 0 and rc == 0 and same, (name, n, rc, same)          # device arena == host packing == the ORACLE reader's base codes

def test_generated_sets_pack_the_same_on_the_device(ctx):
    L, al, d, files = ctx
    ref = os.path.join(d, 'g.fa'); M.gensim('genome', 8801, 400000, 3, 0.1, out=ref)
    fa = os.path.join(d, 'r.fa'); fq = os.path.join(d, 'r.fq'); M.gensim('reads', 8802, ref, 3.0, 'pacbio', 'fa', 5000, 2500, out=fa); M.gensim('reads', 8803, ref, 2.0, 'ont', 'fq', out=fq)
    # the same reads wrapped at 61 columns with CR LF line ends and some N / lower case / IUPAC letters, and gzip-compressed in two members
    wrapped = os.path.join(d, 'w.fa'); gz = os.path.join(d, 'r.fa.gz')
    with open(fa, 'rb') as f, open(wrapped, 'wb') as g:
        for k, line in enumerate(f):
            if line.startswith(b'>'): g.write(line); continue
            sq = bytearray(line.rstrip(b'\n'))
            for i in range(7, len(sq), 97): sq[i] = b'NnRYacgtu-'[(i + k) % 10]
            for i in range(0, len(sq), 61): g.write(bytes(sq[i:i + 61]) + b'\r\n')
    blob = open(fa, 'rb').read(); cut = blob.rfind(b'>', 0, len(blob) // 2)
    with open(gz, 'wb') as g: g.write(gzip.compress(blob[:cut]) + gzip.compress(blob[cut:]))
    for p in (fa, fq, wrapped, gz, ref):
        res = _check(L, al, p); assert res is not None
        assert res[0] > 0 and res[1] == 0 and res[2], (p, res)
