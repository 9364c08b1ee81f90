"""GPU: reads packed on the device from the text of their file (K0: mm_text_codes_kernel + mm_codes_pack_kernel -- bseq_read_fasta's base conversion with the table
of minialign.c:223-229 and the 2-bit / N-mask packing) against the host packing over the parser's base codes, arena word by arena word: on the oddly formatted
files of the reader tests (wrapped lines, CR LF, lower case, IUPAC letters and punctuation, blank lines, no final newline, a delimiter inside a line, FASTQ with
'@' in the qualities ...), on generated FASTA / FASTQ, wrapped and gzip-compressed.  The command-line program packs this way, so every SAM parity test runs
through it as well."""
import ctypes, gzip, os, tempfile
import pytest
import mmlib as M
from golden.make_parse_golden import make_parse_inputs, CASES

pytestmark = pytest.mark.gpu

@pytest.fixture(scope='module')
def ctx():
    from minialign_amd import multi
    os.environ.setdefault('MM_SLAB_GB', '4')
    L = multi.load_library(); assert L.mm_set_device(0) == 0
    L.mm_reads_load_text.restype = ctypes.c_void_p; L.mm_pack_check.restype = ctypes.c_int64; L.mm_pack_check.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    with tempfile.TemporaryDirectory() as d:
        ref, files = make_parse_inputs(d)
        o = ctypes.c_void_p(L.mm_opt_init()); argv = (ctypes.c_char_p * 3)(b'minialign', b'-xpacbio', ref.encode()); fl = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
        assert L.mm_opt_parse(o, 3, argv, fl, 8, ctypes.byref(nf)) == 0
        mi = ctypes.c_void_p(L.mm_idx_gen(o, ref.encode())); al = ctypes.c_void_p(L.mm_align_init(o, mi)); assert al
        yield L, al, d, files

def _check(L, al, path):
    r = ctypes.c_void_p(L.mm_reads_load_text(path.encode()))
    if not r: return None
    n = L.mm_reads_count(r); rc = L.mm_pack_check(al, r); L.mm_reads_free(r)
    return n, rc

@pytest.mark.parametrize('name', sorted(CASES))
def test_oddly_formatted_files_pack_the_same_on_the_device(ctx, name):
    L, al, d, files = ctx
    res = _check(L, al, files[name])
    if res is None: pytest.skip('the reader gives up on this file (as the reference does)')
    n, rc = res
    assert n > 0 and rc == 0, (name, n, rc)

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
        assert res[0] > 0 and res[1] == 0, (p, res)
