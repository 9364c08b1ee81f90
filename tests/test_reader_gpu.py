"""GPU: the device reader (mm_device.hpp K0r: record scanning over the raw text in HBM -- newline / delimiter masks, prefix scans, a record table; K0: base conversion
and 2-bit packing) against the ORACLE's reader (oracle/ora_mm.c om_read_fasta_ex, bseq_read_fasta restated and pinned on the compiled reference's output for these
files, tests/golden/parse_cases.json.gz): names, base codes, qualities and comments of every record, on the 14 oddly formatted files of the reader tests (wrapped lines,
CR LF, lower case, IUPAC letters, blank lines, no final newline, tabs and spaces in headers, empty records, a delimiter in the middle of a sequence line, FASTQ with
'@' / '+' in the qualities, wrapped, short qualities -> rejected) and on generated sets -- with the text cut into stretches of 64 bytes .. 64 KB as well, so that
records straddle stretch boundaries and outgrow a stretch.  The command-line program reads through the same code, so every SAM golden runs through it too."""
import ctypes, gzip, os, tempfile
import numpy as np, pytest
import mmlib as M
from golden.make_parse_golden import make_parse_inputs, CASES

pytestmark = pytest.mark.gpu

class Seqs(ctypes.Structure): _fields_ = [('a', ctypes.POINTER(M.OmSeq)), ('n', ctypes.c_uint64)]

def _oracle(path, keep_qual):
    OL = ctypes.CDLL(os.path.join(M.ROOT, 'oracle', 'liboracle.so')); OL.om_read_fasta_ex.restype = Seqs
    want = OL.om_read_fasta_ex(path.encode(), keep_qual, 1)
    if ctypes.c_int.in_dll(OL, 'om_read_error').value: return None
    out = []
    for i in range(want.n):
        q = want.a[i]
        if q.l_seq < 1: continue          # -L 1: empty records are dropped (minialign.c:2077)
        out.append((q.name[:q.l_name], bytes(ctypes.string_at(q.seq, q.l_seq)), (q.qual or b'') if keep_qual else b'', q.comment))
    return out

@pytest.fixture(scope='module')
def ctx():
    from minialign_amd import multi
    os.environ.setdefault('MM_SLAB_GB', '4')
    L = multi.load_library(); assert L.mm_set_device(0) == 0
    L.mm_reads_scan.restype = ctypes.c_void_p; L.mm_reads_scan.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    L.mm_reads_codes.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]; L.mm_reads_codes.restype = ctypes.c_uint32
    L.mm_reads_qual.argtypes = [ctypes.c_void_p, ctypes.c_uint32]; L.mm_reads_qual.restype = ctypes.c_char_p
    L.mm_reads_comment.argtypes = [ctypes.c_void_p, ctypes.c_uint32]; L.mm_reads_comment.restype = ctypes.c_char_p
    L.mm_reads_free.argtypes = [ctypes.c_void_p]
    with tempfile.TemporaryDirectory() as d:
        ref, files = make_parse_inputs(d)
        o = ctypes.c_void_p(L.mm_opt_init()); argv = (ctypes.c_char_p * 3)(b'minialign', b'-xpacbio', ref.encode()); fl = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
        assert L.mm_opt_parse(o, 3, argv, fl, 8, ctypes.byref(nf)) == 0
        mi = ctypes.c_void_p(L.mm_idx_gen(o, ref.encode())); al = ctypes.c_void_p(L.mm_align_init(o, mi)); assert al
        yield L, al, d, files

def _scan(L, al, path, keep_qual, chunk=None):
    if chunk is None: os.environ.pop('MM_CHUNK_BYTES', None)
    else: os.environ['MM_CHUNK_BYTES'] = str(chunk)
    hs = ctypes.c_uint64(0)
    r = ctypes.c_void_p(L.mm_reads_scan(al, path.encode(), keep_qual, 1, ctypes.byref(hs)))
    os.environ.pop('MM_CHUNK_BYTES', None)
    if not r: return None, 0
    out = []
    for i in range(L.mm_reads_count(r)):
        n = L.mm_reads_codes(r, i, None, 0); buf = np.zeros(n + 1, dtype=np.uint8); L.mm_reads_codes(r, i, buf.ctypes.data_as(ctypes.c_void_p), n)
        out.append((L.mm_reads_name(r, i), bytes(buf[:n]), L.mm_reads_qual(r, i) if keep_qual else b'', L.mm_reads_comment(r, i)))
    L.mm_reads_free(r)
    return out, hs.value

@pytest.mark.parametrize('keep_qual', [0, 1])
@pytest.mark.parametrize('name', sorted(CASES))
def test_device_reader_matches_the_oracle_reader_on_oddly_formatted_files(ctx, name, keep_qual):
    L, al, d, files = ctx
    want = _oracle(files[name], keep_qual)
    for chunk in (None, 64, 128):
        got, host_scanned = _scan(L, al, files[name], keep_qual, chunk)
        if want is None:
            assert got is None, (name, chunk)          # the reference gives up on the run (exit 1): so does the reader
            continue
        assert got == want, (name, keep_qual, chunk)
        if name.startswith('fa_'): assert host_scanned == 0          # FASTA never leaves the device
        if name in ('fq_plain', 'fq_at_qual', 'fq_noeol') and chunk is None: assert host_scanned == 0, name      # four lines per record: scanned on the device

def test_device_reader_on_generated_sets_and_across_stretch_boundaries(ctx):
    L, al, d, files = ctx
    ref = os.path.join(d, 'g.fa'); M.gensim('genome', 8811, 500000, 4, 0.1, out=ref)
    fa = os.path.join(d, 'r.fa'); fq = os.path.join(d, 'r.fq'); M.gensim('reads', 8812, ref, 3.0, 'pacbio', 'fa', 5000, 2500, out=fa); M.gensim('reads', 8813, ref, 2.0, 'ont', 'fq', out=fq)
    wrapped = os.path.join(d, 'w.fa'); gz = os.path.join(d, 'r.fa.gz'); gzt = os.path.join(d, 'r.fa.gz.txt'); wfq = os.path.join(d, 'w.fq')
    with open(fa, 'rb') as f, open(wrapped, 'wb') as g:
        for k, line in enumerate(f):
            if line.startswith(b'>'): g.write(line.rstrip(b'\n') + b' comment %d\tx\r\n' % k); continue
            sq = bytearray(line.rstrip(b'\n'))
            for i in range(7, len(sq), 97): sq[i] = b'NnRYacgtu-'[(i + k) % 10]
            for i in range(0, len(sq), 61): g.write(bytes(sq[i:i + 61]) + b'\r\n')
    blob = open(fa, 'rb').read(); cut = blob.rfind(b'>', 0, len(blob) // 2)
    with open(gz, 'wb') as g: g.write(gzip.compress(blob[:cut]) + gzip.compress(blob[cut:]))
    open(gzt, 'wb').write(blob)
    # FASTQ with wrapped sequence and quality lines (the sequential grammar: host reader), cut into stretches that end inside records
    with open(fq, 'rb') as f, open(wfq, 'wb') as g:
        lines = f.read().split(b'\n')
        for i in range(0, len(lines) - 3, 4):
            h, sq, p, ql = lines[i:i + 4]
            g.write(h + b'\n' + b'\n'.join(sq[j:j + 80] for j in range(0, len(sq), 80)) + b'\n' + p + b'\n' + b'\n'.join(ql[j:j + 70] for j in range(0, len(ql), 70)) + b'\n')
    for path, ora_path, kq, expect_host in ((fa, fa, 0, False), (ref, ref, 0, False), (wrapped, wrapped, 0, False), (gz, gzt, 0, False), (fq, fq, 0, False), (fq, fq, 1, False), (wfq, wfq, 1, True)):
        want = _oracle(ora_path, kq); assert want
        for chunk in (None, 1 << 16, 4096):
            got, host_scanned = _scan(L, al, path, kq, chunk)
            assert got is not None and len(got) == len(want), (path, chunk)
            assert got == want, (path, kq, chunk)
            assert (host_scanned > 0) == expect_host, (path, chunk, host_scanned)
