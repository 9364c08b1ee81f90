"""CPU tests (no GPU): the C-ABI library loads and exports every symbol the headers declare; the multi-GPU host logic
(sharding, settling the carried value between ranks, merge in rank order) under 2- and 3-process gloo groups."""
import ctypes, os, re, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def _declared(header):
    txt = open(os.path.join(ROOT, 'include', header)).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    txt = re.sub(r'^[ \t]*#[ \t]*define.*$', '', txt, flags=re.M)          # function-like macros are not symbols
    names = set()
    for m in re.finditer(r'\b((?:gaba|mm)_[a-z0-9_]+)\s*\(', txt):
        names.add(m.group(1))
    return names

@pytest.fixture(autouse=True)
def _index_on_the_host(monkeypatch):
    """no GPU in these tests: the index is built by the host threads (MM_HOST_INDEX; the default is the device build of mm_index.hpp, tests/test_index_gpu.py)"""
    monkeypatch.setenv('MM_HOST_INDEX', '1')

def test_library_exports_every_declared_symbol():
    lib = os.path.join(ROOT, 'minialign_amd', 'libminialign_amd.so')
    assert os.path.exists(lib), 'run __graft_entry__.build() first'
    L = ctypes.CDLL(lib)          # loading needs no GPU; no compute entry point is called here
    missing = [n for h in ('gaba.h', 'minialign.h') for n in sorted(_declared(h)) if not hasattr(L, n)]
    assert not missing, 'declared in include/*.h but not exported: %r' % missing

def test_shard_bounds_cover_everything_once():
    from minialign_amd.multi import shard_bounds
    for n in (0, 1, 7, 8, 16, 1000, 22308):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


class FakeLib:
    """a stand-in for libminialign_amd.so with the same carried-value structure as the mapper (minialign.c:3864): read i tests `apos0 >= carried`, its records
    and the reference it loads last depend on that decision, the carried value behind it is the length of that reference (or unchanged when it loads none).
    Lets the settling protocol of minialign_amd/multi.py run under a gloo group without a GPU."""
    NIL = 0xffffffff
    def __init__(self, n, seed):
        import random
        rnd = random.Random(seed)
        self.lens = [1000 * (i + 1) for i in range(12)]
        # (read names repeat and records do not start with them: the splice must not depend on the text, see mm_head_offset)
        self.reads = [dict(name=b'r%d' % (i % 7), apos0=(self.NIL if rnd.random() < 0.1 else rnd.randrange(0, 13000)), cond0=int(rnd.random() < 0.05),
                           r_no=(self.NIL if rnd.random() < 0.15 else rnd.randrange(12)), r_yes=rnd.randrange(12), lines=rnd.randrange(0, 3)) for i in range(n)]
        self.carry = 0; self.head = []; self.head_in = 0; self.offs = [0]
    def _one(self, i, cur):
        r = self.reads[i]
        dec = r['apos0'] != self.NIL and not r['cond0'] and r['apos0'] >= cur
        rid = r['r_yes'] if dec else r['r_no']
        text = b''.join(b'a score=%d\ns %s\t%d\t%d\t%d\n' % (i, r['name'], j, int(dec), rid) for j in range(r['lines']))
        return text, rid
    def mm_align_set_carry(self, al, v): self.carry = v
    def mm_align_get_carry(self, al): return self.carry
    def mm_head_offset(self, al, i): return self.offs[i] if i < len(self.offs) else 0xffffffffffffffff
    def mm_map_reads(self, al, reads, first, n, lanes, cb, opaque):
        import ctypes
        cur = self.carry; self.head = []; self.head_in = cur; out = []
        for i in range(first, first + n):
            text, rid = self._one(i, cur)
            self.head.append((self.reads[i]['apos0'], self.reads[i]['cond0'], cur, rid)); out.append(text)
            if rid != self.NIL: cur = self.lens[rid]
        self.carry = cur
        self.offs = [0]
        for t in out: self.offs.append(self.offs[-1] + len(t))
        self.offs = self.offs[:4097]          # the first 4 096 reads (and the end of a shorter stream), as the library records them
        for k in range(0, len(out), 50):
            piece = b''.join(out[k:k + 50]); buf = ctypes.create_string_buffer(piece, len(piece))
            cb(None, k // 50, ctypes.addressof(buf), len(piece))
        return 0
    def mm_carry_check(self, al, truth, fa):
        cur = truth
        if cur == self.head_in: return 0
        for i, (apos0, cond0, used, rid) in enumerate(self.head[:4096]):
            if apos0 != self.NIL and not cond0 and ((apos0 >= used) != (apos0 >= cur)):
                fa._obj.value = i; return 1
            if rid != self.NIL: return 0
        return 2
    def mm_carry_after(self, al, i):
        cur = self.head_in
        if i >= min(len(self.head), 4096): return 0xffffffff
        for (_, _, _, rid) in self.head[:i + 1]:
            if rid != self.NIL: cur = self.lens[rid]
        return cur

def _settle_worker(rank, world, port, n, seed, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from minialign_amd import multi
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    L = FakeLib(n, seed); lo, hi = multi.shard_bounds(n, rank, world)
    sm = multi.ShardMapper(L, None, None, lo, hi - lo, guess=7777 if rank else 0).map()
    sm.settle(dist, rank, world, 0, None)
    parts = [None] * world
    dist.all_gather_object(parts, sm.col.text())
    dist.barrier(); dist.destroy_process_group()
    q.put((rank, b''.join(parts), sm.stats))

@pytest.mark.parametrize('world,n,seed', [(2, 700, 1), (3, 1000, 2), (3, 130, 3), (2, 3, 4)])
def test_sharded_settling_protocol_under_gloo(world, n, seed):
    """minialign_amd/multi.py over a gloo group (CPU): shards started from a guess, end values exchanged, heads re-mapped in doubling windows or whole shards
    re-mapped, until the merged text is what one stream over all reads gives"""
    import ctypes, torch.multiprocessing as mp
    one = FakeLib(n, seed); got = []
    one.mm_map_reads(None, None, 0, n, 1, lambda o, k, p, ln: got.append(ctypes.string_at(p, ln)) or 0, None)
    want = b''.join(got)
    ctx = mp.get_context('spawn'); q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    ps = [ctx.Process(target=_settle_worker, args=(r, world, port, n, seed, q)) for r in range(world)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps: p.join(60)
    assert all(r[1] == want for r in res), [r[2] for r in res]
    assert sum(r[2]['checks'] for r in res) > 0


def test_product_cigar_known_answers():
    """the product's host CIGAR printers against the reference's in-tree known answers (gaba.c:4297-4522), via test_oracle_gaba's table"""
    from test_oracle_gaba import CIGAR_KAT_FW, CIGAR_KAT_RV, _cigar
    L = ctypes.CDLL(os.path.join(ROOT, 'minialign_amd', 'libminialign_amd.so'))
    for words, ofs, ln, want in CIGAR_KAT_FW:
        assert _cigar(L, 'gaba_dump_cigar_forward', words, ofs, ln)[0] == want
    for words, ofs, ln, want in CIGAR_KAT_RV:
        assert _cigar(L, 'gaba_dump_cigar_reverse', words, ofs, ln)[0] == want

def test_product_cigar_printer_callbacks_match_dump():
    """gaba_print_cigar_{forward,reverse} (gaba.h:394-406) drive a caller's printer with the same runs gaba_dump_cigar_* writes"""
    import numpy as np
    from test_oracle_gaba import CIGAR_KAT_FW, CIGAR_KAT_RV, _cigar
    L = ctypes.CDLL(os.path.join(ROOT, 'minialign_amd', 'libminialign_amd.so'))
    PR = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char)
    for name, table in (('forward', CIGAR_KAT_FW), ('reverse', CIGAR_KAT_RV)):
        fn = getattr(L, 'gaba_print_cigar_' + name); fn.restype = ctypes.c_uint64
        for words, ofs, ln, want in table:
            got = []
            def pr(fp, n, c, got=got):
                s = b'%d%s' % (n, c); got.append(s); return len(s)
            buf = np.zeros(len(words) + 8, dtype=np.uint32); buf[2:2 + len(words)] = words      # two header words in front, as in gaba_alignment_s
            base = buf.ctypes.data + 8
            clen = fn(PR(pr), None, ctypes.c_void_p(base), ctypes.c_uint64(ofs), ctypes.c_uint64(ln))
            assert b''.join(got).decode() == want and clen == len(want)


def test_host_index_matches_oracle_and_survives_dump_and_load(tmp_path):
    """mm_idx_gen of the product is host code (no GPU): occurrence thresholds and value lists against the oracle's index, then the same
    answers from a block written by `minialign -d` (mm_idx_dump) and read back (mm_idx_load); a cut-off file is refused, not half-loaded."""
    import subprocess, numpy as np, mmlib as M
    ref = str(tmp_path / 'ref.fa'); ref2 = str(tmp_path / 'ref2.fa'); mai = str(tmp_path / 'idx.mai')
    M.gensim('genome', 901, 200000, 3, 0.15, out=ref); M.gensim('genome', 902, 50000, 1, 0.0, out=ref2)
    L = ctypes.CDLL(os.path.join(ROOT, 'minialign_amd', 'libminialign_amd.so'))
    for f in ('mm_opt_init', 'mm_idx_gen', 'mm_idx_load'): getattr(L, f).restype = ctypes.c_void_p
    libc = ctypes.CDLL(None); libc.fopen.restype = ctypes.c_void_p; libc.fclose.argtypes = [ctypes.c_void_p]
    o = ctypes.c_void_p(L.mm_opt_init())
    argv = (ctypes.c_char_p * 3)(b'minialign', b'-xpacbio', ref.encode()); files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
    assert L.mm_opt_parse(o, 3, argv, files, 8, ctypes.byref(nf)) == 0
    mi = ctypes.c_void_p(L.mm_idx_gen(o, ref.encode())); assert mi
    refseq = M.read_fasta(ref); ora = M.OracleMM('pacbio', refseq)
    keys = [int(m) >> 8 for _, q in refseq for m in ora.sketch(q)[::37]] + [12345, 1 << 29]
    buf = (ctypes.c_uint64 * 65536)()
    def answers(h):
        return [L.mm_idx_n_seq(h)] + [L.mm_idx_occ(h, i) for i in range(3)] + [[int(buf[i]) for i in range(L.mm_idx_get(h, ctypes.c_uint64(k), buf, 65536))] for k in keys]
    want = answers(mi)
    assert want[1:4] == ora.occ()[:3]
    assert want[4:] == [[int(v) for v in ora.idx_get(k)] for k in keys]
    assert sum(len(v) for v in want[4:]) > len(keys) // 2
    # -d: two reference files -> two blocks in one file; no GPU is touched on this path
    cli = os.path.join(ROOT, 'minialign_amd', 'minialign')
    r = subprocess.run([cli, '-xpacbio', '-d', mai, ref, ref2], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and r.stdout == b'', r.stderr.decode()
    fp = ctypes.c_void_p(libc.fopen(mai.encode(), b'rb')); eof = ctypes.c_int(0)
    m1 = ctypes.c_void_p(L.mm_idx_load(fp, ctypes.byref(eof))); assert m1 and eof.value == 0
    assert answers(m1) == want
    m2 = ctypes.c_void_p(L.mm_idx_load(fp, ctypes.byref(eof))); assert m2 and L.mm_idx_n_seq(m2) == 1
    assert not L.mm_idx_load(fp, ctypes.byref(eof)) and eof.value == 1
    libc.fclose(fp)
    # a truncated block and a foreign file
    data = open(mai, 'rb').read()
    for name, blob in (('cut.mai', data[:len(data) // 3]), ('other.mai', b'MAI\x08' + data[4:])):
        p = str(tmp_path / name); open(p, 'wb').write(blob)
        fp = ctypes.c_void_p(libc.fopen(p.encode(), b'rb'))
        assert not L.mm_idx_load(fp, ctypes.byref(eof)) and eof.value == 0
        libc.fclose(fp)
    # gzip-compressed input (one member, and two members back to back) builds the same index; a cut-off stream is an error
    import gzip
    text = open(ref, 'rb').read()
    open(ref + '.gz', 'wb').write(gzip.compress(text)); open(ref + '.2.gz', 'wb').write(gzip.compress(text[:70001]) + gzip.compress(text[70001:]))
    open(str(tmp_path / 'cut.fa.gz'), 'wb').write(gzip.compress(text)[:5000])
    plain = str(tmp_path / 'plain.mai'); subprocess.run([cli, '-xpacbio', '-d', plain, ref], check=True, stderr=subprocess.DEVNULL)
    for z in (ref + '.gz', ref + '.2.gz'):
        subprocess.run([cli, '-xpacbio', '-d', mai, z], check=True, stderr=subprocess.DEVNULL)
        assert open(mai, 'rb').read() == open(plain, 'rb').read()
    assert subprocess.run([cli, '-xpacbio', '-d', mai, str(tmp_path / 'cut.fa.gz')], stderr=subprocess.DEVNULL).returncode == 1
    for h in (mi, m1, m2): L.mm_idx_destroy(h)
    L.mm_opt_destroy(o)


def test_host_index_of_circular_references_matches_oracle(tmp_path):
    """-c: minimizers of the windows that span the origin of a circular sequence, with the positions the reference's stream decoder gives them
    (minialign.c:2438-2444, 2831-2835) -- product host index against the oracle's, all sequences circular and by name"""
    import numpy as np, mmlib as M
    sys_path = os.path.join(ROOT, 'tests', 'golden')
    import sys; sys.path.insert(0, sys_path)
    from make_circ_golden import make_circ_inputs
    ref, _ = make_circ_inputs(str(tmp_path))
    L = ctypes.CDLL(os.path.join(ROOT, 'minialign_amd', 'libminialign_amd.so'))
    for f in ('mm_opt_init', 'mm_idx_gen'): getattr(L, f).restype = ctypes.c_void_p
    refseq = M.read_fasta(ref)
    for copt, names in ((b'-c*', None), (b'-cplasmid', b'plasmid')):
        o = ctypes.c_void_p(L.mm_opt_init())
        argv = (ctypes.c_char_p * 4)(b'minialign', b'-xpacbio', copt, ref.encode()); files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
        assert L.mm_opt_parse(o, 4, argv, files, 8, ctypes.byref(nf)) == 0 and nf.value == 1
        mi = ctypes.c_void_p(L.mm_idx_gen(o, ref.encode())); assert mi
        ora = M.OracleMM('pacbio', refseq, circ=(names if names else b''))
        assert [L.mm_idx_occ(mi, i) for i in range(3)] == ora.occ()[:3]
        buf = (ctypes.c_uint64 * 65536)()
        n_wrap = 0
        for name, q in refseq:
            dbl = np.concatenate([q, q[:40]])                  # k-mers across the origin are looked up too
            for m in ora.sketch(dbl)[::3]:
                k = int(m) >> 8
                got = [int(buf[i]) for i in range(L.mm_idx_get(mi, ctypes.c_uint64(k), buf, 65536))]
                want = [int(v) for v in ora.idx_get(k)]
                assert got == want
                n_wrap += sum(1 for v in want if (v & 0xffffffff) + 15 > len(q) and (v >> 33) == [n for n, _ in refseq].index(name))
        assert n_wrap > 0
        L.mm_idx_destroy(mi); L.mm_opt_destroy(o)


def test_reader_matches_oracle_on_oddly_formatted_files(tmp_path):
    """FASTA / FASTQ reader (bseq_read_fasta): wrapped lines, CRLF (a CR in a sequence line reads as a base), lower case, IUPAC letters, blank lines, no final
    newline, tabs and spaces in headers, empty records, a delimiter in the middle of a sequence line, FASTQ with '@' qualities / wrapped / short qualities.
    The product reads each file through `minialign -d` (host only); names and bases must equal the oracle reader's, which is pinned on the compiled
    reference's output for the same files (tests/golden/parse_cases.json.gz, tests/test_oracle_mm.py)."""
    import struct, subprocess, sys, numpy as np, mmlib as M
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from make_parse_golden import make_parse_inputs
    _, files = make_parse_inputs(str(tmp_path))
    OL = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'liboracle.so'))
    class Seqs(ctypes.Structure): _fields_ = [('a', ctypes.POINTER(M.OmSeq)), ('n', ctypes.c_uint64)]
    OL.om_read_fasta_ex.restype = Seqs
    cli = os.path.join(ROOT, 'minialign_amd', 'minialign')
    n_err = 0
    for name, path in files.items():
        want = OL.om_read_fasta_ex(path.encode(), 0, 0)
        err = ctypes.c_int.in_dll(OL, 'om_read_error').value
        recs = [(want.a[i].name[:want.a[i].l_name], bytes(ctypes.string_at(want.a[i].seq, want.a[i].l_seq))) for i in range(want.n)]
        mai = str(tmp_path / (name + '.mai'))
        r = subprocess.run([cli, '-xpacbio', '-d', mai, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        if err or not recs:
            assert r.returncode == 1, name; n_err += bool(err); continue
        assert r.returncode == 0, (name, r.stderr.decode())
        blob = open(mai, 'rb').read(); off = 4 + 4 * 12 + 8 * 4; got = []
        n_seq = struct.unpack_from('<Q', blob, 4 + 4 * 12)[0]
        for _ in range(n_seq):
            ln, ls, _c = struct.unpack_from('<QQQ', blob, off); off += 24
            got.append((blob[off:off + ln], blob[off + ln:off + ln + ls])); off += ln + ls
        assert got == recs, name
    assert n_err >= 1


C_CALLER = r'''
#include <stdio.h>
#include <stdlib.h>
#include "gaba.h"
#include "minialign.h"
/* what a libgaba user writes (INTEGRATION.md, per-call form) -- compiled and linked only, never run here */
static int printer(void *fp, int64_t len, char c) { return fprintf((FILE *)fp, "%ld%c", (long)len, c); }
int main(int argc, char **argv)
{
	struct gaba_params_s p; char cig[64]; uint8_t a[64] = { 0 }, b[64] = { 0 };
	if(argc < 100) { return 0; }
	p = (struct gaba_params_s){ .xdrop = 50 };
	gaba_t *ctx = gaba_init(&p);
	gaba_arena_t *ra = gaba_arena_upload(a, 64), *qa = gaba_arena_upload(b, 64);
	gaba_dp_t *dp = gaba_dp_init(ctx);
	struct gaba_section_s sa = gaba_build_section(0, a, 64), sb = gaba_build_section(1, b, 64);
	gaba_fill_t *f = gaba_dp_fill_root(dp, &sa, 0, &sb, 0, 0);
	gaba_pos_pair_t *m = gaba_dp_search_max(dp, f);
	gaba_alignment_t *r = gaba_dp_trace(dp, f, NULL);
	gaba_dump_cigar_forward(cig, sizeof(cig), r->path, 0, r->plen);
	gaba_print_cigar_reverse(printer, stdout, r->path, 0, r->plen);
	gaba_dp_res_free(dp, r); gaba_dp_flush(dp); gaba_dp_clean(dp); gaba_clean(ctx);
	(void)m; (void)ra; (void)qa; (void)argv;
	return mm_main(argc, argv);
}
'''


def test_headers_stand_alone_and_a_c_caller_links(tmp_path):
    """include/*.h are what a maintainer binds: each must compile on its own as C99 and as C++, and a C caller written against the reference's
    names must link against the library."""
    inc = os.path.join(ROOT, 'include')
    for h in sorted(os.listdir(inc)):
        for lang, std in (('c', '-std=c99'), ('c++', '-std=c++11')):
            r = subprocess.run(['gcc', std, '-Wall', '-Werror', '-fsyntax-only', '-x', lang, os.path.join(inc, h)], capture_output=True, text=True)
            assert r.returncode == 0, (h, lang, r.stderr[:2000])
    src = tmp_path / 'caller.c'; src.write_text(C_CALLER)
    lib = os.path.join(ROOT, 'minialign_amd')
    r = subprocess.run(['gcc', '-std=gnu99', '-Wall', '-I', inc, str(src), '-o', str(tmp_path / 'caller'), '-L', lib, '-lminialign_amd',
                        '-Wl,-rpath,' + lib, '-Wl,--allow-shlib-undefined'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[:3000]


def test_host_mm_sketch_matches_oracle_stream():
    """mm_sketch (minialign.c:2410) exported on the host: the stream words (hash << 8 | strand << 7 | index in the block of w) against the oracle's restatement,
    N runs included, and the decoded positions against the stream decoder's rule (minialign.c:2831-2835)"""
    import numpy as np, mmlib as M
    L = ctypes.CDLL(os.path.join(ROOT, 'minialign_amd', 'libminialign_amd.so'))
    ora = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'liboracle.so')); ora.om_sketch.restype = ctypes.c_uint64
    rng = np.random.default_rng(5)
    for (w, k, n) in ((10, 15, 5000), (5, 15, 3000), (10, 15, 14), (10, 15, 15), (10, 15, 40), (3, 7, 400), (16, 21, 6000)):
        seq = rng.integers(0, 4, n).astype(np.uint8)
        if n > 1000: seq[700:760] = 4; seq[n // 2] = 4
        want = np.zeros(n + 512, dtype=np.uint64)
        nw = ora.om_sketch(w, k, seq.ctypes.data_as(ctypes.c_void_p), n, want.ctypes.data_as(ctypes.c_void_p))
        got = np.zeros(n + 512, dtype=np.uint64); pos = np.zeros(n + 512, dtype=np.uint32)
        ng = L.mm_sketch(seq.ctypes.data_as(ctypes.c_void_p), n, w, k, got.ctypes.data_as(ctypes.c_void_p), pos.ctypes.data_as(ctypes.c_void_p), n + 512)
        assert ng == nw and (got[:ng] == want[:nw]).all(), (w, k, n)
        base = -w; prev = w; dec = []
        for v in want[:nw]:                       # the decoder of minialign.c:2831-2835
            lu = int(v) & 0x7f
            if lu <= prev: base += w
            prev = lu; dec.append(base + lu)
        assert dec == [int(p) for p in pos[:ng]], (w, k, n)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, 'oracle', '_ref', 'minialign')), reason='oracle/_ref not built (needs /root/reference)')
def test_option_words_are_walked_as_the_reference_walks_them(tmp_path):
    """mm_opt_parse_argv (minialign.c:5786-5812) and mm_opt_atoi (:5745): clustered boolean letters, an option behind them, required arguments that look like
    options, unparsable numbers -- same accept / reject decision as the compiled reference (`-d` runs: index only, no GPU)"""
    import mmlib as M
    ref = str(tmp_path / 'ref.fa'); M.gensim('genome', 31, 30000, 1, 0.0, out=ref)
    cli = os.path.join(ROOT, 'minialign_amd', 'minialign'); refbin = os.path.join(ROOT, 'oracle', '_ref', 'minialign')
    lines = [['-XA'], ['-PQ'], ['-Qxpacbio'], ['-XAk14'], ['-k15x'], ['-s', '-5'], ['-k', '15'], ['-PQw', '8'], ['-Z'], ['-k'], ['-w7', '-k13'], ['-r3,x'], ['-m0.3'], ['-m0,3x'],
             ['-f0.05,0.01'], ['-f0.01,0.05'], ['-Xh'], ['-c'], ['-c', 'ctg0000'], ['-xpacbio.ccs'], ['-xont.r9.4.1dsq'], ['-xnone'], ['-t', '2'], ['-t2x'], ['-Y60', '-p5']]
    for ln in lines:
        rc = []
        for exe in (refbin, cli):
            out = str(tmp_path / ('x_%s.mai' % ('r' if exe is refbin else 'o')))
            r = subprocess.run([exe] + ln + ['-d', out, ref], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            rc.append(r.returncode != 0)
        assert rc[0] == rc[1], (ln, rc)

def test_batches_are_cut_by_bases_lanes_and_the_longest_read(tmp_path):
    """batch_spans (mm_host.hip) through mm_batch_pack_all, host only: contiguous spans in input order that cover every read once; a set smaller than lanes x 300 Mb
    is cut into one batch per lane but not below 64 Mi bases; MM_LANES and MM_BATCH_BASES are read at call time"""
    import mmlib as M
    ref = str(tmp_path / 'ref.fa'); rd = str(tmp_path / 'rd.fa')
    M.gensim('genome', 8101, 3000000, 1, 0.0, out=ref); M.gensim('reads', 8102, ref, 24.0, 'pacbio', 'fa', 6000, 2500, out=rd)
    lens = []
    for l in open(rd):
        if l.startswith('>'): lens.append(0)
        else: lens[-1] += len(l.strip())
    total = sum(lens); assert 60e6 < total < 4 * (64 << 20)
    L = ctypes.CDLL(os.path.join(ROOT, 'minialign_amd', 'libminialign_amd.so'))
    L.mm_reads_load.restype = ctypes.c_void_p; L.mm_reads_load.argtypes = [ctypes.c_char_p]
    L.mm_reads_count.restype = ctypes.c_uint32; L.mm_reads_count.argtypes = [ctypes.c_void_p]
    L.mm_batch_pack_all.restype = ctypes.c_uint32; L.mm_batch_pack_all.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32]
    L.mm_batch_reads.restype = ctypes.c_uint32; L.mm_batch_reads.argtypes = [ctypes.c_void_p]
    L.mm_batch_free.argtypes = [ctypes.c_void_p]; L.mm_reads_free.argtypes = [ctypes.c_void_p]
    reads = L.mm_reads_load(rd.encode()); assert reads and L.mm_reads_count(reads) == len(lens)
    def spans(first=0, n=None, **env):
        old = {k: os.environ.get(k) for k in ('MM_LANES', 'MM_BATCH_BASES')}
        for k in old: os.environ.pop(k, None)
        os.environ.update({k: str(v) for k, v in env.items()})
        try:
            arr = (ctypes.c_void_p * 256)(); nb = L.mm_batch_pack_all(reads, first, len(lens) - first if n is None else n, arr, 256); assert 0 < nb <= 256
            cnt = [L.mm_batch_reads(arr[k]) for k in range(nb)]
            for k in range(nb): L.mm_batch_free(arr[k])
        finally:
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None: os.environ[k] = v
        out = []; at = first
        for c in cnt: out.append(sum(lens[at:at + c])); at += c
        assert at == first + (len(lens) - first if n is None else n)          # every read once, in order
        return out
    floor = 64 << 20
    b4 = spans()                                     # 4 lanes: total / 4 is below the floor of 64 Mi bases
    assert len(b4) == 2 and b4[0] <= floor < b4[0] + max(lens) and sum(b4) == total
    assert spans(MM_LANES=1) == [total]              # one lane: one batch
    small = spans(MM_BATCH_BASES=10000000)
    assert len(small) >= total // 10000000 and all(x <= 10000000 for x in small) and sum(small) == total
    part = spans(first=100, n=500, MM_BATCH_BASES=1000000)
    assert sum(part) == sum(lens[100:600]) and all(x <= 1000000 or x <= max(lens) for x in part)
    L.mm_reads_free(reads)

def test_part_loader_applies_the_readers_options(tmp_path):
    """mm_reads_load_part_opt: the host parser's part of a FASTQ file with the command line's reader options (-L drops short reads as bseq_read_fasta does, minialign.c:2077;
    -Q keeps qualities) -- what minialign_amd/multi.py uses for inputs that do not go through the text path, so that one command line gives one output whatever the format"""
    from minialign_amd import multi
    L = multi.load_library()
    L.mm_reads_qual.restype = ctypes.c_char_p; L.mm_reads_qual.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    fq = tmp_path / 'r.fq'; lens = [10, 100, 200, 50, 300, 70]
    fq.write_text(''.join('@r%d c%d\n%s\n+\n%s\n' % (i, i, 'ACGT' * (n // 4) + 'A' * (n % 4), 'I' * n) for i, n in enumerate(lens)))
    def load(args, part=0, n_parts=1):
        o = ctypes.c_void_p(L.mm_opt_init()); av = [b'minialign'] + args + [b'ref.fa', str(fq).encode()]
        argv = (ctypes.c_char_p * len(av))(*av); files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
        assert L.mm_opt_parse(o, len(av), argv, files, 8, ctypes.byref(nf)) == 0
        r = ctypes.c_void_p(L.mm_reads_load_part_opt(o, str(fq).encode(), part, n_parts)); assert r
        return r
    r = load([b'-L60', b'-Q'])
    assert L.mm_reads_count(r) == 4 and [L.mm_reads_name(r, i) for i in range(4)] == [b'r1', b'r2', b'r4', b'r5']
    assert L.mm_reads_qual(r, 1) == b'I' * 200
    L.mm_reads_free(r)
    r = load([])
    assert L.mm_reads_count(r) == 6 and L.mm_reads_qual(r, 0) in (b'', None)
    L.mm_reads_free(r)
    n = 0
    for p in range(3):
        r = load([b'-L60'], p, 3); n += L.mm_reads_count(r); L.mm_reads_free(r)
    assert n == 4          # the parts, in order, are the file
