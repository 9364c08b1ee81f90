"""One read set over N GPUs: one process per GPU (torch.distributed), reads sharded contiguously, the index replicated into every GPU's HBM, no
collective on the data path and no RCCL (BASELINE.json north_star): the process group is gloo and carries three integers per rank.
Output order = rank order, as the reference's drain keeps input order (mm_align_drain, minialign.c:4633-4645): every rank writes its own records to the
job's standard output when the rank in front of it has finished writing (a token handed from rank to rank), or -- MM_MULTI_PARTS=prefix -- to a file of its
own, `prefix.<rank>`, all ranks at once (`cat prefix.*` is the output).  Nothing is funnelled through one process.

The one thing reads share in the reference is `self->rlen` of its thread buffer (minialign.c:3864, DESIGN.md 5: the `apos >= rlen` test of
mm_search_load_pos reads the length of the reference sequence the *previous* read loaded last).  A shard therefore starts from the value the shard in
front of it ends with.  Every rank maps its shard right away with a guess for that value, the ranks then exchange what their shards ended with (one tiny
all_gather), and a rank whose guess was wrong asks the library what the true value changes (mm_carry_check, from the `apos`, the decision and the reference
of the first reads): almost always nothing; otherwise the head of the shard is mapped again from the first read that decides differently, in a window that
is doubled until the chain of values meets the old one again (mm_carry_after), and the new records replace the old ones at the byte offsets the library
recorded for them (mm_head_offset).  A changed end value travels on to the next rank in the next sweep; N - 1 sweeps at most, one in practice.

    python -m torch.distributed.run --nproc-per-node N -m minialign_amd.multi [-x preset ...] ref.fa reads.fa > out.sam

This module holds the host logic only; the hot path is libminialign_amd.so (HIP), reached through the C-ABI of include/minialign.h."""
import ctypes, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SINK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64)
NO_OFFSET = 0xffffffffffffffff          # mm_head_offset beyond what was recorded
NO_CARRY = 0xffffffff                   # mm_carry_after beyond the recorded head


def load_library(path=None):
    path = path or os.environ.get('MM_LIB_OVERRIDE') or os.path.join(ROOT, 'minialign_amd', 'libminialign_amd.so')
    if not os.path.exists(path):
        raise RuntimeError('libminialign_amd.so is not built (run __graft_entry__.build()); there is no CPU fallback')
    L = ctypes.CDLL(path)
    for f in ('mm_opt_init', 'mm_idx_gen', 'mm_align_init', 'mm_reads_load', 'mm_reads_load_text', 'mm_batch_pack'): getattr(L, f).restype = ctypes.c_void_p
    L.mm_reads_bases.restype = ctypes.c_uint64; L.mm_reads_name.restype = ctypes.c_char_p
    L.mm_reads_name.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    L.mm_reads_bases.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32]
    L.mm_reads_count.argtypes = [ctypes.c_void_p]; L.mm_reads_append.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.mm_reads_load_part.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32]; L.mm_reads_load_part.restype = ctypes.c_void_p
    L.mm_reads_load_part_opt.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32]; L.mm_reads_load_part_opt.restype = ctypes.c_void_p
    L.mm_align_get_carry.argtypes = [ctypes.c_void_p]; L.mm_align_get_carry.restype = ctypes.c_uint32
    L.mm_align_set_carry.argtypes = [ctypes.c_void_p, ctypes.c_uint32]; L.mm_align_set_carry.restype = None
    L.mm_carry_check.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
    L.mm_carry_after.argtypes = [ctypes.c_void_p, ctypes.c_uint32]; L.mm_carry_after.restype = ctypes.c_uint32
    L.mm_head_offset.argtypes = [ctypes.c_void_p, ctypes.c_uint32]; L.mm_head_offset.restype = ctypes.c_uint64
    L.mm_head_text_offset.argtypes = [ctypes.c_void_p, ctypes.c_uint32]; L.mm_head_text_offset.restype = ctypes.c_uint64
    L.mm_head_count.argtypes = [ctypes.c_void_p]; L.mm_head_count.restype = ctypes.c_uint32
    L.mm_map_text.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, SINK, ctypes.c_void_p]
    L.mm_map_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, SINK, ctypes.c_void_p]
    L.mm_map_reads.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int, SINK, ctypes.c_void_p]
    L.mm_map_packed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, SINK, ctypes.c_void_p]
    L.mm_batch_pack_all.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]; L.mm_batch_pack_all.restype = ctypes.c_uint32
    L.mm_batch_free.argtypes = [ctypes.c_void_p]; L.mm_batch_free.restype = None
    L.mm_idx_max_len.argtypes = [ctypes.c_void_p]; L.mm_idx_max_len.restype = ctypes.c_uint32
    L.mm_align_devices.argtypes = [ctypes.c_void_p]; L.mm_align_devices.restype = ctypes.c_int
    return L


def shard_bounds(n_units, rank, world):
    """contiguous, as-even-as-possible split of n_units; rank r gets [lo, hi)"""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Collector:
    """the sink of the streaming entries: keeps the text as the pieces it arrives in (keep = None: all of it; keep = N: pieces until N bytes are held) and
    counts everything"""
    def __init__(self, keep=None):
        self.keep = keep; self.pieces = []; self.kept = 0; self.total = 0
        self.cb = SINK(self._sink)
    def _sink(self, opaque, batch, ptr, n):
        self.total += n
        if self.keep is None or self.kept < self.keep:
            self.pieces.append(ctypes.string_at(ptr, n)); self.kept += n
        return 0
    def complete(self): return self.keep is None or self.total == self.kept
    def text(self): return b''.join(self.pieces)
    def head(self, n):
        """the first n bytes of the text (None when fewer were kept)"""
        if n > self.kept: return None
        out = []; left = n
        for p in self.pieces:
            if left <= 0: break
            out.append(p[:left]); left -= len(out[-1])
        return b''.join(out)
    def splice(self, lo, hi, new_pieces):
        """bytes [lo, hi) of the kept text replaced by new_pieces; the pieces outside the cut are not copied"""
        assert 0 <= lo <= hi <= self.kept
        out = []; at = 0; put = False
        for p in self.pieces:
            end = at + len(p)
            if end <= lo: out.append(p)
            elif at >= hi:
                if not put: out.extend(new_pieces); put = True
                out.append(p)
            else:
                if at < lo: out.append(p[:lo - at])
                if not put: out.extend(new_pieces); put = True
                if end > hi: out.append(p[hi - at:])
            at = end
        if not put: out.extend(new_pieces)
        d = sum(len(x) for x in new_pieces) - (hi - lo)
        self.pieces = out; self.kept += d; self.total += d


class ShardMapper:
    """maps one shard of a read set on this process' device and settles the carried value with the other ranks.  The shard is either reads [first, first + n) of a
    loaded (host-parsed) read set -- optionally as batches packed ahead of time --, or `text = (address, length)`: the FASTA / FASTQ text of the shard in host
    memory, which the library's device reader scans and packs (mm_map_text: the whole input path inside the call)"""
    def __init__(self, L, al, reads, first, n, lanes=0, packed=None, keep=None, guess=0, text=None):
        self.L, self.al, self.reads, self.first, self.n, self.lanes, self.packed, self.keep, self.guess, self.text = L, al, reads, first, n, lanes, packed, keep, guess, text
        self.col = None; self.carry_in = None; self.carry_out = None; self._stale = False; self.stats = dict(sweeps=0, checks=0, remapped_reads=0, full_remaps=0)

    def _map(self, first, n, carry_in, keep, packed=None, whole=False):
        """reads [first, first + n) of the shard (whole = all of it, the only form the packed batches and a text whose records are not known yet allow)"""
        col = Collector(keep)
        self.L.mm_align_set_carry(self.al, carry_in)
        if self.text is not None:
            lo, hi = (0, self.text[1]) if whole else (self.L.mm_head_text_offset(self.al, first), self.L.mm_head_text_offset(self.al, first + n))
            if lo == NO_OFFSET or hi == NO_OFFSET: raise RuntimeError('window beyond the recorded head')
            rc = self.L.mm_map_text(self.al, ctypes.c_void_p(self.text[0] + lo), hi - lo, self.lanes, col.cb, None)
        elif packed is not None:
            arr = (ctypes.c_void_p * len(packed))(*packed)
            rc = self.L.mm_map_packed(self.al, arr, len(packed), self.lanes, col.cb, None)
        else:
            rc = self.L.mm_map_reads(self.al, self.reads, first, n, self.lanes, col.cb, None)
        if rc != 0: raise RuntimeError('mapping failed (rc %d)' % rc)
        return col, self.L.mm_align_get_carry(self.al)

    def map(self, carry_in=None):
        self.carry_in = self.guess if carry_in is None else carry_in
        self.col, self.carry_out = self._map(self.first, self.n, self.carry_in, self.keep, self.packed, whole=True); self._stale = False
        if self.text is not None: self.n = self.L.mm_head_count(self.al)          # (the reads of a text are known once it has been scanned; only the head matters here)
        return self

    def _settle_local(self, truth):
        """this shard against the true value at its start; returns True if the value at its end changed"""
        if truth == self.carry_in: return False
        old_out = self.carry_out
        self.stats['checks'] += 1
        fa = ctypes.c_uint32(0)
        rc = 2 if self._stale else self.L.mm_carry_check(self.al, truth, ctypes.byref(fa))
        if rc == 0: self.carry_in = truth; return False
        if rc == 1 and self.n > 0:
            # read i0 is the first that decides differently: map a window [i0, i0 + m) again with the true value and splice its records in as soon as the
            # chain of values behind the window is what it was (then nothing behind it changes); m doubles while it is not.  Where the records of a read begin
            # in the text comes from the library (mm_head_offset, recorded by the writer of the stream), not from reading the text: formats other than SAM
            # and read names that repeat are cut just as well
            i0 = fa.value; limit = min(self.n, 4096)
            ends = []; m = 64
            while i0 + m <= limit: ends.append(m); m *= 2
            # read now: a window run replaces the stream these come from
            after = {m: self.L.mm_carry_after(self.al, i0 + m - 1) for m in ends}
            offs = {m: self.L.mm_head_offset(self.al, i0 + m) for m in ends}
            cut_lo = self.L.mm_head_offset(self.al, i0)
            if self.text is not None:          # the windows as slices of the text
                txt = {m: (self.L.mm_head_text_offset(self.al, i0), self.L.mm_head_text_offset(self.al, i0 + m)) for m in ends}
            for m in ends:
                cut_hi = offs[m]
                if cut_lo == NO_OFFSET or cut_hi == NO_OFFSET or cut_hi > self.col.kept or after[m] == NO_CARRY: break      # beyond what was recorded / kept
                if self.text is not None:
                    if NO_OFFSET in txt[m]: break
                    col = Collector(None); self.L.mm_align_set_carry(self.al, truth)
                    if self.L.mm_map_text(self.al, ctypes.c_void_p(self.text[0] + txt[m][0]), txt[m][1] - txt[m][0], self.lanes, col.cb, None) != 0: raise RuntimeError('mapping failed')
                    out = self.L.mm_align_get_carry(self.al)
                else:
                    col, out = self._map(self.first + i0, m, truth, None)
                self._stale = True
                self.stats['remapped_reads'] += m
                if out == after[m]:
                    self.col.splice(cut_lo, cut_hi, col.pieces)
                    self.carry_in = truth
                    return False
        # undecided inside the recorded head, or the window outgrew what can be spliced: the whole shard again with the true value
        self.stats['full_remaps'] += 1
        self.col, self.carry_out = self._map(self.first, self.n, truth, self.keep, self.packed, whole=True)
        self.carry_in = truth; self._stale = False
        return self.carry_out != old_out

    def settle(self, dist=None, rank=0, world=1, initial=0, device=None):
        """exchange end values until every shard has run with the value the shard in front of it ended with"""
        if world == 1:
            self._settle_local(initial); return self
        import torch
        for sweep in range(world):
            t = torch.tensor([self.carry_out, 1 if self.n > 0 else 0], dtype=torch.int64, device=device or 'cpu')
            outs = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            outs = [(int(o[0]), int(o[1])) for o in outs]
            # the value in front of rank r: the end value of the nearest non-empty shard before it (an empty shard passes its start value on)
            truth = initial
            for r in range(rank):
                if outs[r][1]: truth = outs[r][0]
            changed = self._settle_local(truth)
            self.stats['sweeps'] = sweep + 1
            flag = torch.tensor([1 if changed else 0], dtype=torch.int64, device=device or 'cpu')
            dist.all_reduce(flag)
            if int(flag[0]) == 0: break
        return self


def text_part(fn, rank, world):
    """this rank's stretch of a plain FASTA file as (keepalive, address, length): the file is mapped and cut by bytes where a '>' starts a line (such a '>' always
    starts a record for the reader, minialign.c:1996-2090), so the stretches in rank order are the file and nothing is read twice; None for anything else (gzip,
    FASTQ, stdin: the caller falls back to mm_reads_load_part)"""
    import mmap, numpy as np
    if fn == '-' or not os.path.isfile(fn): return None
    with open(fn, 'rb') as f:
        head = f.read(4); size = os.fstat(f.fileno()).st_size
        first = next((c for c in head if c in b'>@'), None)
        if size == 0 or first != ord('>'): return None
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    def cut(b):
        if b <= 0: return 0
        if b >= size: return size
        q = mm.find(b'\n>', b - 1)
        return size if q < 0 else q + 1
    lo, hi = cut(size * rank // world), cut(size * (rank + 1) // world)
    arr = np.frombuffer(mm, dtype=np.uint8)
    return (mm, arr), arr.ctypes.data + lo, hi - lo


def write_in_rank_order(dist, rank, world, fd, chunks):
    """rank r writes `chunks` to fd once rank r - 1 has finished writing (a one-integer token from rank to rank): the ordered drain of minialign.c:4633-4645
    across processes, every rank writing its own records"""
    import torch
    tok = torch.zeros(1, dtype=torch.int64)
    if world > 1 and rank > 0: dist.recv(tok, src=rank - 1)
    for c in chunks:
        v = memoryview(c)
        while len(v): v = v[os.write(fd, v):]
    if world > 1 and rank + 1 < world: dist.send(tok, dst=rank + 1)


def main(argv=None):
    """torchrun entry: every rank maps its shard and writes its own records, in rank order (rank 0 writes the header first)"""
    import torch.distributed as dist
    argv = list(sys.argv[1:] if argv is None else argv)
    # the records go to the process' real standard output; whatever libraries print there meanwhile (gloo announces its connections on stdout) goes to stderr
    sys.stdout.flush(); out_fd = os.dup(1); os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0')); local = int(os.environ.get('LOCAL_RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1'))
    dev = 0 if os.environ.get('MM_MULTI_SAME_DEVICE') is not None else local          # test hook: every rank on device 0 (one-GPU boxes)
    if world > 1:
        dist.init_process_group('gloo')          # three integers per rank and a token: no RCCL on this path
        os.environ['MM_DEVICES'] = '1'           # one process per GPU here: the context of a rank stays on its own device (a lone process spans all it sees)
    L = load_library()
    if L.mm_set_device(dev) != 0: raise RuntimeError('no HIP device %d' % dev)
    o = ctypes.c_void_p(L.mm_opt_init())
    args = [b'minialign'] + [a.encode() for a in argv]
    av = (ctypes.c_char_p * len(args))(*args); files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
    if L.mm_opt_parse(o, len(args), av, files, 8, ctypes.byref(nf)) != 0 or nf.value != 2: raise SystemExit('usage: minialign_amd.multi [options] ref.fa reads.fa')
    mi = ctypes.c_void_p(L.mm_idx_gen(o, files[0])); al = ctypes.c_void_p(L.mm_align_init(o, mi)) if mi else None
    if not al: raise RuntimeError('index / device context failed')
    # this rank's part of the read file only: a plain FASTA file is mapped and cut by bytes at record starts, and the stretch goes to the device as text (records
    # found and packed there); anything else through the host's parser, the part keeping its share of the records.  The parts in rank order are the file.
    tp = text_part(files[1].decode(), rank, world) if not os.environ.get('MM_HOST_READER') else None
    if tp is not None:
        sm = ShardMapper(L, al, None, 0, 0, guess=L.mm_idx_max_len(mi), text=(tp[1], tp[2])).map()
    else:
        reads = ctypes.c_void_p(L.mm_reads_load_part_opt(o, files[1], rank, world))          # (with -L / -Q / -T CO of the command line, as the text path applies them)
        if not reads: raise RuntimeError('cannot read %r' % files[1])
        sm = ShardMapper(L, al, reads, 0, L.mm_reads_count(reads), guess=L.mm_idx_max_len(mi)).map()
    sm.settle(dist if world > 1 else None, rank, world, 0, None)
    head = []
    if rank == 0:
        r, w = os.pipe()          # the header through the library's own printer
        libc = ctypes.CDLL(None); libc.fdopen.restype = ctypes.c_void_p; libc.fdopen.argtypes = [ctypes.c_int, ctypes.c_char_p]; libc.fclose.argtypes = [ctypes.c_void_p]
        import threading
        got = []; t = threading.Thread(target=lambda: got.append(os.fdopen(r, 'rb').read())); t.start()
        fp = ctypes.c_void_p(libc.fdopen(w, b'w'))
        L.mm_print_sam_header.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
        L.mm_print_sam_header(al, fp, b' '.join(args)); libc.fclose(fp); t.join()
        head = got
    prefix = os.environ.get('MM_MULTI_PARTS')
    if prefix:
        with open('%s.%04d' % (prefix, rank), 'wb') as f:
            for c in head + sm.col.pieces: f.write(c)
    else:
        write_in_rank_order(dist if world > 1 else None, rank, world, out_fd, head + sm.col.pieces)
    os.close(out_fd)
    if rank == 0: sys.stderr.write('[minialign_amd.multi] %d rank(s); carried value (rank 0): %r\n' % (world, sm.stats))
    if world > 1: dist.barrier(); dist.destroy_process_group()


if __name__ == '__main__':
    main()
