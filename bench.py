#!/usr/bin/env python3
"""bench.py -- aligned Gbases/s of the MI355X hot path (sketch + index lookup -> seed sort + chaining -> banded extension)
on BASELINE.json's configs[1]: an E.coli-MG1655-sized reference x PBSIM-CLR-like reads at x100 (about 460 Mb), -xpacbio.

A "step" is one pass of the hot path over the whole read set (one batch), with the 2-bit packed reads, the reference
and the index already resident in HBM.  Synthetic data comes from the repo's own seeded generator (tools/gensim.c).
For N > 1 GPUs (torch.distributed.run, one process per GPU) every rank maps its own read set against its own replica
of the index: weak scaling, no collective on the data path (the only collectives are the barrier / max of the timing).

Prints ONE JSON line (see the contract in the task description) with `roofline` (dominant kernel mm_extend_kernel:
algorithmic bytes = DP vectors x 40.5 B + traceback steps x 32 B per launch, SURVEY.md 8d, over the kernel's average
launch time from HIP events) and `cpu_baseline` (the compiled reference when oracle/_ref travelled with the snapshot,
else the repo's plain-C oracle, on a bounded sample of the same reads)."""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')      # before the HIP runtime starts: the lanes' streams should not share hardware queues
import argparse, ctypes, json, os, re, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.abspath(__file__))
GENOME_LEN = 4641652          # E.coli K-12 MG1655
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s

class Stats(ctypes.Structure):
    _fields_ = [('k1_ms', ctypes.c_double), ('k2_ms', ctypes.c_double), ('k3_ms', ctypes.c_double),
                ('k1_launches', ctypes.c_uint64), ('k2_launches', ctypes.c_uint64), ('k3_launches', ctypes.c_uint64),
                ('reads', ctypes.c_uint64), ('bases', ctypes.c_uint64), ('minimizers', ctypes.c_uint64), ('seeds', ctypes.c_uint64),
                ('fills', ctypes.c_uint64), ('vectors', ctypes.c_uint64), ('blocks', ctypes.c_uint64), ('traces', ctypes.c_uint64),
                ('trace_steps', ctypes.c_uint64), ('reruns', ctypes.c_uint64),
                ('host_post_ms', ctypes.c_double), ('host_sam_ms', ctypes.c_double), ('wall_ms', ctypes.c_double),
                ('k3_cycles_fill', ctypes.c_uint64), ('k3_cycles_leaf', ctypes.c_uint64), ('k3_cycles_trace', ctypes.c_uint64),
                ('k3_cycles_total', ctypes.c_uint64), ('k3_cycles_next', ctypes.c_uint64), ('k3_cycles_max', ctypes.c_uint64), ('k3_waves', ctypes.c_uint64), ('k2_cycles_sort', ctypes.c_uint64), ('k2_cycles_chain', ctypes.c_uint64),
                ('k2_cycles_total', ctypes.c_uint64), ('k2_reads_hbm', ctypes.c_uint64)]

def gensim(*args, out):
    exe = os.path.join(ROOT, 'tools', 'gensim')
    if not os.path.exists(exe):
        subprocess.check_call(['gcc', '-O2', '-o', exe, os.path.join(ROOT, 'tools', 'gensim.c'), '-lm'])
    with open(out, 'wb') as f:
        subprocess.check_call([exe] + [str(a) for a in args], stdout=f)

def cpu_baseline(ref_fa, reads_fa, workdir, budget_reads):
    """time the CPU path on a bounded sample of the same reads (rank 0, N = 1 only)"""
    sample = os.path.join(workdir, 'sample.fa')
    n = 0; bases = 0
    with open(reads_fa, 'rb') as f, open(sample, 'wb') as g:
        for line in f:
            if line.startswith(b'>'):
                n += 1
                if n > budget_reads: break
            else:
                bases += len(line) - 1
            g.write(line)
    refbin = os.path.join(ROOT, 'oracle', '_ref', 'minialign')
    cores = os.cpu_count() or 1
    if os.path.exists(refbin):
        nth = min(cores, 16)
        r = subprocess.run([refbin, '-xpacbio', '-t%d' % nth, ref_fa, sample], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        # map phase = "finished mapping" - "loaded/built index" stamps (minialign.c:6417,6431), as the README measures it
        ts = [float(m.group(1)) for m in re.finditer(rb'\[M::main_align::([0-9.]+)\*', r.stderr)]
        sec = (ts[-1] - ts[0]) if len(ts) >= 2 else None
        if r.returncode == 0 and sec and sec > 0:
            return {'value': bases / sec * 1e-9, 'unit': 'Gbases/s', 'cores': nth, 'kind': 'reference',
                    'sample': 'first %d reads (%.1f Mb) of the same set, oracle/_ref/minialign -xpacbio -t%d, map phase only' % (min(n, budget_reads), bases / 1e6, nth)}
    ora = os.path.join(ROOT, 'oracle', 'ora_minialign')
    small = os.path.join(workdir, 'sample_small.fa'); k = 0; b2 = 0
    with open(sample, 'rb') as f, open(small, 'wb') as g:
        for line in f:
            if line.startswith(b'>'):
                k += 1
                if k > 300: break
            else: b2 += len(line) - 1
            g.write(line)
    r = subprocess.run([ora, '-xpacbio', ref_fa, small], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    m = re.search(rb'in ([0-9.]+) s', r.stderr)
    sec = float(m.group(1)) if m else None
    return {'value': (b2 / sec * 1e-9) if sec else None, 'unit': 'Gbases/s', 'cores': 1, 'kind': 'port',
            'sample': 'first %d reads (%.1f Mb), oracle/ora_minialign (plain-C restatement, single thread), mm_align_seq time only' % (min(k, 300), b2 / 1e6)}

def pmc_traffic(args, world):
    """HBM bytes per mm_extend_kernel launch from the committed rocprofv3 PMC passes (same workload only), else None"""
    fn = os.path.join(ROOT, 'profiles', 'round1_j_pmc.json')          # tools/pmc_traffic.sh on the code as it stands
    if world != 1 or args.depth != 100.0 or args.repeat_frac != 0.05 or args.genome_len != GENOME_LEN or args.contigs != 1 or not os.path.exists(fn): return None
    try:
        with open(fn) as f: return json.load(f)['mm_extend_kernel_per_launch']['hbm_bytes']
    except Exception:
        return None

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1); ap.add_argument('--steps', type=int, default=6); ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--depth', type=float, default=100.0, help='read depth over the 4.64 Mb reference (x100 = BASELINE configs[1])')
    ap.add_argument('--genome-len', type=int, default=GENOME_LEN, help='length of the synthetic reference (default workload: E.coli MG1655, 4 641 652)')
    ap.add_argument('--contigs', type=int, default=1, help='number of contigs of the synthetic reference (default workload: 1)')
    ap.add_argument('--repeat-frac', type=float, default=0.05, help='fraction of the synthetic reference made of planted repeats (default workload: 0.05)')
    ap.add_argument('--check', action='store_true', help='also verify the SAM of a sample against the CPU oracle')
    ap.add_argument('--stagger-ms', type=float, default=0.0, help='delay of the second lane at the start of the timed region')
    ap.add_argument('--inflight', type=int, default=3, choices=(1, 2, 3, 4), help='batches in flight per GPU: consecutive steps go to alternating lanes of the device context and overlap, as the batches of a read stream do (1: strictly one after the other)')
    args = ap.parse_args()
    rank = int(os.environ.get('RANK', '0')); local = int(os.environ.get('LOCAL_RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group('nccl')
    lib = os.environ.get('MM_LIB_OVERRIDE') or os.path.join(ROOT, 'minialign_amd', 'libminialign_amd.so')     # override: kernel experiments only
    if not os.path.exists(lib):
        sys.path.insert(0, ROOT); import __graft_entry__; __graft_entry__.build()
    L = ctypes.CDLL(lib)
    for f in ('mm_opt_init', 'mm_idx_gen', 'mm_align_init', 'mm_reads_load', 'mm_batch_upload', 'mm_batch_upload_lane'): getattr(L, f).restype = ctypes.c_void_p
    L.mm_reads_bases.restype = ctypes.c_uint64
    assert L.mm_set_device(local) == 0, 'no HIP device %d' % local

    work = tempfile.mkdtemp(prefix='mmbench_')
    ref_fa = os.path.join(work, 'ref.fa'); reads_fa = os.path.join(work, 'reads_%d.fa' % rank)
    gensim('genome', 0x5eed0001, args.genome_len, args.contigs, args.repeat_frac, out=ref_fa)
    gensim('reads', 0x5eed0002 + rank, ref_fa, args.depth, 'pacbio', 'fa', 20000, 2000, out=reads_fa)

    o = ctypes.c_void_p(L.mm_opt_init())
    argv = (ctypes.c_char_p * 4)(b'minialign', b'-xpacbio', ref_fa.encode(), reads_fa.encode())
    files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
    assert L.mm_opt_parse(o, 4, argv, files, 8, ctypes.byref(nf)) == 0
    t0 = time.time()
    mi = ctypes.c_void_p(L.mm_idx_gen(o, ref_fa.encode())); assert mi
    al = ctypes.c_void_p(L.mm_align_init(o, mi)); assert al, 'mm_align_init failed (no GPU?)'
    t_index = time.time() - t0
    reads = ctypes.c_void_p(L.mm_reads_load(reads_fa.encode())); assert reads
    n_reads = L.mm_reads_count(reads); bases = L.mm_reads_bases(reads, 0, n_reads)
    # one copy of the batch per lane; a step = one pass of the hot path over the batch.  With --inflight 2 consecutive steps go to
    # alternating lanes and overlap (the way consecutive batches of a read stream do); every step still runs K1..K3 completely.
    batches = [ctypes.c_void_p(L.mm_batch_upload_lane(al, reads, 0, n_reads, ln)) for ln in range(args.inflight)]
    assert all(batches), 'upload failed'
    batch = batches[0]

    def sync():
        torch.cuda.synchronize()
        if dist: dist.barrier()
    def run_steps(k):
        if args.inflight == 1:
            for _ in range(k):
                assert L.mm_batch_run(al, batch) == 0          # blocks until the last kernel of the step has finished
            return
        pending = [False] * args.inflight
        for i in range(k):
            ln = i % args.inflight
            if pending[ln]: assert L.mm_batch_wait(al, batches[ln]) == 0
            if i == 1 and args.stagger_ms > 0: time.sleep(args.stagger_ms * 1e-3)
            assert L.mm_batch_run_async(al, batches[ln]) == 0; pending[ln] = True
        for ln in range(args.inflight):
            if pending[ln]: assert L.mm_batch_wait(al, batches[ln]) == 0
    run_steps(args.warmup)
    L.mm_stats(al, None, 1)
    sync(); t0 = time.perf_counter()
    run_steps(args.steps)
    sync(); dt = time.perf_counter() - t0
    st = Stats(); L.mm_stats(al, ctypes.byref(st), 0)
    if dist:
        t = torch.tensor([dt], device='cuda'); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
        tb = torch.tensor([float(bases)], device='cuda', dtype=torch.float64); dist.all_reduce(tb); total_bases = float(tb.item())
    else:
        total_bases = float(bases)
    # one untimed finish (D2H + post-map + SAM) to report the end-to-end rate and, optionally, check a sample
    sam = ctypes.c_char_p(); slen = ctypes.c_uint64(0)
    t1 = time.perf_counter(); assert L.mm_batch_finish(al, batch, ctypes.byref(sam), ctypes.byref(slen)) == 0; t_finish = time.perf_counter() - t1
    st2 = Stats(); L.mm_stats(al, ctypes.byref(st2), 0)

    if rank == 0:
        k3_launch_ms = st.k3_ms / max(1, st.k3_launches)
        per_step = lambda x: x / max(1, args.steps)
        # work counters are read at finish time and cover the last pass over the batch (each pass re-initialises the device state)
        vec = float(st2.vectors); trs = float(st2.trace_steps)
        # SURVEY.md 8d per-unit figures for the extension kernel; the two lanes of a step launch it once each (half of the batch)
        k3_per_step = max(1.0, st.k3_launches / max(1, args.steps))
        alg_bytes = (vec * 40.5 + trs * 32.0) / k3_per_step
        achieved = alg_bytes / (k3_launch_ms * 1e-3) / 1e9 if k3_launch_ms > 0 else None
        out = {
            'metric': 'aligned Gbases/sec (hot path: sketch+lookup, sort+chain, banded extension; SAM bit-exact vs CPU ref)',
            'value': total_bases * args.steps / dt * 1e-9, 'unit': 'Gbases/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'i8',
            'data': 'synthetic (tools/gensim.c: %.2f Mb reference with %g %% planted repeats, PBSIM-CLR-like reads 20k+-2k, acc 0.88+-0.07)' % (args.genome_len / 1e6, args.repeat_frac * 100),
            'config': {'workload': ('E.coli MG1655-size ref' if (args.genome_len == GENOME_LEN and args.contigs == 1 and args.repeat_frac == 0.05) else 'synthetic %.1f Mb / %d contig(s) / %g repeats ref' % (args.genome_len / 1e6, args.contigs, args.repeat_frac))
                                   + ' x PBSIM-like x%g (%.0f Mb, %d reads) -xpacbio on 1 MI355X per rank' % (args.depth, bases / 1e6, n_reads),
                       'reads_per_rank': n_reads, 'bases_per_rank': bases, 'batches_in_flight': args.inflight, 'parallelism': 'reads sharded, index replicated (no collective)',
                       'kernel_ms_per_step': {'sketch_seed': per_step(st.k1_ms), 'sort_chain': per_step(st.k2_ms), 'extend': per_step(st.k3_ms)},
                       'extend_wave_time_split': {k: getattr(st2, 'k3_cycles_' + k) / max(1, st2.k3_cycles_total) for k in ('fill', 'leaf', 'trace', 'next')},
                       'extend_wave_balance (mean / max lifetime)': st2.k3_cycles_total / max(1, st2.k3_cycles_max * st2.k3_waves),
                       'sort_chain_wave_time_split': {'sort': st2.k2_cycles_sort / max(1, st2.k2_cycles_total), 'chain': st2.k2_cycles_chain / max(1, st2.k2_cycles_total),
                                                      'reads_not_in_lds': st2.k2_reads_hbm, 'sort_cycles_per_seed': st2.k2_cycles_sort / max(1, st2.seeds), 'chain_cycles_per_seed': st2.k2_cycles_chain / max(1, st2.seeds), 'seeds_per_read': st2.seeds / max(1, st2.reads)},
                       'dp_vectors_per_base': vec / bases, 'trace_steps_per_base': trs / bases, 'reruns_per_step': per_step(st.reruns), 'index_build_s': t_index,
                       'finish_s (D2H + post-map + SAM, untimed)': t_finish, 'sam_bytes': slen.value},
            'roofline': {'bound': 'hbm', 'kernel': 'mm_extend_kernel', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': (achieved / HBM_PEAK_GBS) if achieved else None, 'traffic': pmc_traffic(args, world),
                         'alg_bytes_per_launch': alg_bytes, 'avg_launch_ms': k3_launch_ms,
                         # with several batches in flight the launches of the lanes share the chip: one launch lasts longer than it would alone, so besides the
                         # per-launch figure above, the same bytes over the wall time of the timed region (all lanes of this rank together)
                         'achieved_all_lanes': (alg_bytes / (dt / args.steps) * 1e-9) if alg_bytes else None,
                         'note': 'the kernel is integer-VALU-issue bound, not HBM bound (DESIGN.md 4): traffic = PMC bytes per launch from profiles/round1_j_pmc.json'},
        }
        if world == 1:
            out['cpu_baseline'] = cpu_baseline(ref_fa, reads_fa, work, 20000)      # about 15 s of CPU work at 16 threads
        print(json.dumps(out), flush=True)
    if dist: dist.destroy_process_group()

if __name__ == '__main__':
    main()
