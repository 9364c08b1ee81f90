#!/usr/bin/env python3
"""bench.py -- aligned Gbases/s of the MI355X mapper on BASELINE.json's headline shape, timed end to end over the map phase.

Default workload (`--workload hg38`): a human-genome-size synthetic reference (3.1 Gb in 25 contigs, 5 % planted repeats, N runs) x PBSIM-CLR-like reads at
x3 (9.2 Gb, about 447 000 reads of 20 kb +- 2 kb, accuracy 0.88 +- 0.07, sub:ins:del 10:60:30), `-xpacbio` -- the configuration BASELINE.json's metric is quoted
on.  `--workload dm6 | ecoli | ont` give the other BASELINE shapes (D.melanogaster size x20; E.coli MG1655 size x100; human size x ONT-like reads with
`-xont.1dsq`), `--genome-len / --contigs / --depth / --repeat-frac` any other.  Data comes from the repo's own seeded generator (tools/gensim.c, 16 parts
generated side by side; the set is their concatenation).

A step = the whole read set once through the map phase of the reference (minialign.c:6417-6431; its reader bseq_read_fasta :1996 included): the timed region
starts from the FASTA TEXT of the read set in host memory and ends with the SAM text of every read in host memory -- text to HBM, K0r record scanning, K0 base
conversion + 2-bit packing, K1 sketch + lookup, K2 sort + chain, K3 banded extension (in rounds, several batches in flight on the lanes of the device context),
D2H, post-map, SAM formatting.  Only index construction is outside (as in the README's figure).  `config.value_from_packed` keeps the earlier rounds' figure (the
same steps from reads parsed and 2-bit packed ahead of time), `config.cli_map_phase_s` is the map phase of the command-line program itself over the same set
(`minialign_amd/minialign ref.fa reads.fa > /dev/null`, between its "loaded/built index" and "finished mapping" stamps as minialign.c:6417,6431 puts them).
With N > 1 GPUs the SAME read set goes over N devices (strong scaling), either way with a replica of the index in every GPU's HBM and no collective on the data path:
  * `python bench.py --gpus N` launched plainly: ONE process, the drop-in's own way -- the library's context spans N devices (mm_align_init, MM_DEVICES=N), its
    streaming engine deals the batches of the one text to device x lane, verifies the carried value in batch order and writes in input order (what the command-line
    program does with N visible GPUs);
  * under torch.distributed.run (WORLD_SIZE = N, one process per GPU, as the driver launches N > 1): rank r maps parts r*16/N .. of the set on its own device, the
    ranks settle the one value reads share (the carried reference length, minialign_amd/multi.py) with one tiny all_gather over gloo, value = total bases / max
    over ranks of the time.

Prints ONE JSON line: metric / value / ... as the contract asks, `roofline` for the dominant kernel (mm_extend_kernel: DP vectors x 40.5 B + traceback
steps x 32 B, SURVEY.md 8d, counted by the kernel, over the summed launch time from HIP events on the launch streams), `cpu_baseline` (the compiled
reference oracle/_ref/minialign when it travelled with the snapshot, else the plain-C oracle, on a bounded sample of the same reads; N = 1 only) and
`sam_identical` (the records of the first reads against the compiled reference at -t1 -- the reference's output depends on its thread count through the
carried value, DESIGN.md 5 -- or against the oracle; N = 1 or --check)."""
import os as _os
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')      # before the HIP runtime starts: the lanes' streams should not share hardware queues
import argparse, ctypes, json, os, re, shutil, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
PARTS = 16
WORKLOADS = {
    'hg38':  dict(genome_len=3100000000, contigs=25,   repeat_frac=0.05, depth=3.0,   kind='pacbio', preset='pacbio',   name='human hg38-size ref (3.1 Gb, 25 contigs)'),
    'dm6':   dict(genome_len=143700000,  contigs=1870, repeat_frac=0.05, depth=20.0,  kind='pacbio', preset='pacbio',   name='D.melanogaster dm6-size ref (143.7 Mb, 1870 contigs)'),
    'ecoli': dict(genome_len=4641652,    contigs=1,    repeat_frac=0.05, depth=100.0, kind='pacbio', preset='pacbio',   name='E.coli MG1655-size ref (4.64 Mb)'),
    'hg38hard': dict(genome_len=3100000000, contigs=25, repeat_frac=0.45, depth=1.0,  kind='pacbio', preset='pacbio',  hard=True, baseline_reads=20000, check_reads=2000, name='human-size ref with mammalian repeat structure (3.1 Gb, 25 contigs: tools/gensim.c genomehard -- SINE / LINE-like families of 10^5 .. 10^6 copies, 200 mid-size families, segmental duplications, satellite arrays, N gaps)'),
    'ont':   dict(genome_len=3100000000, contigs=25,   repeat_frac=0.05, depth=1.0,   kind='ont',    preset='ont.1dsq', baseline_reads=8000, check_reads=1000, name='human hg38-size ref (3.1 Gb, 25 contigs), ONT-like reads'),
}

class Stats(ctypes.Structure):
    _fields_ = [('k1_ms', ctypes.c_double), ('k2_ms', ctypes.c_double), ('k3_ms', ctypes.c_double),
                ('k1_launches', ctypes.c_uint64), ('k2_launches', ctypes.c_uint64), ('k3_launches', ctypes.c_uint64),
                ('reads', ctypes.c_uint64), ('bases', ctypes.c_uint64), ('minimizers', ctypes.c_uint64), ('seeds', ctypes.c_uint64),
                ('fills', ctypes.c_uint64), ('vectors', ctypes.c_uint64), ('blocks', ctypes.c_uint64), ('traces', ctypes.c_uint64),
                ('trace_steps', ctypes.c_uint64), ('reruns', ctypes.c_uint64),
                ('host_post_ms', ctypes.c_double), ('host_sam_ms', ctypes.c_double), ('wall_ms', ctypes.c_double),
                ('k3_cycles_fill', ctypes.c_uint64), ('k3_cycles_leaf', ctypes.c_uint64), ('k3_cycles_trace', ctypes.c_uint64),
                ('k3_cycles_total', ctypes.c_uint64), ('k3_cycles_next', ctypes.c_uint64), ('k3_cycles_max', ctypes.c_uint64), ('k3_waves', ctypes.c_uint64), ('k2_cycles_sort', ctypes.c_uint64), ('k2_cycles_chain', ctypes.c_uint64),
                ('k2_cycles_total', ctypes.c_uint64), ('k2_reads_hbm', ctypes.c_uint64), ('pool_grows', ctypes.c_uint64), ('batch_splits', ctypes.c_uint64), ('pool_regrows', ctypes.c_uint64),
                ('text_bytes', ctypes.c_uint64), ('reader_ms', ctypes.c_double), ('d2h_bytes', ctypes.c_uint64), ('cigar_bytes_device', ctypes.c_uint64), ('k3_aborts', ctypes.c_uint64)]

def gensim_exe():
    exe = os.path.join(ROOT, 'tools', 'gensim')
    if not os.path.exists(exe):
        subprocess.check_call(['gcc', '-O2', '-o', exe, os.path.join(ROOT, 'tools', 'gensim.c'), '-lm'])
    return exe

def generate(work, w, seed=0x5eed0001):
    """reference + the PARTS read files of the set (generated side by side)"""
    exe = gensim_exe(); ref_fa = os.path.join(work, 'ref.fa')
    with open(ref_fa, 'wb') as f: subprocess.check_call([exe, 'genomehard' if w.get('hard') else 'genome', str(seed), str(w['genome_len']), str(w['contigs']), str(w['repeat_frac'])], stdout=f)
    parts = [os.path.join(work, 'reads_%02d.fa' % p) for p in range(PARTS)]; procs = []
    for p, fn in enumerate(parts):
        f = open(fn, 'wb')
        procs.append((subprocess.Popen([exe, 'reads', str(seed + 1), ref_fa, str(w['depth']), w['kind'], 'fa', '20000', '2000', str(p), str(PARTS)], stdout=f), f))
    for pr, f in procs:
        if pr.wait() != 0: raise RuntimeError('gensim failed')
        f.close()
    return ref_fa, parts

def head_reads(parts, n, out):
    """the first n reads of the set as one FASTA file; returns (reads written, bases)"""
    k = 0; bases = 0
    with open(out, 'wb') as g:
        for fn in parts:
            with open(fn, 'rb') as f:
                for line in f:
                    if line.startswith(b'>'):
                        k += 1
                        if k > n: return n, bases
                    else: bases += len(line) - 1
                    g.write(line)
    return k, bases

def strip_header(sam): return b''.join(l for l in sam.splitlines(True) if not l.startswith(b'@'))

def reference_runs(w, ref_fa, parts, work, n_check, n_time, want_check):
    """the CPU side, rank 0 only: the compiled reference when it travelled with the snapshot (oracle/_ref; index file built once, then the first n_check reads
    at -t1 for the identity check and the first n_time reads on many threads for the baseline), else the repo's plain-C oracle on a small sample"""
    refbin = os.path.join(ROOT, 'oracle', '_ref', 'minialign'); cores = os.cpu_count() or 1
    out = {'cpu_baseline': None, 'check_sam': None, 'check_reads': 0, 'check_kind': None}
    if os.path.exists(refbin):
        nth = max(1, min(cores, 64))          # the reference refuses 128 threads and more (minialign.c:5973)
        mai = os.path.join(work, 'ref.mai')
        r = subprocess.run([refbin, '-x' + w['preset'], '-t%d' % nth, '-d', mai, ref_fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        if r.returncode == 0:
            tf = os.path.join(work, 'time.fa'); nt, bt = head_reads(parts, n_time, tf)
            # its pipeline does not scale with the thread count on every host (one source / drain thread): the best of a few counts is the baseline
            for t in sorted(set(max(1, min(cores, x)) for x in (16, 32, 64))):
                r = subprocess.run([refbin, '-x' + w['preset'], '-t%d' % t, mai, tf], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                # map phase = "finished mapping" - "loaded/built index" stamps (minialign.c:6417,6431), as the README measures it
                ts = [float(m.group(1)) for m in re.finditer(rb'\[M::main_align::([0-9.]+)\*', r.stderr)]
                sec = (ts[-1] - ts[0]) if len(ts) >= 2 else None
                if r.returncode == 0 and sec and sec > 0 and (out['cpu_baseline'] is None or bt / sec * 1e-9 > out['cpu_baseline']['value']):
                    out['cpu_baseline'] = {'value': bt / sec * 1e-9, 'unit': 'Gbases/s', 'cores': t, 'kind': 'reference',
                                           'sample': 'first %d reads (%.1f Mb) of the same set, oracle/_ref/minialign -x%s -t%d (best of -t16/32/64) from its own index file, map phase only (%.2f s)' % (nt, bt / 1e6, w['preset'], t, sec)}
            if want_check:
                cf = os.path.join(work, 'check.fa'); nc, _ = head_reads(parts, n_check, cf)
                r = subprocess.run([refbin, '-x' + w['preset'], '-t1', mai, cf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                if r.returncode == 0: out.update(check_sam=strip_header(r.stdout), check_reads=nc, check_kind='oracle/_ref/minialign -t1 (the compiled reference)')
            return out
    ora = os.path.join(ROOT, 'oracle', 'ora_minialign')
    sf = os.path.join(work, 'small.fa'); ns, bs = head_reads(parts, min(n_check, 300), sf)
    r = subprocess.run([ora, '-x' + w['preset'], ref_fa, sf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    m = re.search(rb'in ([0-9.]+) s', r.stderr); sec = float(m.group(1)) if m else None
    out['cpu_baseline'] = {'value': (bs / sec * 1e-9) if sec else None, 'unit': 'Gbases/s', 'cores': 1, 'kind': 'port',
                           'sample': 'first %d reads (%.1f Mb), oracle/ora_minialign (plain-C restatement, single thread), mm_align_seq time only' % (ns, bs / 1e6)}
    if want_check and r.returncode == 0: out.update(check_sam=strip_header(r.stdout), check_reads=ns, check_kind='oracle/ora_minialign (plain-C restatement)')
    return out

def cli_map_phase(w, ref_fa, parts, work):
    """the drop-in itself: `minialign -x<preset> ref.fa reads.fa > /dev/null` over the same set (the parts as one file); map phase = "finished mapping" - "loaded/built index" stamps"""
    cli = os.path.join(ROOT, 'minialign_amd', 'minialign'); rd = os.path.join(work, 'reads_all.fa')
    try:
        with open(rd, 'wb') as g:
            for p in parts:
                with open(p, 'rb') as f: shutil.copyfileobj(f, g, 64 << 20)
        t0 = time.time()
        with open(os.devnull, 'wb') as dn: r = subprocess.run([cli, '-x' + w['preset'], ref_fa, rd], stdout=dn, stderr=subprocess.PIPE)
        wall = time.time() - t0
        ts = [float(m.group(1)) for m in re.finditer(rb'\[M::main_align::([0-9.]+)\]', r.stderr)]
        return {'cli_map_phase_s': (ts[-1] - ts[0]) if (r.returncode == 0 and len(ts) >= 2) else None, 'cli_wall_s (index build included)': wall, 'cli_index_s': ts[0] if ts else None}
    except Exception as e:
        return {'cli_map_phase_s': None, 'cli_error': repr(e)}
    finally:
        try: os.unlink(rd)
        except OSError: pass

def pmc_traffic(wname, world, alg_bytes_per_launch):
    """HBM bytes per mm_extend_kernel launch from the committed rocprofv3 PMC passes of the same workload (tools/pmc_traffic.sh; this round's, else the last round's), else None.  The PMC run maps
    a tenth of the set on one lane; its traffic per launch is scaled by the algorithmic bytes per launch of this run over those of that run (same kernel, same reads:
    traffic per vector is what the counters measured)."""
    fn = next((f for f in (os.path.join(ROOT, 'profiles', t + '_pmc.json') for t in ('round6', 'round5', 'round4', 'round3', 'round2')) if os.path.exists(f)), None)
    if world != 1 or fn is None: return None
    try:
        with open(fn) as f: d = json.load(f)
        per = d['mm_extend_kernel_per_launch']
        if d.get('workload') != wname: return None
        return per['hbm_bytes'] * alg_bytes_per_launch / per['alg_bytes_per_launch'] if per.get('alg_bytes_per_launch') else per['hbm_bytes']
    except Exception:
        return None

def valu_position(vectors, wall_s, world):
    """Where the run sits against the integer-VALU issue limit of the chip (the real bound of mm_extend_kernel): the VALU cycles per DP vector that the SQ
    counters measured (tools/pmc_sq.sh -> profiles/round2_pmc_sq.json, SQ_ACTIVE_INST_VALU) x the vectors of this run, over SIMDs x clock x wall time."""
    fn = next((f for f in (os.path.join(ROOT, 'profiles', t + '_pmc_sq.json') for t in ('round6', 'round5', 'round4', 'round3', 'round2')) if os.path.exists(f)), '')
    try:
        with open(fn) as f: per = json.load(f)['mm_extend_kernel_per_dp_vector']
        simds, clock = 256 * 4, 2.4e9          # MI355X: 256 CUs x 4 SIMDs, 2.4 GHz maximum engine clock (MI355X_MICROARCH.md)
        try:
            import torch
            pr = torch.cuda.get_device_properties(0); simds = int(pr.multi_processor_count) * 4
            if getattr(pr, 'clock_rate', 0): clock = float(pr.clock_rate) * 1e3
        except Exception:
            pass
        busy_s = per['valu_busy_cycles_per_dp_vector'] * vectors / world / simds / clock
        return {'valu_busy_cycles_per_dp_vector': per['valu_busy_cycles_per_dp_vector'], 'simds': simds, 'clock_hz (device maximum)': clock, 'valu_busy_s_per_step_per_gpu': busy_s, 'frac_of_wall': busy_s / wall_s,
                'note': 'fraction of the step during which the vector ALUs of the chip are issuing mm_extend_kernel instructions, at the maximum clock (the sustained clock is lower, so this is a lower bound)'}
    except Exception as e:
        return {'error': repr(e)}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1); ap.add_argument('--steps', type=int, default=2); ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='hg38', choices=sorted(WORKLOADS))
    ap.add_argument('--depth', type=float); ap.add_argument('--genome-len', type=int); ap.add_argument('--contigs', type=int); ap.add_argument('--repeat-frac', type=float)
    ap.add_argument('--lanes', type=int, default=4, help='batches in flight per GPU (lanes of the device context)')
    ap.add_argument('--check', action='store_true', help='verify the records of the first reads against the CPU reference also when N > 1')
    ap.add_argument('--check-reads', type=int); ap.add_argument('--baseline-reads', type=int)          # defaults: 4000 / 60000 reads; the ONT-like set has 385 kb reads, on which 64 threads of the reference need a lot of host memory: 1000 / 8000
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU legs (baseline and identity check)')
    ap.add_argument('--no-packed', action='store_true', help='skip the value_from_packed leg'); ap.add_argument('--no-cli', action='store_true', help='skip the command-line run')
    ap.add_argument('--keep', action='store_true', help='keep the generated data (prints the directory)')
    ap.add_argument('--early-line', action='store_true', help='print the JSON line once the timed steps are over, and again with the CPU legs in it (the hard-repeat record: the measurement is not lost when the CPU legs outlast the budget)')
    ap.add_argument('--no-hard', action='store_true', help='skip the hard-repeat record (config.hard_repeats: the hg38hard workload in a process of its own behind the timed steps; default workload on one GPU only)')
    args = ap.parse_args()
    os.environ['MM_LANES'] = str(args.lanes)          # the batch rule of the library (one batch per lane for a small set) sees the lanes this run uses
    rank = int(os.environ.get('RANK', '0')); local = int(os.environ.get('LOCAL_RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1'))
    # devices: under torch.distributed.run every rank drives ONE (its own); launched plainly, this one process drives --gpus N through the library's multi-device context
    n_inproc = max(1, args.gpus) if world == 1 else 1
    n_gpus = world if world > 1 else n_inproc
    w = dict(WORKLOADS[args.workload]); custom = False
    for k in ('depth', 'genome_len', 'contigs', 'repeat_frac'):
        if getattr(args, k) is not None: w[k] = getattr(args, k); custom = True
    if args.check_reads is None: args.check_reads = w.get('check_reads', 4000)
    if args.baseline_reads is None: args.baseline_reads = w.get('baseline_reads', 60000)
    import torch
    dist = None; device = None
    same_dev = os.environ.get('MM_BENCH_SAME_DEVICE') is not None      # test hook: every rank on device 0 (one-GPU boxes)
    if same_dev: local = 0
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group('gloo')                                 # a few integers per step (the carried value, the reductions of the report): no RCCL on this path (north_star)
    tdev = 'cpu'                                                        # where the few scalars of the reductions live
    from minialign_amd import multi
    lib = os.environ.get('MM_LIB_OVERRIDE') or os.path.join(ROOT, 'minialign_amd', 'libminialign_amd.so')     # override: kernel experiments only
    if not os.path.exists(lib):
        import __graft_entry__; __graft_entry__.build()
    L = multi.load_library(lib)
    assert L.mm_set_device(local) == 0, 'no HIP device %d' % local
    os.environ['MM_DEVICES'] = str(n_inproc)            # (read by mm_align_init: the devices the context spans, from the current one on)
    same_box = False
    if n_inproc > 1 and torch.cuda.device_count() < n_inproc:
        if not same_dev: raise SystemExit('bench.py --gpus %d: only %d HIP device(s) visible (MM_BENCH_SAME_DEVICE=1 puts the %d device contexts on one GPU: a test of the path, not a measurement)' % (n_inproc, torch.cuda.device_count(), n_inproc))
        os.environ['MM_DEVICES'] = str(max(1, torch.cuda.device_count())); os.environ['MM_DEVICE_CONTEXTS'] = str(n_inproc); same_box = True
        os.environ.setdefault('MM_SLAB_GB', str(max(4, 48 // n_inproc)))
    if world > 1 and not os.environ.get('MM_HOST_THREADS'):
        os.environ['MM_HOST_THREADS'] = str(max(8, (os.cpu_count() or 8) // world - 4))        # the ranks share the host cores

    # data: rank 0 generates into a directory every rank can name
    work = os.path.join(tempfile.gettempdir(), 'mmbench_%s_%s' % (os.environ.get('MASTER_PORT', 'solo'), os.environ.get('TORCHELASTIC_RUN_ID', str(os.getppid() if world > 1 else os.getpid()))))
    t_gen0 = time.time()
    if rank == 0:
        shutil.rmtree(work, ignore_errors=True); os.makedirs(work)
        ref_fa, parts = generate(work, w)
        open(os.path.join(work, 'ready'), 'w').close()
    if dist: dist.barrier()
    ref_fa = os.path.join(work, 'ref.fa'); parts = [os.path.join(work, 'reads_%02d.fa' % p) for p in range(PARTS)]
    t_gen = time.time() - t_gen0

    # the drop-in itself first, while this process holds nothing on the device (two processes with a dozen streams each would take turns on the hardware queues)
    cli_info = cli_map_phase(w, ref_fa, parts, work) if (rank == 0 and n_gpus == 1 and not args.no_cli) else {}
    hard_rec = None
    if rank == 0 and args.workload == 'hg38' and not custom and n_gpus == 1 and world == 1 and not args.no_hard and not args.no_cpu:
        t_h0 = time.time()
        if True:
            # the same path on a reference with mammalian repeat structure (45 % repeats: the headline set has 5 %), in a process of its own BEFORE this one has made its
            # lanes and streams: two processes with 16 hardware queues each oversubscribe the device's runlist and take turns on it (1.0 - 1.3 against 1.8 G bases/s
            # for this workload beside such a tenant) -- the record is what a user who runs the workload gets.  Value, DP vectors per base, records of the first reads against the
            # compiled reference, the reference's own speed beside it
            env = dict(os.environ); env.pop('MM_LIB_OVERRIDE', None)
            cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'hg38hard', '--steps', '2', '--warmup', '1', '--no-cli', '--no-packed', '--early-line', '--check-reads', '1000', '--baseline-reads', '8000', '--lanes', str(args.lanes)]
            try:
                # (the mapping itself is over within half a minute -- generation 10 s, index 3 s, three steps of 2 s; the line it prints then is kept whatever becomes of the CPU
                # legs behind it, which build the reference's own index of another 3.1 Gb genome on the host cores: 300 s for all of it.  An extension launch that does not
                # end is called off by the library's watchdog after 20 s and mapped again: the record then says so, extension_launches_called_off_by_the_watchdog)
                class R: pass
                r = R()
                try:
                    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300); r.stdout, r.stderr, r.note = p.stdout, p.stderr, None
                except subprocess.TimeoutExpired as e:
                    r.stdout, r.stderr, r.note = e.stdout or b'', e.stderr or b'', 'the CPU legs of the record did not finish within 300 s: the line printed behind the timed steps'
                    if not any(l.startswith('{') for l in r.stdout.decode(errors='replace').splitlines()): raise
                h = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith('{')][-1])
                hard_rec = {'value': h['value'], 'unit': h['unit'], 'ms_per_step': h['ms_per_step'], 'workload': h['config']['workload'], 'dp_vectors_per_base': h['config']['dp_vectors_per_base'],
                                                 'reruns_per_step': h['config']['reruns_per_step (rank 0)'], 'extend_wave_balance': h['config']['extend_wave_balance (mean / max lifetime)'],
                                                 'sam_identical': h.get('sam_identical'), 'sam_check': h.get('sam_check'), 'cpu_baseline': h.get('cpu_baseline'),
                                                 'extension_launches_called_off_by_the_watchdog': h['config'].get('extension_launches_called_off_by_the_watchdog (rank 0, timed steps)'),
                                                 'watchdog_log': [l for l in r.stderr.decode(errors='replace').splitlines() if 'watchdog' in l][:40] or None,
                                                 'note': 'python bench.py --workload hg38hard --steps 2 --warmup 1, run in a process of its own in front of the headline workload (this process holds one hardware queue then: DESIGN.md 4b)' + ('; ' + r.note if r.note else '')}
            except Exception as e:
                tail = getattr(e, 'stderr', None)
                hard_rec = {'error': repr(e)[:300], 'stderr_tail': (tail.decode(errors='replace')[-600:] if tail else None)}
            sys.stderr.write('[bench] hard-repeat record done (%.1f s)\n' % (time.time() - t_h0)); sys.stderr.flush()
    o = ctypes.c_void_p(L.mm_opt_init())
    argv = (ctypes.c_char_p * 4)(b'minialign', ('-x' + w['preset']).encode(), ref_fa.encode(), b'reads.fa')
    files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
    assert L.mm_opt_parse(o, 4, argv, files, 8, ctypes.byref(nf)) == 0
    t0 = time.time()
    mi = ctypes.c_void_p(L.mm_idx_gen(o, ref_fa.encode())); assert mi
    al = ctypes.c_void_p(L.mm_align_init(o, mi)); assert al, 'mm_align_init failed (no GPU?)'
    t_index = time.time() - t0
    L.mm_idx_occ.restype = ctypes.c_uint32; L.mm_idx_occ.argtypes = [ctypes.c_void_p, ctypes.c_uint32]; idx_occ = [int(L.mm_idx_occ(mi, i)) for i in range(3)]
    assert L.mm_align_devices(al) == n_inproc, 'the context spans %d device(s), %d were asked for' % (L.mm_align_devices(al), n_inproc)
    # this rank's shard: parts [p0, p1) of the set (PARTS is a multiple of every N the driver uses; otherwise the split is by parts, as even as it gets)
    p0, p1 = multi.shard_bounds(PARTS, rank, world)
    # the FASTA text of the shard in host memory (what a reader of the file finds in the page cache)
    t0 = time.time()
    sizes = [os.path.getsize(parts[p]) for p in range(p0, p1)]; text = bytearray(sum(sizes)); at = 0
    for p, sz in zip(range(p0, p1), sizes):
        with open(parts[p], 'rb') as f: assert f.readinto(memoryview(text)[at:at + sz]) == sz
        at += sz
    text_addr = ctypes.addressof((ctypes.c_char * max(1, len(text))).from_buffer(text)) if len(text) else 0
    t_load = time.time() - t0
    guess = L.mm_idx_max_len(mi)
    keep_bytes = 160 << 20          # text kept per step: enough for the identity check and for a spliced head window

    def sync():
        for d in range(min(n_inproc, torch.cuda.device_count())) if n_inproc > 1 else (local,): torch.cuda.synchronize(d)
        if dist: dist.barrier()
    def one_step(packed=None, reads=None, n_reads=0):
        if packed is not None: sm = multi.ShardMapper(L, al, reads, 0, n_reads, lanes=args.lanes, packed=packed, keep=keep_bytes, guess=0 if rank == 0 else guess)
        else: sm = multi.ShardMapper(L, al, None, 0, 0, lanes=args.lanes, keep=keep_bytes, guess=0 if rank == 0 else guess, text=(text_addr, len(text)))
        sm.map()
        sm.settle(dist, rank, world, 0, device)
        return sm
    T00 = time.time()
    def stage(what):
        if rank == 0: sys.stderr.write('[bench] %7.1f s  %s\n' % (time.time() - T00, what)); sys.stderr.flush()
    sm = None
    for _ in range(args.warmup): sm = one_step()
    stage('warmup done')
    L.mm_stats(al, None, 1)
    sync(); t0 = time.perf_counter()
    for _ in range(args.steps): sm = one_step()
    sync(); dt = time.perf_counter() - t0
    stage('timed steps done')
    st = Stats(); L.mm_stats(al, ctypes.byref(st), 0)
    sam_bytes = sm.col.total if sm else 0
    K = max(1, args.steps)
    bases = st.bases // K; n_reads = st.reads // K; nb = int(round(st.k1_launches / K))          # per step, counted by the library (re-runs of the carried value add a few launches)
    # ... and the earlier rounds' figure: the same steps from reads parsed and 2-bit packed ahead of time (N = 1 only: a report, not the value)
    from_packed = None; t_pack = None
    if n_gpus == 1 and not args.no_packed:
        tq = time.time()
        reads = ctypes.c_void_p(L.mm_reads_load(parts[p0].encode())) if p1 > p0 else None
        for p in range(p0 + 1, p1): assert L.mm_reads_append(reads, parts[p].encode()) == 0
        nr = L.mm_reads_count(reads) if reads else 0
        cap = 4096; arr = (ctypes.c_void_p * cap)()
        npk = L.mm_batch_pack_all(reads, 0, nr, arr, cap) if reads else 0
        packed = [arr[i] for i in range(npk)]; t_pack = time.time() - tq
        one_step(packed, reads, nr)
        sync(); tq = time.perf_counter()
        for _ in range(2): one_step(packed, reads, nr)
        sync(); from_packed = L.mm_reads_bases(reads, 0, nr) * 2 / (time.perf_counter() - tq) * 1e-9
        for h in packed: L.mm_batch_free(h)
        L.mm_reads_free(reads)
        sm_check = one_step()          # (the identity check below reads the head of the last text stream)
        sm = sm_check
    if dist:
        t = torch.tensor([dt], device=tdev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
        v = torch.tensor([float(bases), float(n_reads), float(sam_bytes), st.k1_ms, st.k2_ms, st.k3_ms, float(st.vectors), float(st.trace_steps), float(st.k3_launches), float(sm.stats['checks']), float(sm.stats['remapped_reads']), float(sm.stats['full_remaps'])], device=tdev, dtype=torch.float64)
        dist.all_reduce(v); tot = [float(x) for x in v.tolist()]
    else:
        tot = [float(bases), float(n_reads), float(sam_bytes), st.k1_ms, st.k2_ms, st.k3_ms, float(st.vectors), float(st.trace_steps), float(st.k3_launches), float(sm.stats['checks']), float(sm.stats['remapped_reads']), float(sm.stats['full_remaps'])]
    total_bases, total_reads, total_sam, k1_ms, k2_ms, k3_ms, vec, trs, k3_launches, n_checks, n_remap, n_full = tot

    if rank == 0:
        # the dominant kernel: algorithmic bytes (units counted by the kernel itself, SURVEY.md 8d per-unit figures) over its launch time (HIP events on the launch streams)
        alg_bytes = vec * 40.5 + trs * 32.0                      # all launches of the timed region, all ranks
        k3_launch_ms = k3_ms / max(1.0, k3_launches)
        achieved = (alg_bytes / max(1.0, k3_launches)) / (k3_launch_ms * 1e-3) / 1e9 if k3_ms > 0 else None
        out = {
            'metric': 'aligned Gbases/sec (whole node), map phase end to end: FASTA text of the reads in host memory -> SAM text in host memory',
            'value': total_bases * args.steps / dt * 1e-9, 'unit': 'Gbases/s', 'n_gpus': n_gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / K * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'i8',
            'data': 'synthetic (tools/gensim.c, seed 0x5eed0001: %.2f Mb reference in %d contig(s) with %g %% repeats; %s-like reads in %d parts)' % (w['genome_len'] / 1e6, w['contigs'], w['repeat_frac'] * 100, 'ONT' if w['kind'] == 'ont' else 'PBSIM-CLR', PARTS),
            'config': {'workload': '%s%s x %s x%g (%.2f Gb, %d reads) -x%s, one read set split over %d MI355X' % (w['name'], ' [custom shape]' if custom else '', 'ONT-like' if w['kind'] == 'ont' else 'PBSIM-like', w['depth'], total_bases / 1e9, int(total_reads), w['preset'], n_gpus),
                       'workload_key': args.workload if not custom else 'custom', 'reads_total': int(total_reads), 'bases_total': int(total_bases), 'batches_per_rank0': nb, 'lanes': args.lanes,
                       'parallelism': ('one process per GPU (torch.distributed.run, gloo): reads sharded contiguously, index replicated (no data-path collective; one all_gather of 2 integers per step for the carried value)' if world > 1 else
                                       'one process, %d device(s): batches of the one text dealt to device x lane by the library\'s streaming engine, index replicated, carried value verified in batch order, one ordered writer (no collective)%s' % (n_inproc, ' -- ALL DEVICE CONTEXTS ON ONE GPU (MM_BENCH_SAME_DEVICE): a test of the path, not a measurement' if same_box else '')),
                       'devices_in_process': n_inproc, 'processes': world,
                       'timed_region': 'text H2D + K0r record scan + K0 base conversion / 2-bit pack + K1 sketch/lookup + K2 sort/chain + K3 extension (rounds, carried-value verification) + D2H + post-map + SAM text; only the index build is outside',
                       'value_from_packed': from_packed, 'value_from_packed_note': 'the same steps from reads parsed and 2-bit packed on the host ahead of time (the timed region of rounds 1-2), 2 steps',
                       'device_only_gbases_per_s (sum of kernel time, lanes overlap)': total_bases * K / max(1e-9, (k1_ms + k2_ms + k3_ms) * 1e-3) * 1e-9 / 1.0,
                       'kernel_ms_per_step (summed over lanes and ranks)': {'sketch_seed': k1_ms / K, 'sort_chain': k2_ms / K, 'extend': k3_ms / K},
                       'host_ms_per_step (rank 0, summed over its threads\' critical paths)': {'d2h': st.host_post_ms / K, 'post_map_and_sam_text': st.host_sam_ms / K},
                       'extend_wave_time_split': ({k: getattr(st, 'k3_cycles_' + k) / max(1, st.k3_cycles_total) for k in ('fill', 'leaf', 'trace', 'next')} if st.k3_cycles_fill else 'not compiled in (libminialign_amd_prof.so through MM_LIB_OVERRIDE has it)'),
                       'extend_wave_balance (mean / max lifetime)': st.k3_cycles_total / max(1, st.k3_cycles_max * st.k3_waves),
                       'sort_chain_wave_time_split': {'sort_cycles_per_seed': st.k2_cycles_sort / max(1, st.seeds), 'chain_cycles_per_seed': st.k2_cycles_chain / max(1, st.seeds), 'seeds_per_read': st.seeds / max(1, st.reads), 'reads_not_in_lds': st.k2_reads_hbm},
                       'dp_vectors_per_base': vec / max(1.0, total_bases * K), 'trace_steps_per_base': trs / max(1.0, total_bases * K), 'reruns_per_step (rank 0)': st.reruns / K,
                       'pool_overflows (batches run again with larger device pools, rank 0, timed steps)': int(st.pool_grows), 'batch_splits (rank 0, timed steps)': int(st.batch_splits), 'sketch_launches_repeated_with_pools_sized_to_the_demand (rank 0, timed steps)': int(st.pool_regrows),
                       'extension_launches_called_off_by_the_watchdog (rank 0, timed steps)': int(st.k3_aborts),
                       'd2h_bytes_per_step (result pools + CIGAR text made on the device, rank 0)': st.d2h_bytes / K, 'cigar_text_bytes_per_step (made on the device, rank 0)': st.cigar_bytes_device / K,
                       'reader_gb_per_s (text to HBM, per uploader thread while it copies; one uploader per device)': (st.text_bytes * 1e-6 / st.reader_ms) if st.reader_ms > 0 else None,
                       'carried_value': {'checks': n_checks, 'remapped_reads': n_remap, 'full_remaps': n_full},
                       'sam_bytes_per_step': total_sam, 'generate_s': t_gen, 'index_build_s': t_index, 'index_occurrence_thresholds (minialign.c:2951: what -f makes of the reference)': idx_occ, 'text_load_s (outside)': t_load, 'host_parse_and_pack_s (value_from_packed only)': t_pack},
            'roofline': {'bound': 'hbm', 'kernel': 'mm_extend_kernel', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': (achieved / HBM_PEAK_GBS) if achieved else None, 'traffic': pmc_traffic(args.workload if not custom else 'custom', n_gpus, alg_bytes / max(1.0, k3_launches)),
                         'alg_bytes_per_launch': alg_bytes / max(1.0, k3_launches), 'avg_launch_ms': k3_launch_ms, 'launches': k3_launches,
                         # launches of different lanes share the chip, so one launch lasts longer than it would alone: the same bytes over the wall time of the timed region
                         'achieved_all_lanes': alg_bytes / dt * 1e-9 / n_gpus,
                         'note': 'the kernel is integer-VALU-issue bound, not HBM bound (DESIGN.md 4); achieved = algorithmic bytes per launch / mean launch time (HIP events on the launch streams, this run), achieved_all_lanes = per GPU over the wall time; traffic is NOT measured in this run: it is the HBM bytes per launch of the committed rocprofv3 PMC passes of the same workload (profiles/roundN_pmc.json, FETCH_SIZE / WRITE_SIZE in separate runs) scaled by the algorithmic bytes per launch of this run over those of that one',
                         'valu_issue': valu_position(vec / K, dt / K, n_gpus)},
        }
        if args.early_line: print(json.dumps(out), flush=True)
        if (n_gpus == 1 or args.check) and not args.no_cpu:
            cpu = reference_runs(w, ref_fa, parts, work, args.check_reads, args.baseline_reads, True)
            stage('CPU legs done')
            if n_gpus == 1: out['cpu_baseline'] = cpu['cpu_baseline']
            if cpu['check_sam'] is not None:
                # the records of the first check_reads reads of the stream: the library recorded where those of read check_reads begin (mm_head_offset)
                cut = L.mm_head_offset(al, cpu['check_reads']) if not sm._stale else multi.NO_OFFSET
                ours = sm.col.head(cut) if cut != multi.NO_OFFSET else None
                out['sam_identical'] = bool(ours is not None and ours == cpu['check_sam'])
                out['sam_check'] = 'records of the first %d reads (%d bytes) against %s' % (cpu['check_reads'], len(cpu['check_sam']), cpu['check_kind'])
            else:
                out['sam_identical'] = None; out['sam_check'] = 'not run'
        out['config'].update(cli_info)
        if hard_rec is not None: out['config']['hard_repeats'] = hard_rec
        print(json.dumps(out), flush=True)
    if dist: dist.barrier()
    if rank == 0:
        if args.keep: sys.stderr.write('[bench] data kept in %s\n' % work)
        else: shutil.rmtree(work, ignore_errors=True)
    if dist: dist.destroy_process_group()

if __name__ == '__main__':
    main()
