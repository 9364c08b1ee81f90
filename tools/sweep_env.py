#!/usr/bin/env python3
"""sweep_env.py -- the headline workload once (data, index, context), then a list of environment settings, each timed over a few steps of mm_map_text.
usage: sweep_env.py [--workload hg38] [--steps 2] 'LANES=4' 'LANES=6 MM_SLAB_GB=96' ...   (LANES is the lanes argument, the rest environment)"""
import argparse, ctypes, os, sys, tempfile, time, shutil
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from minialign_amd import multi

def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--workload', default='hg38'); ap.add_argument('--steps', type=int, default=2); ap.add_argument('--depth', type=float); ap.add_argument('configs', nargs='+')
    a = ap.parse_args()
    w = dict(bench.WORKLOADS[a.workload])
    if a.depth: w['depth'] = a.depth
    work = tempfile.mkdtemp(prefix='mmsweep_')
    try:
        ref_fa, parts = bench.generate(work, w)
        L = multi.load_library(os.environ.get('MM_LIB_OVERRIDE')); assert L.mm_set_device(0) == 0
        o = ctypes.c_void_p(L.mm_opt_init()); argv = (ctypes.c_char_p * 4)(b'minialign', ('-x' + w['preset']).encode(), ref_fa.encode(), b'reads.fa'); files = (ctypes.c_char_p * 8)(); nf = ctypes.c_int(0)
        assert L.mm_opt_parse(o, 4, argv, files, 8, ctypes.byref(nf)) == 0
        mi = ctypes.c_void_p(L.mm_idx_gen(o, ref_fa.encode())); al = ctypes.c_void_p(L.mm_align_init(o, mi)); assert al
        text = bytearray()
        for p in parts:
            with open(p, 'rb') as f: text += f.read()
        addr = ctypes.addressof((ctypes.c_char * len(text)).from_buffer(text))
        bases = sum(1 for _ in ()) ; nb = None
        for cfg in a.configs:
            kv = dict(x.split('=', 1) for x in cfg.split()) if cfg.strip() else {}
            lanes = int(kv.pop('LANES', '4')); os.environ['MM_LANES'] = str(lanes)
            for k, v in kv.items(): os.environ[k] = v
            times = []
            for s in range(a.steps + 1):
                col = multi.Collector(keep=0); L.mm_align_set_carry(al, 0)
                t0 = time.perf_counter(); rc = L.mm_map_text(al, ctypes.c_void_p(addr), len(text), lanes, col.cb, None); dt = time.perf_counter() - t0
                assert rc == 0
                if s: times.append(dt)
            st = bench.Stats(); L.mm_stats(al, ctypes.byref(st), 1)
            per = st.bases / (a.steps + 1)
            print('%-70s  best %.3f s  mean %.3f s  %.2f Gbases/s (best)   kernels/step: sketch %.0f sort+chain %.0f extend %.0f ms' % (cfg, min(times), sum(times) / len(times), per / min(times) * 1e-9, st.k1_ms / (a.steps + 1), st.k2_ms / (a.steps + 1), st.k3_ms / (a.steps + 1)), flush=True)
            for k in kv: os.environ.pop(k, None)
    finally:
        shutil.rmtree(work, ignore_errors=True)
if __name__ == '__main__':
    main()
