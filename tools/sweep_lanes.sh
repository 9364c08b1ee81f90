#!/bin/bash
# bench.py over lanes / batch sizes / extension waves per SIMD on the headline workload (no CPU legs): one line per setting.  Usage: tools/sweep_lanes.sh <outfile>
OUT=${1:-gpurun_out/sweep.txt}; : > "$OUT"
run() { desc="$1"; shift; r=$(env "$@" python bench.py --steps 3 --warmup 1 --no-cpu --no-cli --no-packed --lanes ${LANES:-4} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; k=c['kernel_ms_per_step (summed over lanes and ranks)']
print('%.3f Gb/s  %.0f ms/step  k1 %.0f k2 %.0f k3 %.0f  k3 launch %.1f ms  valu %.3f' % (d['value'], d['ms_per_step'], k['sketch_seed'], k['sort_chain'], k['extend'], d['roofline']['avg_launch_ms'], d['roofline']['valu_issue']['frac_of_wall']))"); echo "$desc: $r" | tee -a "$OUT"; }
LANES=4 run "lanes 4" A=1
LANES=5 run "lanes 5" A=1
LANES=6 run "lanes 6" A=1
LANES=8 run "lanes 8" A=1
LANES=4 run "lanes 4, 200 Mb batches" MM_BATCH_BASES=200000000
LANES=6 run "lanes 6, 200 Mb batches" MM_BATCH_BASES=200000000
LANES=4 run "lanes 4, 8 ext waves per SIMD" MM_K3_WAVES_PER_SIMD=8
LANES=6 run "lanes 6, 4 ext waves per SIMD" MM_K3_WAVES_PER_SIMD=4
