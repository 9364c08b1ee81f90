mkdir -p gpurun_out/r5
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["value"], d["unit"], d["ms_per_step"], "ms per step", "kernel ms summed", c["kernel_ms_per_step (summed over lanes and ranks)"], "balance", c["extend_wave_balance (mean / max lifetime)"], "reruns", c["reruns_per_step (rank 0)"])'
B="timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu"
echo "== default (short re-runs)"; $B 2> /dev/null | python3 -c "$P"
echo "== MM_SLOW_RERUN=1"; MM_SLOW_RERUN=1 $B 2> /dev/null | python3 -c "$P"
for w in 6 7 8; do echo "== $w waves per SIMD per launch"; MM_K3_WAVES_PER_SIMD=$w $B 2> /dev/null | python3 -c "$P"; done
echo "== 8 waves, 3 lanes"; MM_K3_WAVES_PER_SIMD=8 $B --lanes 3 2> /dev/null | python3 -c "$P"
echo "== 8 waves, 6 lanes"; GPU_MAX_HW_QUEUES=32 MM_K3_WAVES_PER_SIMD=8 $B --lanes 6 2> /dev/null | python3 -c "$P"
echo "== 8 waves, helpers 1 in 32"; MM_K3_HELPERS=32 MM_K3_WAVES_PER_SIMD=8 $B 2> /dev/null | python3 -c "$P"
echo "== default again"; $B 2> /dev/null | python3 -c "$P"
echo "== hard default"; timeout 900 python bench.py --steps 1 --warmup 1 --no-cli --no-packed --no-cpu --workload hg38hard --depth 0.3 2> /dev/null | python3 -c "$P"
