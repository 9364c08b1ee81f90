mkdir -p gpurun_out/r5
B="python bench.py --steps 1 --warmup 1 --no-cli --no-packed --no-cpu"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["value"], d["unit"], d["ms_per_step"], "ms per step", "dp vectors per base", c["dp_vectors_per_base"], "kernel ms summed", c["kernel_ms_per_step (summed over lanes and ranks)"], "balance", c["extend_wave_balance (mean / max lifetime)"], "overflows", c["pool_overflows (batches run again with larger device pools, rank 0, timed steps)"], "reruns", c["reruns_per_step (rank 0)"], "batches", c["batches_per_rank0"])'
for cfg in "" "MM_K3_DYN_ROUND0=2" "MM_K3_RESCUE_FIRST=1" "MM_K3_DYN_ROUND0=2 MM_K3_RESCUE_FIRST=1"; do
  echo "== hard: ${cfg:-default}"
  env $cfg MM_VERBOSE=1 timeout 900 $B --workload hg38hard --depth 0.3 2> gpurun_out/r5/c4_hard_verbose.err | python3 -c "$P"
  python3 tools/lane_trace.py gpurun_out/r5/c4_hard_verbose.err > "gpurun_out/r5/c4_hard_lane_trace_$(echo ${cfg:-default} | tr ' =' '__').txt" 2>&1
done
grep -v "^\[minialign_amd\] batch\|^\[minialign_amd\]   " gpurun_out/r5/c4_hard_verbose.err | head -60 > gpurun_out/r5/c4_hard_verbose_rest.txt
echo "== headline verbose"
MM_VERBOSE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cli --no-packed --no-cpu 2> gpurun_out/r5/c4_head_verbose.err | python3 -c "$P"
python3 tools/lane_trace.py gpurun_out/r5/c4_head_verbose.err > gpurun_out/r5/c4_head_lane_trace.txt 2>&1; tail -3 gpurun_out/r5/c4_head_lane_trace.txt
rm -f gpurun_out/r5/c4_head_verbose.err
echo "== headline 6 lanes, 32 hardware queues"
GPU_MAX_HW_QUEUES=32 timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu --lanes 6 2> /dev/null | python3 -c "$P"
echo "== headline 4 lanes, 32 hardware queues"
GPU_MAX_HW_QUEUES=32 timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "$P"
