mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_multi_gpu.py tests/test_zz_bench_devices_gpu.py tests/test_hard_gpu.py tests/test_mm_gpu.py -x -q -k "alternative_schedules or replicas or ladder or zz or hard or test_mm" > gpurun_out/r5/c5_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r5/c5_tests.log
B="python bench.py --steps 1 --warmup 1 --no-cli --no-packed --no-cpu"
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["value"], d["unit"], d["ms_per_step"], "ms per step", "dp vectors per base", c["dp_vectors_per_base"], "kernel ms summed", c["kernel_ms_per_step (summed over lanes and ranks)"], "balance", c["extend_wave_balance (mean / max lifetime)"], "overflows", c["pool_overflows (batches run again with larger device pools, rank 0, timed steps)"], "reruns", c["reruns_per_step (rank 0)"], "batches", c["batches_per_rank0"])'
for cfg in "" "MM_K3_DYN_ROUND0=2" "MM_NO_CARRY_DEPS=1"; do
  echo "== hard: ${cfg:-default}"
  env $cfg MM_VERBOSE=1 timeout 900 $B --workload hg38hard --depth 0.3 2> gpurun_out/r5/c5_hard_verbose.err | python3 -c "$P"
  python3 tools/lane_trace.py gpurun_out/r5/c5_hard_verbose.err > "gpurun_out/r5/c5_hard_lane_trace_$(echo ${cfg:-default} | tr ' =' '__').txt" 2>&1
done
rm -f gpurun_out/r5/c5_hard_verbose.err
for cfg in "" "MM_NO_CARRY_DEPS=1"; do
echo "== headline ${cfg:-default}"
env $cfg MM_VERBOSE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu 2> gpurun_out/r5/c5_head_verbose.err | python3 -c "$P"
python3 tools/lane_trace.py gpurun_out/r5/c5_head_verbose.err > gpurun_out/r5/c5_head_lane_trace_$(echo ${cfg:-default} | tr ' =' '__').txt 2>&1; tail -1 gpurun_out/r5/c5_head_lane_trace_$(echo ${cfg:-default} | tr ' =' '__').txt
done
rm -f gpurun_out/r5/c5_head_verbose.err
