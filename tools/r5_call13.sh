mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_multi_gpu.py tests/test_zz_bench_devices_gpu.py tests/test_hard_gpu.py -x -q -k "alternative_schedules or replicas or ladder or zz or hard" > gpurun_out/r5/c13_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5/c13_tests.log
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["value"], d["unit"], d["ms_per_step"], "ms per step", "kernel ms summed", c["kernel_ms_per_step (summed over lanes and ranks)"], "balance", c["extend_wave_balance (mean / max lifetime)"], "reruns", c["reruns_per_step (rank 0)"])'
B="timeout 600 python bench.py --steps 4 --warmup 1 --no-cli --no-packed --no-cpu"
for cfg in "" "MM_K3_JOBS_BETWEEN_READS=0 MM_K3_NO_ROUND_JOBS=1 MM_NO_CARRY_DEPS=1" "" "MM_K3_JOBS_BETWEEN_READS=0 MM_K3_NO_ROUND_JOBS=1 MM_NO_CARRY_DEPS=1" "MM_K3_NO_ROUND_JOBS=1"; do
  echo "== headline: ${cfg:-default}"; env $cfg $B 2> /dev/null | python3 -c "$P"
done
B="timeout 900 python bench.py --workload ont --steps 2 --warmup 1 --no-cli --no-packed --no-cpu"
for cfg in "" "MM_K3_JOBS_BETWEEN_READS=0" ""; do
  echo "== ont: ${cfg:-default}"; env $cfg $B 2> /dev/null | python3 -c "$P"
done
echo "== hard"; timeout 900 python bench.py --workload hg38hard --steps 2 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "$P"
echo "== hard, no jobs between reads"; MM_K3_JOBS_BETWEEN_READS=0 timeout 900 python bench.py --workload hg38hard --steps 2 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "$P"
