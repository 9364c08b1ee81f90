mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_multi_gpu.py -x -q -k "alternative_schedules or replicas or ladder" > gpurun_out/r5/c3_multi.log 2>&1; echo "multi rc=$?"; tail -5 gpurun_out/r5/c3_multi.log
timeout 600 python -m pytest tests/test_hard_gpu.py -x -q > gpurun_out/r5/c3_hard.log 2>&1; echo "hard rc=$?"; tail -5 gpurun_out/r5/c3_hard.log
for cfg in "" "MM_K3_DYN_ROUND0=2" "MM_K3_NO_ROUND_JOBS=1"; do
  echo "== hard: ${cfg:-default}"
  env $cfg timeout 900 python bench.py --steps 1 --warmup 1 --no-cli --no-packed --no-cpu --workload hg38hard --depth 0.3 2> gpurun_out/r5/c3_hard_bench.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms per step', 'dp vectors per base', d['config']['dp_vectors_per_base'], 'extend ms summed', d['config']['kernel_ms_per_step (summed over lanes and ranks)']['extend'], 'balance', d['config']['extend_wave_balance (mean / max lifetime)'])"
done
for cfg in "" "MM_K3_DYN_ROUND0=2"; do
  echo "== headline: ${cfg:-default}"
  env $cfg timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms per step', 'extend ms summed', d['config']['kernel_ms_per_step (summed over lanes and ranks)']['extend'], 'balance', d['config']['extend_wave_balance (mean / max lifetime)'])"
done
