mkdir -p gpurun_out/r5
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["value"], d["unit"], d["ms_per_step"], "ms per step", "kernel ms summed", c["kernel_ms_per_step (summed over lanes and ranks)"], "balance", c["extend_wave_balance (mean / max lifetime)"])'
for l in 4 6 8; do
echo "== $l lanes"; GPU_MAX_HW_QUEUES=32 MM_VERBOSE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cli --no-packed --no-cpu --lanes $l 2> gpurun_out/r5/c19.err | python3 -c "$P"
python3 tools/k3_overlap.py gpurun_out/r5/c19.err 2>&1 | tail -2; python3 tools/lane_trace.py gpurun_out/r5/c19.err | tail -1
done
rm -f gpurun_out/r5/c19.err
