mkdir -p gpurun_out/r5
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["value"], d["unit"], d["ms_per_step"], "ms per step", "kernel ms summed", c["kernel_ms_per_step (summed over lanes and ranks)"], "balance", c["extend_wave_balance (mean / max lifetime)"], "batches", c["batches_per_rank0"], "k3 launches", d["roofline"]["launches"], "avg launch ms", d["roofline"]["avg_launch_ms"])'
B="timeout 600 python bench.py --steps 2 --warmup 1 --no-cli --no-packed --no-cpu"
for cfg in "2 1000000000 8" "2 1200000000 8" "3 800000000 8" "3 1000000000 8" "3 600000000 8" "3 1000000000 5" "2 800000000 8" "4 450000000 8"; do set -- $cfg
  echo "== $1 lanes, $2 bases per batch, $3 waves per SIMD"; MM_K3_WAVES_PER_SIMD=$3 MM_BATCH_BASES=$2 $B --lanes $1 2> gpurun_out/r5/c9.err | python3 -c "$P" || tail -5 gpurun_out/r5/c9.err
done
