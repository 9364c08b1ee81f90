P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["value"], d["unit"], d["ms_per_step"], "ms per step", "kernel ms summed", c["kernel_ms_per_step (summed over lanes and ranks)"], "balance", c["extend_wave_balance (mean / max lifetime)"], "batches", c["batches_per_rank0"], "k3 launches", d["roofline"]["launches"], "avg launch ms", d["roofline"]["avg_launch_ms"])'
B="timeout 600 python bench.py --steps 2 --warmup 1 --no-cli --no-packed --no-cpu"
echo "== 1 lane, 300 Mb batches (8 waves per SIMD: a lone lane)"; $B --lanes 1 2> /dev/null | python3 -c "$P"
echo "== 1 lane, 1 Gb batches"; MM_BATCH_BASES=1000000000 $B --lanes 1 2> /dev/null | python3 -c "$P"
echo "== 1 lane, 2.4 Gb batches"; MM_BATCH_BASES=2400000000 $B --lanes 1 2> gpurun_out/r5/c8.err | python3 -c "$P" || tail -5 gpurun_out/r5/c8.err
echo "== 2 lanes, 1.2 Gb batches"; MM_BATCH_BASES=1200000000 $B --lanes 2 2> /dev/null | python3 -c "$P"
echo "== 4 lanes, 600 Mb batches"; MM_BATCH_BASES=600000000 $B --lanes 4 2> /dev/null | python3 -c "$P"
echo "== 4 lanes, 600 Mb batches, 8 waves"; MM_K3_WAVES_PER_SIMD=8 MM_BATCH_BASES=600000000 $B --lanes 4 2> /dev/null | python3 -c "$P"
