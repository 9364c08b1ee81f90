mkdir -p gpurun_out/r3x
run() { # name, env..., -- args
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu $ARGS > gpurun_out/r3x/$n.json 2> gpurun_out/r3x/$n.err; echo "$n rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3x/$n.json")); c=d["config"]; print("  $n: %.3f Gb/s %.0f ms/step  balance %.3f  k3 launch %.1f ms  kernels %s" % (d["value"], d["ms_per_step"], c["extend_wave_balance (mean / max lifetime)"], d["roofline"]["avg_launch_ms"], {k: round(v) for k, v in c["kernel_ms_per_step (summed over lanes and ranks)"].items()}))
except Exception as e: print("  $n: no result", e)
PY
}
ARGS="--workload ont"; run ont_early X=1; run ont_late MM_K3_LATE_HELPERS=1; run ont_early32 MM_K3_HELPERS=32
ARGS=""; run hg_early X=1; run hg_late MM_K3_LATE_HELPERS=1; run hg_early32 MM_K3_HELPERS=32
