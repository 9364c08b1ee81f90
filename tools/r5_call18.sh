P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["value"], d["unit"], d["ms_per_step"], "ms per step", "kernel ms summed", c["kernel_ms_per_step (summed over lanes and ranks)"], "balance", c["extend_wave_balance (mean / max lifetime)"], "reruns", c["reruns_per_step (rank 0)"])'
B="timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu"
for cfg in "" "MM_K3_DYN_ROUND0=2" "MM_K3_DYN_ROUND0=3" "MM_K3_DYN_ROUND0=5" "" "MM_K3_DYN_ROUND0=2 MM_K3_NO_JOBS=1" "MM_K3_HELPERS=32" "MM_K3_DYN_ROUND0=3 MM_K3_HELPERS=32"; do
  echo "== headline: ${cfg:-default}"; env $cfg $B 2> /dev/null | python3 -c "$P"
done
