P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["value"], d["unit"], d["ms_per_step"], "ms per step", "kernel ms summed", c["kernel_ms_per_step (summed over lanes and ranks)"], "balance", c["extend_wave_balance (mean / max lifetime)"], "reruns", c["reruns_per_step (rank 0)"])'
for cfg in "" "MM_K3_JOBS_BETWEEN_READS=1" ""; do echo "== hard: ${cfg:-default (retry jobs between reads)}"; env $cfg timeout 900 python bench.py --workload hg38hard --steps 2 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "$P"; done
B="timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu"
for cfg in "" "MM_K3_JOBS_BETWEEN_READS=1" "" "MM_K3_JOBS_BETWEEN_READS=1"; do echo "== headline: ${cfg:-default (retry jobs between reads)}"; env $cfg $B 2> /dev/null | python3 -c "$P"; done
echo "== dm6"; timeout 600 python bench.py --workload dm6 --steps 3 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "$P"
echo "== dm6 between=1"; MM_K3_JOBS_BETWEEN_READS=1 timeout 600 python bench.py --workload dm6 --steps 3 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "$P"
