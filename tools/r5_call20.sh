P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["value"], d["unit"], d["ms_per_step"], "ms per step", "kernel ms summed", c["kernel_ms_per_step (summed over lanes and ranks)"], "balance", c["extend_wave_balance (mean / max lifetime)"])'
B="timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu"
for cfg in "4 0" "6 3" "6 2" "8 3" "8 2" "5 3" "4 3"; do set -- $cfg
echo "== $1 lanes, at most $2 extension launches in flight"; if [ $2 = 0 ]; then GPU_MAX_HW_QUEUES=32 $B --lanes $1 2> /dev/null | python3 -c "$P"; else GPU_MAX_HW_QUEUES=32 MM_K3_CONCURRENT=$2 $B --lanes $1 2> /dev/null | python3 -c "$P"; fi
done
