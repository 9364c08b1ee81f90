#!/bin/bash
# The profiles of a round, on the GPU box: (1) the default bench line with its CPU legs, (2) rocprofv3 --kernel-trace --stats of the same command (without the CPU
# legs) and how the kernels share the device, (3) one batch at a time on one lane (what each launch takes alone), (4) the per-phase split of the extension
# kernel from the profiling build, (5) HBM traffic from the PMC passes, (6) where the waves of each kernel spend their cycles (SQ counters), (7) what the lanes did batch by batch.  Usage: tools/round_profiles.sh <outdir> <tag>
OUT=${1:-gpurun_out/round}; TAG=${2:-round3}; R=$PWD; mkdir -p "$OUT"; export TMPDIR=/tmp
python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof_default" -o p --output-format csv -- python "$R/bench.py" --steps 1 --warmup 1 --no-cpu --no-cli --no-packed > "$R/$OUT/${TAG}_bench_under_rocprof.json" 2> /dev/null )
cp "$OUT"/prof_default/*kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv" 2> /dev/null || cp $(find "$OUT/prof_default" -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_kernel_stats.csv"
python3 tools/trace_concurrency.py $(find "$OUT/prof_default" -name "*kernel_trace.csv" | head -1) > "$OUT/${TAG}_kernel_concurrency.txt"
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$R/$OUT/prof_single" -o p --output-format csv -- python "$R/bench.py" --depth 0.095 --lanes 1 --steps 2 --warmup 1 --no-cpu --no-cli --no-packed > "$R/$OUT/${TAG}_bench_one_batch_one_lane.json" 2> /dev/null )
cp $(find "$OUT/prof_single" -name "*kernel_stats.csv" | head -1) "$OUT/${TAG}_kernel_stats_one_batch_one_lane.csv"
python3 - $(find "$OUT/prof_single" -name "*kernel_trace.csv" | head -1) > "$OUT/${TAG}_kernel_timeline_one_batch_one_lane.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'sketch' in r['Kernel_Name']]; t0 = int(rows[idx[-1]]['Start_Timestamp'])
print('# one 295 Mb batch of the headline workload alone on the device (last step of bench.py --depth 0.095 --lanes 1): launches over 0.2 ms')
for r in rows[idx[-1]:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if d > 0.2 and 'copyBuffer' not in r['Kernel_Name'] and 'fillBuffer' not in r['Kernel_Name']: print('%9.2f ms  +%8.2f ms  %-44s grid %s' % ((int(r['Start_Timestamp']) - t0) / 1e6, d, r['Kernel_Name'][:44], r['Grid_Size_X']))
PY
MM_LIB_OVERRIDE=$R/minialign_amd/libminialign_amd_prof.so python bench.py --steps 1 --warmup 1 --no-cpu --no-cli --no-packed > "$OUT/${TAG}_bench_profiling_build.json" 2> /dev/null
bash tools/pmc_traffic.sh "$OUT/${TAG}_pmc.json" > /dev/null 2>&1
bash tools/pmc_sq.sh "$OUT/${TAG}_pmc_sq.json" > "$OUT/${TAG}_pmc_sq.txt" 2>&1
MM_VERBOSE=1 python bench.py --steps 2 --warmup 1 --no-cpu --no-cli --no-packed > /dev/null 2> "$OUT/verbose.err"; python3 tools/lane_trace.py "$OUT/verbose.err" > "$OUT/${TAG}_lane_trace.txt"; rm -f "$OUT/verbose.err"
rm -rf "$OUT/prof_default" "$OUT/prof_single"
ls -la "$OUT"
