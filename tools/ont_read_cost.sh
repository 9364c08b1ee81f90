#!/bin/bash
# Per-read cost of the extension kernel on the ONT-like hg38-size set, batch by batch (profiling build: wave time per read, its fill / traceback parts, the wait for a DP
# workspace, when the read started).  Usage (GPU box): tools/ont_read_cost.sh <outfile>
OUT=${1:-gpurun_out/ont_read_cost.txt}; D=$(mktemp -d)
MM_LIB_OVERRIDE=$PWD/minialign_amd/libminialign_amd_prof.so MM_DUMP_READ_COST=$D/cost.tsv MM_VERBOSE=1 MM_VERBOSE_SLABS=1 timeout 600 python bench.py --workload ont --steps 1 --warmup 1 --no-cli --no-packed --no-cpu > $D/ont.json 2> $D/ont.err
{
echo "# bench.py --workload ont --steps 1 --warmup 1 with the profiling build and MM_DUMP_READ_COST: the second stream (warm), one table per batch (tools/read_cost.py)."
echo "# Times are s_memtime cycles at 2.1 GHz; the counters of the eight XCDs are not synchronised, so start / end are only comparable within an XCD (the clusters of 'start ms')."
grep "workspace class" $D/ont.err | tail -5
grep "run " $D/ont.err | tail -5
n=$(ls $D/cost.tsv* | wc -l); h=$((n / 2))
for f in $(ls -v $D/cost.tsv* | tail -$h); do echo; echo "== batch file $(basename $f)"; python tools/read_cost.py $f; done
} > "$OUT"
rm -rf $D
