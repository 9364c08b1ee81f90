#!/bin/bash
# The other BASELINE shapes and the shards of the headline set through bench.py (same timed region as the default line), one summary line each.
# Usage (GPU box): tools/workloads.sh <outfile>
OUT=${1:-gpurun_out/workloads.txt}; : > "$OUT"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']; cb=d.get('cpu_baseline') or {}
print('%s\n       %.2f G bases/s end to end (%.0f ms per step, FASTA text -> SAM text), records of the first reads identical to the reference: %s; cpu %s; DP vectors per base %.2f; extension launch %.1f ms x %d, roofline frac %.3f' % (c['workload'], d['value'], d['ms_per_step'], d.get('sam_identical'), ('%.2f G bases/s on %d threads' % (cb['value'], cb['cores'])) if cb.get('value') else 'not run', c['dp_vectors_per_base'], r['avg_launch_ms'], int(r['launches']), r['frac']))"; }
echo "# python bench.py --workload <dm6 | ecoli | ont> --steps 3 --warmup 1 on one MI355X (round 6 tree): the other BASELINE shapes through the same timed region as the default (hg38) line" >> "$OUT"
for W in dm6 ecoli ont hg38hard; do timeout 900 python bench.py --workload $W --steps 3 --warmup 1 --no-cli --no-packed 2>/dev/null | line >> "$OUT"; done
echo >> "$OUT"; echo "Shards of the headline set as the ranks of a multi-GPU job see them (bench.py --depth d --no-cpu, one MI355X, 4 lanes; not a multi-GPU measurement):" >> "$OUT"
for D in 1.5 0.75 0.375; do python bench.py --depth $D --steps 4 --warmup 1 --no-cpu --no-cli --no-packed 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('  --depth %-6s %.2f Gb  %5.0f ms per step  %.2f G bases/s' % ('$D', c['bases_total'] / 1e9, d['ms_per_step'], d['value']))" >> "$OUT"; done
cat "$OUT"
