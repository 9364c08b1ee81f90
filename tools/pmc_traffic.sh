#!/bin/bash
# HBM traffic of the kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, kernel trace only), summed per kernel and
# written as JSON.  Usage (on the GPU box): tools/pmc_traffic.sh <out.json>
OUT=${1:-gpurun_out/pmc.json}; D=$(mktemp -d /tmp/pmc.XXXX); cd "$(dirname "$0")/.."; R=$PWD
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
	( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/$c -o p -- python $R/bench.py --depth 0.3 --lanes 1 --steps 1 --warmup 0 --no-cpu --no-cli --no-packed > $D/$c.log 2>&1 )
done
python3 - "$D" "$OUT" <<'PY'
import csv, glob, json, sys
d, out = sys.argv[1], sys.argv[2]
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for fn in glob.glob('%s/%s/**/*counter_collection.csv' % (d, c), recursive=True):
        for row in csv.DictReader(open(fn)):
            if row.get('Counter_Name') != c: continue
            k = row['Kernel_Name'].split('(')[0]
            e = res.setdefault(k, {}).setdefault(c, {'sum': 0.0, 'dispatches': 0})
            e['sum'] += float(row['Counter_Value']); e['dispatches'] += 1
k3 = res.get('mm::mm_extend_kernel', {})
f = k3.get('FETCH_SIZE', {'sum': 0, 'dispatches': 1}); w = k3.get('WRITE_SIZE', {'sum': 0, 'dispatches': 1})
per = {'FETCH_SIZE_KB_raw': f['sum'] / max(f['dispatches'], 1), 'WRITE_SIZE_KB_raw': w['sum'] / max(w['dispatches'], 1)}
per['fetch_bytes_corrected'] = per['FETCH_SIZE_KB_raw'] * 1024 * 2      # MI355X_MICROARCH.md, HBM section: gfx950 tallies 128-B read requests at 64 B
per['write_bytes'] = per['WRITE_SIZE_KB_raw'] * 1024
per['hbm_bytes'] = per['fetch_bytes_corrected'] + per['write_bytes']
# the algorithmic bytes of a launch of the same run (bench.py's own figure, from its JSON line in the log of the first pass): bench.py scales the traffic to the launches it times
try:
    line = [l for l in open('%s/FETCH_SIZE.log' % d) if l.startswith('{')][-1]; per['alg_bytes_per_launch'] = json.loads(line)['roofline']['alg_bytes_per_launch']
    per['traffic_over_algorithmic'] = per['hbm_bytes'] / per['alg_bytes_per_launch']
except Exception as e:
    per['alg_bytes_per_launch'] = None
per['correction'] = 'MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B -> doubled; WRITE_SIZE taken as is'
json.dump({'command': 'rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> --output-format csv -- python bench.py --depth 0.3 --lanes 1 --steps 1 --warmup 0 --no-cpu --no-cli --no-packed (separate passes)',
           'workload': 'hg38', 'note': 'the headline reference (3.1 Gb, 25 contigs) with a tenth of the headline reads (depth 0.3 instead of 3: four batches of 233 Mb): per launch of mm_extend_kernel, averaged over the round-0 and the rescue-round launches in the same 1 : 2 mix as the full run',
           'counters': res, 'mm_extend_kernel_per_launch': per}, open(out, 'w'), indent=1)
print(json.dumps(per))
PY
