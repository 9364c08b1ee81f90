#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of rocprofv3 against kernels of known HBM traffic (tools/ubench/traffic_cal.hip), separate --pmc passes, kernel trace only.
# Usage (GPU box): tools/pmc_calibrate.sh <out.json>
OUT=${1:-gpurun_out/pmc_calibration.json}; D=$(mktemp -d /tmp/pmccal.XXXX); cd "$(dirname "$0")/.."; R=$PWD; export TMPDIR=/tmp
[ -x tools/ubench/traffic_cal ] || hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o tools/ubench/traffic_cal tools/ubench/traffic_cal.hip || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
	( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/$c -o p -- $R/tools/ubench/traffic_cal 8 > $D/$c.log 2>&1 )
done
python3 - "$D" "$OUT" <<'PY'
import csv, glob, json, sys
d, out = sys.argv[1], sys.argv[2]
known = json.loads([l for l in open('%s/FETCH_SIZE.log' % d) if l.startswith('{')][-1])
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for fn in glob.glob('%s/%s/**/*counter_collection.csv' % (d, c), recursive=True):
        for row in csv.DictReader(open(fn)):
            if row.get('Counter_Name') != c: continue
            k = row['Kernel_Name'].split('(')[0]
            res.setdefault(k, {}).setdefault(c, 0.0); res[k][c] += float(row['Counter_Value']) * 1024          # the counters are in KB
cal = {}
def ratio(k, c, what): 
    v = res.get(k, {}).get(c); return None if v is None else v / known[k][what]
cal['stream_read:  FETCH_SIZE / bytes read'] = ratio('stream_read', 'FETCH_SIZE', 'read')
cal['strided32_read: FETCH_SIZE / useful bytes (32 B per item)'] = ratio('strided32_read', 'FETCH_SIZE', 'useful')
cal['strided32_read: FETCH_SIZE / 64-B granules touched'] = ratio('strided32_read', 'FETCH_SIZE', 'granule64')
cal['strided32_read: FETCH_SIZE / 128-B granules touched'] = ratio('strided32_read', 'FETCH_SIZE', 'granule128')
cal['stream_write: WRITE_SIZE / bytes written'] = ratio('stream_write', 'WRITE_SIZE', 'written')
cal['block_write:  WRITE_SIZE / bytes written (1 296-B blocks)'] = ratio('block_write', 'WRITE_SIZE', 'written')
cal['block_write:  FETCH_SIZE / bytes written (read for partial lines)'] = ratio('block_write', 'FETCH_SIZE', 'written')
cal['scratch_toy:  WRITE_SIZE / scratch bytes stored'] = ratio('scratch_toy', 'WRITE_SIZE', 'written')
cal['scratch_toy:  FETCH_SIZE / scratch bytes loaded'] = ratio('scratch_toy', 'FETCH_SIZE', 'read')
json.dump({'command': 'rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- tools/ubench/traffic_cal 8 (separate passes)', 'known_bytes': known, 'counter_bytes (raw, x 1024)': res, 'counter_over_known': cal}, open(out, 'w'), indent=1)
for k, v in cal.items(): print('%-70s %s' % (k, 'n/a' if v is None else '%.3f' % v))
PY
