#!/bin/bash
# Where the waves of the kernels spend their cycles: rocprofv3 SQ counters (one pass of 8 SQ counters + GRBM_GUI_ACTIVE, kernel trace only) over a short bench run
# with one lane, so that every kernel runs alone.  Usage (on the GPU box): tools/pmc_sq.sh <out.json> [bench args]
OUT=${1:-gpurun_out/pmc_sq.json}; shift; D=$(mktemp -d /tmp/pmcsq.XXXX); cd "$(dirname "$0")/.."; R=$PWD
export TMPDIR=/tmp
ARGS=${*:---depth 0.3 --lanes 1 --steps 1 --warmup 0 --no-cpu --no-cli --no-packed}
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA"
C2="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
i=0
for c in "$C1" "$C2"; do i=$((i+1))
	( cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $D/p$i -o p -- python $R/bench.py $ARGS > $D/p$i.log 2>&1 ) || tail -5 $D/p$i.log
done
python3 - "$D" "$OUT" "$ARGS" <<'PY'
import csv, glob, json, sys
d, out, args = sys.argv[1], sys.argv[2], sys.argv[3]
res = {}
for fn in glob.glob('%s/p*/**/*counter_collection.csv' % d, recursive=True):
    for row in csv.DictReader(open(fn)):
        k = row['Kernel_Name'].split('(')[0]; c = row['Counter_Name']
        e = res.setdefault(k, {}).setdefault(c, {'sum': 0.0, 'dispatches': 0})
        e['sum'] += float(row['Counter_Value']); e['dispatches'] += 1
summ = {}
for k, cs in res.items():
    g = lambda n: cs.get(n, {}).get('sum', 0.0)
    wc = g('SQ_WAVE_CYCLES')
    if not wc: continue
    summ[k] = {'dispatches': cs['SQ_WAVE_CYCLES']['dispatches'],
               'wave_cycles_quad': wc,
               'frac_wait_any(s_waitcnt, barrier)': g('SQ_WAIT_ANY') / wc, 'frac_wait_inst_any(issue stall)': g('SQ_WAIT_INST_ANY') / wc, 'frac_active_inst_any': g('SQ_ACTIVE_INST_ANY') / wc,
               'frac_active_valu': g('SQ_ACTIVE_INST_VALU') / wc, 'frac_active_scalar': g('SQ_ACTIVE_INST_SCA') / wc,
               'valu_insts': g('SQ_INSTS_VALU'), 'salu_insts': g('SQ_INSTS_SALU'), 'vmem_rd': g('SQ_INSTS_VMEM_RD'), 'vmem_wr': g('SQ_INSTS_VMEM_WR'), 'lds': g('SQ_INSTS_LDS'), 'smem': g('SQ_INSTS_SMEM'),
               'waves': g('SQ_WAVES'), 'busy_cycles': g('SQ_BUSY_CYCLES'), 'gui_active': g('GRBM_GUI_ACTIVE')}
# the DP vectors of the same run (bench.py's JSON line in the log of the first pass): VALU time per vector, which bench.py turns into the VALU-issue position of the runs it times
ext = {}
try:
    line = [l for l in open('%s/p1.log' % d) if l.startswith('{')][-1]; b = json.loads(line)
    vec = b['config']['dp_vectors_per_base'] * b['config']['bases_total'] * b['steps']
    k3 = summ['mm::mm_extend_kernel']
    ext = {'dp_vectors': vec, 'valu_busy_cycles_per_dp_vector': k3['frac_active_valu'] * k3['wave_cycles_quad'] * 4 / vec, 'scalar_busy_cycles_per_dp_vector': k3['frac_active_scalar'] * k3['wave_cycles_quad'] * 4 / vec,
           'valu_insts_per_dp_vector': k3['valu_insts'] / vec, 'salu_insts_per_dp_vector': k3['salu_insts'] / vec,
           'note': 'SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_SCA count quad-cycles summed over waves (MI355X_MICROARCH.md); x 4 = SIMD cycles the pipe was held; everything mm_extend_kernel does (fill, traceback, driver) over its DP vectors'}
except Exception as e:
    ext = {'error': str(e)}
json.dump({'command': 'rocprofv3 --kernel-trace --pmc <two passes of SQ counters> --output-format csv -- python bench.py ' + args, 'mm_extend_kernel_per_dp_vector': ext, 'per_kernel': summ, 'raw': res}, open(out, 'w'), indent=1)
print('mm_extend_kernel per DP vector:', ext)
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]['wave_cycles_quad'])[:6]:
    print(k, {a: (round(b, 3) if isinstance(b, float) and b < 10 else b) for a, b in v.items()})
PY
