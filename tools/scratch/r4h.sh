mkdir -p gpurun_out/r4h
for L in 6 8 5; do
MM_VERBOSE=1 timeout 300 python bench.py --workload ont --lanes $L --steps 3 --warmup 1 --no-cli --no-packed --no-cpu > gpurun_out/r4h/ont$L.json 2> gpurun_out/r4h/ont$L.err; echo "lanes $L rc=$?"; grep "device memory" gpurun_out/r4h/ont$L.err | tail -1; python -c "
import json; d=json.load(open('gpurun_out/r4h/ont$L.json')); c=d['config']; print('lanes $L: %.3f Gb/s %.0f ms/step, batches %d, balance %.3f, k3 launch %.1f ms' % (d['value'], d['ms_per_step'], c['batches_per_rank0'], c['extend_wave_balance (mean / max lifetime)'], d['roofline']['avg_launch_ms']))"
done
