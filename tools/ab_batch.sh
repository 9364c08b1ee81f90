mkdir -p gpurun_out/r3z
for B in 480000000 640000000 800000000; do
  MM_BATCH_BASES=$B MM_VERBOSE=1 timeout 250 python bench.py --workload ont --steps 3 --warmup 1 --no-cli --no-packed --no-cpu > gpurun_out/r3z/ont_$B.json 2> gpurun_out/r3z/ont_$B.err; echo "batch $B rc=$?"
  grep "device memory" gpurun_out/r3z/ont_$B.err | tail -1
  python -c "
import json; d=json.load(open('gpurun_out/r3z/ont_$B.json')); c=d['config']; print('  %.3f Gb/s %.0f ms/step, batches %d, balance %.3f, k3 launch %.1f ms' % (d['value'], d['ms_per_step'], c['batches_per_rank0'], c['extend_wave_balance (mean / max lifetime)'], d['roofline']['avg_launch_ms']))"
done
