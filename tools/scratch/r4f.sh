mkdir -p gpurun_out/r4f
for W in ont hg38; do
MM_VERBOSE=1 timeout 300 python bench.py --workload $W --steps 3 --warmup 1 --no-cli --no-packed --no-cpu > gpurun_out/r4f/$W.json 2> gpurun_out/r4f/$W.err; echo "$W rc=$?"; grep "device memory" gpurun_out/r4f/$W.err | tail -1; python -c "
import json; d=json.load(open('gpurun_out/r4f/$W.json')); c=d['config']; print('$W %.3f Gb/s %.0f ms/step, batches %d, balance %.3f, k3 launch %.1f ms, kernels %s' % (d['value'], d['ms_per_step'], c['batches_per_rank0'], c['extend_wave_balance (mean / max lifetime)'], d['roofline']['avg_launch_ms'], {k: round(v) for k, v in c['kernel_ms_per_step (summed over lanes and ranks)'].items()}))"
done
grep -v "^\[minialign_amd\]   " gpurun_out/r4f/ont.err | grep "run " | tail -5
