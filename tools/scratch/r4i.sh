mkdir -p gpurun_out/r4i
for H in 8 16 32; do
MM_K3_HELPERS=$H MM_VERBOSE=1 timeout 300 python bench.py --workload ont --steps 3 --warmup 1 --no-cli --no-packed --no-cpu > gpurun_out/r4i/ont$H.json 2> gpurun_out/r4i/ont$H.err; echo "helpers 1 in $H rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r4i/ont$H.json')); c=d['config']; print('helpers 1 in $H: %.3f Gb/s %.0f ms/step, batches %d, balance %.3f, k3 launch %.1f ms' % (d['value'], d['ms_per_step'], c['batches_per_rank0'], c['extend_wave_balance (mean / max lifetime)'], d['roofline']['avg_launch_ms']))"
grep -v "^\[minialign_amd\]   " gpurun_out/r4i/ont$H.err | grep "run " | tail -5 | cut -c1-100
done
