mkdir -p gpurun_out/r5
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["value"], d["unit"], d["ms_per_step"], "ms per step", "kernel ms summed", c["kernel_ms_per_step (summed over lanes and ranks)"], "balance", c["extend_wave_balance (mean / max lifetime)"], "reruns", c["reruns_per_step (rank 0)"], "split", c["extend_wave_time_split"])'
R=$PWD
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/r5/c6_clocks.txt 2>&1 & W=$!
echo "== default"; timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "$P"
kill $W
for v in lb6 lb5 lb4; do
  echo "== $v"; MM_LIB_OVERRIDE=$R/minialign_amd/libminialign_amd_$v.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "$P"
done
echo "== lb5, 8 waves per SIMD launched"; MM_K3_WAVES_PER_SIMD=8 MM_LIB_OVERRIDE=$R/minialign_amd/libminialign_amd_lb5.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "$P"
echo "== lb5, 6 lanes"; GPU_MAX_HW_QUEUES=32 MM_LIB_OVERRIDE=$R/minialign_amd/libminialign_amd_lb5.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu --lanes 6 2> /dev/null | python3 -c "$P"
echo "== lb4, 8 waves per SIMD launched"; MM_K3_WAVES_PER_SIMD=8 MM_LIB_OVERRIDE=$R/minialign_amd/libminialign_amd_lb4.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "$P"
echo "== profiling build"; MM_LIB_OVERRIDE=$R/minialign_amd/libminialign_amd_prof.so timeout 600 python bench.py --steps 2 --warmup 1 --no-cli --no-packed --no-cpu 2> /dev/null | python3 -c "$P"
echo "== lb5 verbose lane trace"; MM_VERBOSE=1 MM_LIB_OVERRIDE=$R/minialign_amd/libminialign_amd_lb5.so timeout 600 python bench.py --steps 2 --warmup 1 --no-cli --no-packed --no-cpu 2> gpurun_out/r5/c6_v.err | python3 -c "$P"; python3 tools/lane_trace.py gpurun_out/r5/c6_v.err > gpurun_out/r5/c6_lb5_lane_trace.txt; python3 tools/k3_overlap.py gpurun_out/r5/c6_v.err > gpurun_out/r5/c6_lb5_overlap.txt 2>&1; rm -f gpurun_out/r5/c6_v.err; tail -3 gpurun_out/r5/c6_lb5_overlap.txt
sort gpurun_out/r5/c6_clocks.txt | uniq -c | sort -rn | head -8
