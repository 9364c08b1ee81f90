mkdir -p gpurun_out/r5/cost; R=$PWD
MM_LIB_OVERRIDE=$R/minialign_amd/libminialign_amd_prof.so MM_DUMP_READ_COST=$R/gpurun_out/r5/cost/hard MM_VERBOSE=1 timeout 900 python bench.py --workload hg38hard --depth 0.3 --steps 1 --warmup 0 --no-cli --no-packed --no-cpu 2> gpurun_out/r5/c21.err > /dev/null
grep "run \|carry" gpurun_out/r5/c21.err | head -14
for f in gpurun_out/r5/cost/hard gpurun_out/r5/cost/hard.2 gpurun_out/r5/cost/hard.4; do echo "##### $f"; python3 tools/read_cost.py $f 5120 | tail -34; done > gpurun_out/r5/c21_hard_read_cost.txt
rm -rf gpurun_out/r5/cost gpurun_out/r5/c21.err
