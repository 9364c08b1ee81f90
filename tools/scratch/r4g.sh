mkdir -p gpurun_out/r4g
MM_LIB_OVERRIDE=$PWD/minialign_amd/libminialign_amd_prof.so MM_DUMP_READ_COST=gpurun_out/r4g/cost.tsv MM_VERBOSE=1 MM_VERBOSE_SLABS=1 timeout 300 python bench.py --workload ont --steps 1 --warmup 0 --no-cli --no-packed --no-cpu > gpurun_out/r4g/ont.json 2> gpurun_out/r4g/ont.err; echo rc=$?
grep "run \|workspace class" gpurun_out/r4g/ont.err
for f in gpurun_out/r4g/cost.tsv*; do echo "== $f"; python tools/read_cost.py $f | tail -30; done
rm -f gpurun_out/r4g/cost.tsv*
