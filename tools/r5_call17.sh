R=$PWD
echo "== traced, 20 kb"; MM_LIB_OVERRIDE=$R/minialign_amd/libminialign_amd_tprof.so python tools/qb_gaba.py 16384 20000 1 2>&1 | tail -12
echo "== untraced, 20 kb (production lib)"; python tools/qb_gaba.py 16384 20000 0 2>&1 | tail -3
echo "== traced, 20 kb (production lib)"; python tools/qb_gaba.py 16384 20000 1 2>&1 | tail -3
