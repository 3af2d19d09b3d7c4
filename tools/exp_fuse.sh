#!/bin/bash
# experiments around the fused finalize (one box): E1 = fused kernel, nobody completes, separate finalize after;
# E2 = the same with plain partial stores
out=gpurun_out/exp_fuse.txt; : > $out
run() { echo "== $1" >> $out; shift; env "$@" python bench.py --steps 3 --no-cpu-baseline --no-rank-one 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'], {n: v['avg_us'] for n, v in b['kernels'].items()}, b['test_rmse_after'])" >> $out; }
run "unfused" MFX_FUSE_FINALIZE=0
run "fused sorted" MFX_FUSE_FINALIZE=1
run "fused identity order" MFX_FUSE_FINALIZE=2
run "E1: fused kernel (identity order), no completion, separate finalize, sc1 stores" MFX_FUSE_FINALIZE=3
run "E2: same, plain stores" MFX_FUSE_FINALIZE=3 MFX_LIB_PATH=$PWD/build_ubench/libmfx_plain.so
cat $out
