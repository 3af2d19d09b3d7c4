#!/bin/bash
# (r4) small shapes under graph replay, A/B of library builds.  usage: tools/exp_small_libs.sh "libmfx.so libmfx_x.so" [ENV=V ...]
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
out=$O/exp_small_libs.txt
ML1M="--rows 6040 --cols 3706 --nnz 1000000 --k 40"
ML100K="--rows 943 --cols 1682 --nnz 100000 --k 10"
line() { python3 -c "import sys,json; b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k=b['kernels']; print('$1', 'ms_per_step', b['ms_per_step'], {n: (v['avg_us'], v['launches']) for n, v in k.items() if 'pass' in n}, b['test_rmse_after'])"; }
libs=$1; shift
for lib in $libs; do
  env "$@" MFX_LIB_PATH=$PWD/cuda-recommender_amd/$lib python3 bench.py $ML1M --steps 100 --warmup 5 --no-cpu-baseline --no-rank-one --no-als 2>$O/err_small.txt | line "ml1m $lib $*" >> $out || tail -3 $O/err_small.txt >> $out
  env "$@" MFX_LIB_PATH=$PWD/cuda-recommender_amd/$lib python3 bench.py $ML100K --steps 200 --warmup 5 --no-cpu-baseline --no-rank-one --no-als 2>$O/err_small.txt | line "ml100k $lib $*" >> $out || tail -3 $O/err_small.txt >> $out
done
tail -$(( $(echo $libs | wc -w) * 2 )) $out
