#!/bin/bash
# (r4) launch-bound shapes (BASELINE configs[0] / [1]) under hipGraph replay: A/B of an environment knob
# usage: tools/exp_small.sh VAR "v1 v2"
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
out=$O/exp_small_$1.txt; : > $out
ML1M="--rows 6040 --cols 3706 --nnz 1000000 --k 40"
ML100K="--rows 943 --cols 1682 --nnz 100000 --k 10"
line() { python3 -c "import sys,json; b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k=b['kernels']; print('$1', 'ms_per_step', b['ms_per_step'], {n: (v['avg_us'], v['launches']) for n, v in k.items()}, b['layout']['csc']['kind'], b['test_rmse_after'])"; }
for v in $2; do
  env $1=$v python3 bench.py $ML1M --steps 100 --warmup 5 --no-cpu-baseline --no-rank-one --no-als 2>$O/err_small.txt | line "ml1m $1=$v" >> $out || tail -3 $O/err_small.txt >> $out
  env $1=$v python3 bench.py $ML100K --steps 200 --warmup 5 --no-cpu-baseline --no-rank-one --no-als 2>$O/err_small.txt | line "ml100k $1=$v" >> $out || tail -3 $O/err_small.txt >> $out
done
cat $out
