#!/bin/bash
# (r4) where do the segment-owner passes stop paying?  shapes between 1e5 and 4e6 ratings, owner on / off, graph replay
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
out=$O/exp_owner_range.txt; : > $out
line() { python3 -c "import sys,json; b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k=b['kernels']; print('$1', 'ms_per_step', b['ms_per_step'], {n: v['avg_us'] for n, v in k.items() if 'pass' in n or 'final' in n}, b['layout']['csc']['kind'])"; }
for shape in "6040 3706 1000000 40" "20000 8000 2000000 32" "30000 10000 3900000 32" "70000 2000 3900000 32" "2000 70000 3900000 32" "200000 100000 3900000 16"; do
  set -- $shape
  for ow in 1 0; do
    for sc in 1.8 0.5; do
      MFX_OWNER_PASSES=$ow python3 bench.py --rows $1 --cols $2 --nnz $3 --k $4 --sigma-cols $sc --steps 30 --warmup 3 --no-cpu-baseline --no-rank-one --no-als 2>$O/err_or.txt | line "$1x$2 nnz=$3 k=$4 sigma_cols=$sc owner=$ow" >> $out || tail -2 $O/err_or.txt >> $out
    done
  done
done
cat $out
