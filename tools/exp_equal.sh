#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
out=$O/exp_equal.txt; : > $out
line() { python3 -c "import sys,json; b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k=b['kernels']; print('$1', 'ms_per_step', b['ms_per_step'], {n: v['avg_us'] for n, v in k.items() if 'pass' in n or 'final' in n}, b['layout']['csc']['panels'], b['layout']['csc']['panel_rows'], b['layout']['csr']['panels'], b['layout']['csr']['panel_rows'])"; }
for shape in "69878 10677 10000054 40" "138493 26744 20000263 40" "480189 17770 99072112 64" "300000 40000 30000000 32"; do
  set -- $shape
  for eq in 0 1; do
    if [ $eq = 1 ]; then export MFX_EQUAL_PANELS=1; else unset MFX_EQUAL_PANELS; fi
    python3 bench.py --rows $1 --cols $2 --nnz $3 --k $4 --steps 6 --warmup 2 --no-cpu-baseline --no-rank-one --no-als 2>$O/err_mid.txt | line "$1x$2 nnz=$3 k=$4 equal=$eq" >> $out || tail -2 $O/err_mid.txt >> $out
  done
done
cat $out
