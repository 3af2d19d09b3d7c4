#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
out=$O/exp_mid4.txt; : > $out
line() { python3 -c "import sys,json; b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k=b['kernels']; print('$1', 'ms_per_step', b['ms_per_step'], {n: v['avg_us'] for n, v in k.items() if 'pass' in n}, b['layout']['csc']['panels'], b['layout']['csr']['panels'], b['layout']['csc']['tiles_per_span'], b['layout']['csr']['tiles_per_span'])"; }
for shape in "40000 8000 5000000" "55000 9000 7000000" "100000 15000 15000000" "160000 30000 25000000" "70000 10000 4200000"; do
  set -- $shape
  for t in 0 2 4 6 8; do
    python3 bench.py --rows $1 --cols $2 --nnz $3 --k 32 --steps 8 --warmup 2 --no-cpu-baseline --no-rank-one --no-als --tiles $t 2>$O/err_mid.txt | line "$1x$2 nnz=$3 tiles=$t" >> $out || tail -2 $O/err_mid.txt >> $out
  done
done
cat $out
