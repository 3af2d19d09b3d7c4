#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
out=$O/exp_mid2.txt; : > $out
line() { python3 -c "import sys,json; b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k=b['kernels']; print('$1', 'ms_per_step', b['ms_per_step'], {n: v['avg_us'] for n, v in k.items() if 'pass' in n or 'final' in n}, b['layout']['csc']['panels'], b['layout']['csr']['panels'], b['layout']['csr']['kind'])"; }
for fl in "--tiles 2" "--tiles 4" "--tiles 8" "--wg-waves 8 --tiles 4" "--wg-waves 8 --tiles 8" "--wg-waves 4 --tiles 8" "--panel-rows -1" "--panel-rows 3000" "--panel-rows 11000"; do
  python3 bench.py --rows 69878 --cols 10677 --nnz 10000054 --k 40 --steps 10 --warmup 2 --no-cpu-baseline --no-rank-one --no-als $fl 2>$O/err_mid.txt | line "ml10m $fl" >> $out || tail -2 $O/err_mid.txt >> $out
done
cat $out
