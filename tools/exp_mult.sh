#!/bin/bash
# (r4) panel-count multiples of the scatter layout on the config-5 shard shape.  usage: tools/exp_mult.sh "0 32 64 256"
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
out=$O/exp_mult.txt
SHARD="--rows 1250000 --cols 1000000 --nnz 125000000 --sigma-rows 0.5 --sigma-cols 1.0"
for m in $1; do
  MFX_SCATTER_PANEL_MULT=$m timeout -k 10 300 python3 bench.py $SHARD --k 16 --steps 2 --no-cpu-baseline --no-rank-one $2 2>$O/err_mult.txt \
   | python3 -c "import sys,json; b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k=b['kernels']; print('mult=$m', b['ms_per_step'], {n: v['avg_us'] for n, v in k.items() if 'scatter' in n}, b['layout']['csc']['panels'], b['layout']['csr']['panels'], b['test_rmse_after'])" >> $out || { echo "mult=$m failed" >> $out; tail -3 $O/err_mult.txt >> $out; }
done
cat $out
