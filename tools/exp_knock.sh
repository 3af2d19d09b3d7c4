#!/bin/bash
# (r4 experiment) knock-out builds of k_scatter: which pipeline bounds the pass?  usage: tools/exp_knock.sh "libs..." [env]
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
out=$O/exp_knock.txt
SHARD="--rows 1250000 --cols 1000000 --nnz 125000000 --sigma-rows 0.5 --sigma-cols 1.0"
for lib in $1; do
  MFX_LIB_PATH=$PWD/cuda-recommender_amd/$lib timeout -k 10 300 python3 bench.py $SHARD --k 16 --steps 2 --no-cpu-baseline --no-rank-one 2>$O/err_knock.txt \
   | python3 -c "import sys,json; b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k=b['kernels']; print('$lib', '$MFX_SCATTER_ALIGN', b['ms_per_step'], {n: v['avg_us'] for n, v in k.items() if 'scatter' in n})" >> $out || { echo "$lib failed" >> $out; tail -3 $O/err_knock.txt >> $out; }
done
cat $out
