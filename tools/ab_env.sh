#!/bin/bash
# A/B of an environment knob on a bench workload.  usage: tools/ab_env.sh VAR "v1 v2 ..." [bench flags...]  -> gpurun_out/ab_env.txt
var=$1; vals=$2; shift 2
out=gpurun_out/ab_env.txt; mkdir -p gpurun_out; : > $out
for v in $vals; do
  env $var=$v python bench.py --no-cpu-baseline --no-rank-one "$@" 2>/dev/null \
    | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=b['kernels']; print('$var=$v', b['ms_per_step'], {n: x['avg_us'] for n, x in k.items()}, b['test_rmse_after'])" >> $out || exit 1
done
cat $out
