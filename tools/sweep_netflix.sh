#!/bin/bash
# tiles_per_span A/B on the Netflix shape in one box.  usage: tools/sweep_netflix.sh "6 8 10 ..."  -> gpurun_out/sweep_netflix.txt
out=gpurun_out/sweep_netflix.txt
: > $out
for tl in $1; do
  python bench.py --steps 3 --no-cpu-baseline --no-rank-one --tiles $tl 2>/dev/null \
    | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=b['kernels']; print('$tl', b['ms_per_step'], {n: v['avg_us'] for n, v in k.items()}, b['layout']['csc']['tiles_per_span'], b['layout']['csr']['tiles_per_span'])" >> $out || exit 1
done
cat $out
