#!/bin/bash
# ALS iteration time for a list of k, one line each.  usage: tools/ab_als.sh "128 96 64" -> gpurun_out/ab_als.txt
out=gpurun_out/ab_als.txt
mkdir -p gpurun_out; : > $out
for k in $1; do
  python bench.py --solver als --k $k --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k=$k', b['value'], {n: v['ms'] for n, v in b['half_sweeps'].items()}, b['rmse'])" >> $out || exit 1
done
cat $out
