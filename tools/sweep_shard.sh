#!/bin/bash
# A/B of scatter-layout parameters on the config-5 shard shape, all in one box (run-to-run noise between boxes is ~3 %).
# usage: tools/sweep_shard.sh "<panel_rows>:<tiles> ..."   -> gpurun_out/sweep_shard.txt
out=gpurun_out/sweep_shard.txt
: > $out
for pt in $1; do
  pr=${pt%%:*}; tl=${pt##*:}
  python bench.py --workload config5 --rows 1250000 --nnz 125000000 --steps 2 --no-cpu-baseline --no-rank-one --variant 2 --panel-rows $pr --tiles $tl 2>/dev/null \
    | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=b['kernels']; print('$pr', '$tl', b['ms_per_step'], {n: v['avg_us'] for n, v in k.items()})" >> $out || exit 1
done
cat $out
