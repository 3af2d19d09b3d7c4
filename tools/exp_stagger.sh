#!/bin/bash
# A/B of the wave stagger of the scatter pass (MFX_SCATTER_STAGGER) on the config-5 shard shape.  -> gpurun_out/exp_stagger.txt
out=gpurun_out/exp_stagger.txt
: > $out
for sg in $1; do
  MFX_SCATTER_STAGGER=$sg python bench.py --workload config5 --rows 1250000 --nnz 125000000 --k 32 --steps 2 --no-cpu-baseline --no-rank-one --tiles ${2:-4} 2>/dev/null \
    | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=b['kernels']; print('stagger $sg', b['ms_per_step'], {n: v['avg_us'] for n, v in k.items()})" >> $out || exit 1
done
cat $out
