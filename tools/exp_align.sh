#!/bin/bash
# (r4) A/B of the phase-aligned workgroup ranges of the scatter pass on the config-5 shard shape, one box.
# usage: tools/exp_align.sh   -> gpurun_out/r4/exp_align.txt (+ TCC counters of both forms)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
out=$O/exp_align.txt; : > $out
SHARD="--rows 1250000 --cols 1000000 --nnz 125000000 --sigma-rows 0.5 --sigma-cols 1.0"
line() { python3 -c "import sys,json; b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k=b['kernels']; print('$1', b['ms_per_step'], {n: v['avg_us'] for n, v in k.items()}, b['layout'])"; }
for al in 0 1; do
  MFX_SCATTER_ALIGN=$al timeout -k 10 300 python3 bench.py $SHARD --k 16 --steps 2 --no-cpu-baseline --no-rank-one 2>$O/err_$al.txt | line "align=$al" >> $out || { echo "align=$al failed" >> $out; tail -5 $O/err_$al.txt >> $out; }
done
cat $out
for al in 0 1; do
  MFX_SCATTER_ALIGN=$al rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_l2_$al -- python3 bench.py $SHARD --k 8 --steps 1 --warmup 1 --no-cpu-baseline --no-event-pass --no-rank-one > $O/pmc_l2_$al.log 2>&1 || echo "pmc $al failed"
  python3 tools/parse_pmc.py $O/pmc_l2_$al 2>/dev/null | grep -i scatter >> $out
done
cat $out
