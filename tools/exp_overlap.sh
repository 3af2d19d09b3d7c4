#!/bin/bash
# (r4) cost and timeline of the panel-group overlap on ONE GPU (1-rank RCCL communicator, config-5 shard shape)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
out=$O/exp_overlap.txt; : > $out; CFGS=${CFGS:-"1:0 2:0 2:16 3:16 4:16"}
SHARD="--rows 1250000 --cols 1000000 --nnz 125000000 --sigma-rows 0.5 --sigma-cols 1.0"
line() { python3 -c "import sys,json; b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); k=b['kernels']; print('$1', b['ms_per_step'], {n: (v['avg_us'], v['launches']) for n, v in k.items()}, b['layout']['csr'], b['test_rmse_after'])"; }
for cfg in $CFGS; do
  set -- ${cfg/:/ }
  MFX_OVERLAP_GROUPS=$1 MFX_COMM_RESERVE_CUS=$2 timeout -k 10 300 python3 bench.py $SHARD --k 16 --steps 2 --force-comm --no-cpu-baseline --no-rank-one 2>$O/err_ov.txt | line "groups=$1 reserve=$2" >> $out || { echo "groups=$1 failed" >> $out; tail -3 $O/err_ov.txt >> $out; }
done
cat $out
rm -rf $O/trace_ov
MFX_OVERLAP_GROUPS=4 MFX_COMM_RESERVE_CUS=16 rocprofv3 --kernel-trace --output-format csv -d $O/trace_ov -- python3 bench.py $SHARD --k 8 --steps 1 --warmup 1 --force-comm --no-cpu-baseline --no-event-pass --no-rank-one > $O/trace_ov.log 2>&1 || echo "trace failed"
python3 tools/overlap_from_trace.py $O/trace_ov | tee -a $out
find $O/trace_ov -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'head -1 {}; grep -m 40 "k_scatter\|k_finalize\|ccl\|oneRank" {} | tail -24' > $O/trace_ov_head.txt
find $O/trace_ov -name "*.csv" ! -name "*kernel_trace.csv" -delete
