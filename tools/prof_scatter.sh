# Scatter pass: the solver's kernel vs the microbenchmark of the same shape, under rocprofv3 (separate PMC passes).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/prof_scatter; rm -rf $O; mkdir -p $O
B="python3 bench.py --rows 1250000 --cols 1000000 --nnz 125000000 --k 8 --steps 1 --warmup 1 --no-cpu-baseline --no-rank-one --tiles 16 --graph -1"
U="tools/build/ubench_scatter 120 1250000 6144"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b -- $B --no-event-pass > $O/stats_b.log 2>&1 || echo "stats_b failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_u -- $U > $O/stats_u.log 2>&1 || echo "stats_u failed"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/b_$tag -- $B --no-event-pass > $O/b_$tag.log 2>&1 || echo "b $tag failed"
  rocprofv3 --pmc $set --output-format csv -d $O/u_$tag -- $U > $O/u_$tag.log 2>&1 || echo "u $tag failed"
done
for d in $O/b_* $O/u_*; do [ -d $d ] && { echo "== $d"; python3 tools/parse_pmc.py $d k_scatter; }; done > $O/summary.txt 2>&1
for f in $(find $O/stats_b $O/stats_u -name "*kernel_stats.csv"); do echo "== $f"; grep -i "scatter\|Name" $f | cut -c1-300; done >> $O/summary.txt
cat $O/summary.txt
