# Scatter pass on config 5's per-GPU shard (1.25 M x 1 M, 125 M ratings): kernel trace + the counters it lacked in
# round 2 (HBM traffic, L2 hit/miss, wait / LDS-conflict cycles), separate rocprofv3 --pmc passes, k = 8 to keep the
# serialised counter runs short.  usage: bash tools/prof_scatter.sh [out-dir] [extra bench flags]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=${1:-gpurun_out/prof_scatter}; shift; rm -rf $O; mkdir -p $O
B="python3 bench.py --rows 1250000 --cols 1000000 --nnz 125000000 --k 8 --sigma-rows 0.5 --sigma-cols 1.0 --steps 1 --warmup 1 --no-cpu-baseline --no-rank-one --graph -1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --no-event-pass > $O/stats.log 2>&1 || echo "stats failed"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/p_$tag -- $B --no-event-pass > $O/p_$tag.log 2>&1 || echo "pmc $tag failed"
done
for d in $O/p_*; do [ -d $d ] && { echo "== $d"; python3 tools/parse_pmc.py $d k_scatter; }; done > $O/summary.txt 2>&1
for f in $(find $O/stats -name "*kernel_stats.csv"); do echo "== $f"; grep -i "scatter\|Name\|finalize" $f | cut -c1-300; done >> $O/summary.txt
cat $O/summary.txt
