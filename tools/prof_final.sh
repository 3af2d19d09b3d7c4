# Profile set (round 4 on).  Run on the GPU box: bash tools/prof_final.sh [tags...]   (default: all)
# One directory per workload under gpurun_out/final/<tag>/: bench.log (the bench line), stats/ (rocprofv3 --kernel-trace
# --stats of the same command without the event pass), fetch/ write/ (+ l2/ sq/ where asked): separate --pmc passes.
# tools/collect_profiles.py r04 then copies the summaries into profiles/ and rebuilds profiles/traffic.json.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; F=gpurun_out/final; mkdir -p $F
want() { [ -z "$TAGS" ] || echo " $TAGS " | grep -q " $1 "; }
TAGS="$*"
run_set() {  # tag, bench flags for the line, bench flags for the profiled runs, pmc sets...
  local tag=$1 line="$2" prof="$3"; shift 3
  local O=$F/$tag; rm -rf $O; mkdir -p $O
  echo "== $tag"
  timeout -k 10 900 python3 bench.py $line > $O/bench.log 2> $O/bench.err || { echo "$tag: bench failed"; tail -3 $O/bench.err; return 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py $prof --no-cpu-baseline --no-event-pass --no-rank-one > $O/stats.log 2>&1 || echo "$tag: stats failed"
  for set in "$@"; do
    local name=$(echo $set | cut -d' ' -f1 | tr 'A-Z' 'a-z')
    rocprofv3 --pmc $set --output-format csv -d $O/pmc_$name -- python3 bench.py $prof --steps 1 --warmup 1 --no-cpu-baseline --no-event-pass --no-rank-one > $O/pmc_$name.log 2>&1 || echo "$tag: pmc $name failed"
  done
  grep '^{' $O/bench.log | tail -1 | cut -c1-400
}
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
L2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
TCP="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
LDSC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
SHARD="--rows 1250000 --cols 1000000 --nnz 125000000 --sigma-rows 0.5 --sigma-cols 1.0"
want netflix   && run_set netflix   "--steps 5 --warmup 1" "--steps 3 --warmup 1" "FETCH_SIZE" "WRITE_SIZE" "$L2" "$SQ"
want nnz1e9    && run_set nnz1e9    "--workload nnz1e9 --steps 3 --warmup 1" "--workload nnz1e9 --steps 1 --warmup 1" "FETCH_SIZE" "WRITE_SIZE"
want shard     && run_set shard     "$SHARD --k 128 --steps 2 --warmup 1 --no-cpu-baseline" "$SHARD --k 8 --steps 2 --warmup 1" "FETCH_SIZE" "WRITE_SIZE" "$L2" "$SQ" "$TCP" "$LDSC"
want config5   && run_set config5   "--workload config5 --gpus 1 --steps 2 --warmup 1" "--workload config5 --gpus 1 --k 8 --steps 1 --warmup 1" "FETCH_SIZE" "WRITE_SIZE"
want config5c  && { O=$F/config5c; rm -rf $O; mkdir -p $O; echo "== config5c"; timeout -k 10 900 python3 bench.py --workload config5 --gpus 1 --steps 2 --warmup 1 --force-comm --no-cpu-baseline --no-rank-one > $O/bench.log 2> $O/bench.err || echo "config5c failed"; }
want ml1m      && run_set ml1m      "--rows 6040 --cols 3706 --nnz 1000000 --k 40 --steps 100 --warmup 5 --no-als" "--rows 6040 --cols 3706 --nnz 1000000 --k 40 --steps 20 --warmup 5 --no-als"
want ml100k    && run_set ml100k    "--rows 943 --cols 1682 --nnz 100000 --k 10 --steps 200 --warmup 5 --no-als" "--rows 943 --cols 1682 --nnz 100000 --k 10 --steps 50 --warmup 5 --no-als"
# the panel-group overlap of the sharded column pass on ONE GPU (1-rank RCCL communicator): bench line, kernel stats, and where the
# exchange kernels ran relative to the column passes (tools/overlap_from_trace.py on the same kernel trace)
want shardov   && { export MFX_OVERLAP_GROUPS=2 MFX_COMM_RESERVE_CUS=16; run_set shardov "$SHARD --k 128 --steps 2 --warmup 1 --force-comm --no-cpu-baseline --no-rank-one" "$SHARD --k 8 --steps 2 --warmup 1 --force-comm"; unset MFX_OVERLAP_GROUPS MFX_COMM_RESERVE_CUS; python3 tools/overlap_from_trace.py $F/shardov/stats > $F/shardov/extra_overlap.txt 2>&1 < /dev/null; cat $F/shardov/extra_overlap.txt; }
# the reference-order parity mode (kernel_variant -1): owner passes in the reference's summation order (ccd_reforder.hip)
REF="--schedule 0 --variant -1 --no-als --no-rank-one"
want reforder  && run_set reforder  "$REF --steps 3 --warmup 1 --no-cpu-baseline" "$REF --steps 1 --warmup 1" "FETCH_SIZE" "WRITE_SIZE" "$SQ"
want als       && run_set als       "--solver als --steps 3 --warmup 1" "--solver als --steps 2 --warmup 1" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES"
want als128    && run_set als128    "--solver als --k 128 --steps 3 --warmup 1" "--solver als --k 128 --steps 2 --warmup 1"
echo "prof_final done"
