# Round-2 profile set: bench line, rocprofv3 kernel stats, HBM traffic PMC passes (separate runs), ALS, the
# hyper-sparse shard shape (scatter layout).  Run on the GPU box: bash tools/prof_r02_final.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/final; rm -rf $O; mkdir -p $O
python3 bench.py --steps 5 --warmup 1 > $O/bench.log 2>&1; echo "bench exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-event-pass > $O/stats.log 2>&1 || echo "stats failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1 || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1 || echo "write failed"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/l2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/l2.log 2>&1 || echo "l2 failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/sq.log 2>&1 || echo "sq failed"
python3 bench.py --solver als --steps 3 --warmup 1 > $O/bench_als.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_als -- python3 bench.py --solver als --steps 3 --warmup 1 > $O/stats_als.log 2>&1 || echo "als stats failed"
python3 bench.py --solver als --k 128 --steps 3 --warmup 1 > $O/bench_als128.log 2>&1
# config 5's per-GPU shard (1.25 M x 1 M, 125 M ratings, k = 128): scatter layout
S="--rows 1250000 --cols 1000000 --nnz 125000000 --k 128 --sigma-rows 0.5 --sigma-cols 1.0"
python3 bench.py $S --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_shard.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_shard -- python3 bench.py $S --k 16 --steps 2 --warmup 1 --no-cpu-baseline --no-event-pass --no-rank-one > $O/stats_shard.log 2>&1 || echo "shard stats failed"
grep '^{' $O/bench.log | tail -1 | cut -c1-600
