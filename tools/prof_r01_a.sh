set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_stats.log 2>&1 || echo "stats failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_fetch.log 2>&1 || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_write.log 2>&1 || echo "write failed"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/prof_l2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_l2.log 2>&1 || echo "l2 failed"
for t in 2 4 8 16; do python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --tiles $t > $O/bench_tiles$t.log 2>&1; done
ls -R $O | head -50
