export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/final; mkdir -p $O; rm -rf $O/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-event-pass > $O/stats.log 2>&1 || echo "stats failed"
python3 bench.py --steps 3 --warmup 1 > $O/bench.log 2>&1
grep '^{' $O/stats.log | cut -c1-200
