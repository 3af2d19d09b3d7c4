export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests -m gpu -q -x -k "rccl or ml1m" > $O/pytest_gpu7.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu7.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_stats2.log 2>&1 || echo "stats failed"
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_stats2/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'mfx' in r['Name']: print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
