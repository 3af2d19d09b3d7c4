cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 800 python3 -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/pytest_gpu6.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/pytest_gpu6.log
run() { name=$1; shift; timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/sw_$name.log 2>&1; python3 - <<PY
import json
try:
    l=[x for x in open('gpurun_out/sw_$name.log') if x.startswith('{')][-1]; j=json.loads(l)
    print('$name', j['ms_per_step'], 'ms', {k:v['avg_us'] for k,v in j['kernels'].items()}, 'rmse', j['test_rmse_after'])
except Exception as e: print('$name FAILED', e)
PY
}
run auto
MFX_DBG=1 run dbg1
MFX_DBG=4 run dbg4
MFX_DBG=2 run dbg2
run t16 --tiles 16
