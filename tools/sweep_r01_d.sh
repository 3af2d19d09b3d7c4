cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 800 python3 -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/pytest_gpu3.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_gpu3.log
run() { name=$1; shift; timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/sw_$name.log 2>&1; python3 - <<PY
import json
try:
    l=[x for x in open('gpurun_out/sw_$name.log') if x.startswith('{')][-1]; j=json.loads(l)
    print('$name', j['ms_per_step'], 'ms', {k:v['avg_us'] for k,v in j['kernels'].items()}, 'rmse', j['test_rmse_after'])
except Exception as e: print('$name FAILED', e)
PY
}
run auto
for d in 1 2 4 6; do MFX_DBG=$d run dbg$d; done
run w16 --wg-waves 16
run w4 --wg-waves 4
run t8 --tiles 8
run pr3500 --panel-rows 3500
run pr14k_w16 --panel-rows 14000 --wg-waves 16
