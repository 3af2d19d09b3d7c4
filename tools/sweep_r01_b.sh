cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/sw_$name.log 2>&1; python3 - <<PY
import json
try:
    l=[x for x in open('gpurun_out/sw_$name.log') if x.startswith('{')][-1]; j=json.loads(l)
    print('$name', j['ms_per_step'], 'ms', '%.3g nnz/s'%j['value'], {k:v['avg_us'] for k,v in j['kernels'].items()}, 'frac', j['roofline']['frac'], 'rmse', j['test_rmse_after'])
except Exception as e: print('$name FAILED', e)
PY
}
run auto
run off --panel-rows -1
run w4 --wg-waves 4
run w16 --wg-waves 16
run t8 --tiles 8
run t16 --tiles 16
run pr16k_w16 --panel-rows 16383 --wg-waves 16
run pr4k --panel-rows 4095
run pr2k_w4 --panel-rows 2047 --wg-waves 4
