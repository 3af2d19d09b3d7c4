cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/sw_$name.log 2>&1; python3 - <<PY
import json
try:
    l=[x for x in open('gpurun_out/sw_$name.log') if x.startswith('{')][-1]; j=json.loads(l)
    print('$name', j['ms_per_step'], 'ms', {k:v['avg_us'] for k,v in j['kernels'].items()})
except Exception as e: print('$name FAILED', e)
PY
}
run auto
run w8_pr4600 --wg-waves 8 --panel-rows 4600
run w8_pr3500 --wg-waves 8 --panel-rows 3500
run w8_pr2500 --wg-waves 8 --panel-rows 2500
run w4_pr1700 --wg-waves 4 --panel-rows 1700
run w16_t4 --tiles 4
run w16_t6 --tiles 6
run w16_t12 --tiles 12
run w8_pr3500_t16 --wg-waves 8 --panel-rows 3500 --tiles 16
