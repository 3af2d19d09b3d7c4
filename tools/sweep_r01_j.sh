cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/sw_$name.log 2>&1; python3 - <<PY
import json
try:
    l=[x for x in open('gpurun_out/sw_$name.log') if x.startswith('{')][-1]; j=json.loads(l)
    print('$name', j['ms_per_step'], 'ms', {k:v['avg_us'] for k,v in j['kernels'].items()})
except Exception as e: print('$name FAILED', e)
PY
}
for kb in 56 72 64 48 40; do MFX_SLICE_KB=$kb run slice$kb; done
