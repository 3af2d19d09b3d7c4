#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
out=$O/exp_als_libs.txt
for lib in $1; do
  MFX_LIB_PATH=$PWD/cuda-recommender_amd/$lib python3 bench.py --solver als --steps 4 --warmup 1 2>$O/err_als.txt | python3 -c "import sys,json; b=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$lib', b['value'], {n: h['ms'] for n, h in b['half_sweeps'].items()}, b['rmse'][-1])" >> $out || tail -2 $O/err_als.txt >> $out
done
tail -$(echo $1 | wc -w) $out
