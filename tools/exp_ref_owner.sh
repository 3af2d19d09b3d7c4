#!/bin/bash
# (r4) reference-order owner passes at the Netflix shape: per-kernel durations by long-segment threshold (MFX_REF_LONG), by cap of
# k_ref_quad workgroups (MFX_REF_QUAD_WGS) and for the as-written sequence (MFX_REF_FUSED=0).  Run on the GPU box from the repo root;
# writes gpurun_out/r4b/exp_ref_owner.txt.
mkdir -p gpurun_out/r4b
OUT=$PWD/gpurun_out/r4b/exp_ref_owner.txt
REPO=$PWD
: > $OUT
cd /tmp && export TMPDIR=/tmp
one() {  # label, env...
  label=$1; shift
  rm -rf $REPO/gpurun_out/r4b/prof_e
  ( export "$@"; timeout -k 10 300 rocprofv3 --kernel-trace -d $REPO/gpurun_out/r4b/prof_e -o ref -- python3 $REPO/bench.py --schedule 0 --variant -1 --steps 1 --warmup 1 --no-cpu-baseline --no-als --no-rank-one > $REPO/gpurun_out/r4b/prof_e.log 2>&1 )
  echo "== $label" >> $OUT
  python3 $REPO/tools/prof_db.py $REPO/gpurun_out/r4b/prof_e k_ | cut -c28-52,105-200 >> $OUT
  grep -o '"ms_per_step": [0-9.]*' $REPO/gpurun_out/r4b/prof_e.log | head -1 >> $OUT
}
one "default (owner passes, split from 2048 entries where a side has a segment of >= 8192)" MFX_REF_FUSED=1
for L in 1024 2048 8192 32768 150000; do one "threshold $L" MFX_REF_LONG=$L; done
for W in 512 2048; do one "k_ref_quad capped at $W workgroups" MFX_REF_QUAD_WGS=$W; done
one "as-written sequence (MFX_REF_FUSED=0)" MFX_REF_FUSED=0
cat $OUT
