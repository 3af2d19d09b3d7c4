#!/bin/bash
# (r4) per-launch durations of the reference-order sweeps (v-sweep, u-sweep alternate) at the Netflix shape; usage: tools/exp_ref.sh "thr..."
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
for thr in $1; do
  rm -rf $O/trace_ref
  MFX_REF_LONG=$thr MFX_REF_SWEEP_DPP=${DPP:-0} rocprofv3 --kernel-trace --output-format csv -d $O/trace_ref -- python3 bench.py --schedule 0 --variant -1 --steps 1 --warmup 1 --k 4 --no-cpu-baseline --no-rank-one --no-als --no-event-pass > $O/trace_ref.log 2>&1 || tail -3 $O/trace_ref.log
  python3 - "$thr" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/r4/trace_ref/**/*kernel_trace.csv", recursive=True)
if not f:
    print("no trace"); sys.exit(0)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f[0])) if "k_sweep_ref" in r["Kernel_Name"]]
rows.sort()
print("MFX_REF_LONG=" + sys.argv[1], "sweep launches (us, v and u alternate):", [round(d / 1e3) for _, d in rows[-6:]])
PY
done
