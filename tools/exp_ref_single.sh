#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4; mkdir -p $O
for cfg in "1 0" "1 1" "64 0" "64 1"; do
  set -- $cfg
  rm -rf $O/trace_rs
  MFX_REF_SWEEP_DPP=$2 rocprofv3 --kernel-trace --output-format csv -d $O/trace_rs -- python3 tools/exp_ref_single.py $1 > $O/trace_rs.log 2>&1 || tail -3 $O/trace_rs.log
  grep "bit-identical" $O/trace_rs.log
  python3 - "$1 columns dpp=$2" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/r4/trace_rs/**/*kernel_trace.csv", recursive=True)
rows = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f[0])) if "k_sweep_ref" in r["Kernel_Name"]]
print(sys.argv[1], "sweep kernel us:", [round(d / 1e3) for d in rows], "-> clocks per entry of a 240000-entry chain at 2.4 GHz:", round(min(rows) * 2.4 / 240000, 2))
PY
done
