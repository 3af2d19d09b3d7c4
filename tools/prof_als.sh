export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/als; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/sq -- python3 bench.py --solver als --steps 1 --warmup 1 > $O/sq.log 2>&1 || echo sq failed
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/sq2 -- python3 bench.py --solver als --steps 1 --warmup 1 > $O/sq2.log 2>&1 || echo sq2 failed
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --solver als --steps 2 --warmup 1 > $O/stats.log 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/als/stats/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'mfx' in r['Name']: print(r['Name'][:80], r['Calls'], r['AverageNs'])
PY
