# SQ counters of the ALS half-sweeps (separate --pmc passes, kernel filter by name in parse_pmc.py)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out/als_pmc; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $O/a -- python3 bench.py --solver als --steps 1 --warmup 1 > $O/a.log 2>&1 || echo "a failed"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/b -- python3 bench.py --solver als --steps 1 --warmup 1 > $O/b.log 2>&1 || echo "b failed"
python3 tools/parse_pmc.py $O/a k_als; python3 tools/parse_pmc.py $O/b k_als
