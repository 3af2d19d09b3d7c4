export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/prof_sq1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_sq1.log 2>&1 || echo sq1 failed
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/prof_sq2 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_sq2.log 2>&1 || echo sq2 failed
MFX_DBG=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $O/prof_sq1_dbg1 -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/prof_sq1d.log 2>&1 || echo sq1d failed
ls $O/prof_sq1/*/ | head
