set -e
OUT=gpurun_out/strict/pmc
mkdir -p $OUT
export TMPDIR=/tmp FLAGS=0 NSWEEP=6 NSTEP=0
CFD_VARIANT=8 timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/p1 -- python3 scripts/prof_target.py > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/p2 -- python3 scripts/prof_target.py > $OUT/p2.log 2>&1
CFD_VARIANT=8 timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p4 -- python3 scripts/prof_target.py > $OUT/p4.log 2>&1
python3 scripts/pmc_summary.py strict/pmc
