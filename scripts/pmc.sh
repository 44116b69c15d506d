# usage: bash scripts/pmc.sh <tag>   -- separate counter passes (never combined with tracing), then a kernel trace
set -e
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/p1 -- python3 scripts/prof_target.py > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p2 -- python3 scripts/prof_target.py > $OUT/p2.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p3 -- python3 scripts/prof_target.py > $OUT/p3.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/p4 -- python3 scripts/prof_target.py > $OUT/p4.log 2>&1 || true
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 scripts/prof_target.py > $OUT/kt.log 2>&1
find $OUT -name '*counter_collection.csv' | head; find $OUT -name '*kernel_stats.csv' | xargs cat
