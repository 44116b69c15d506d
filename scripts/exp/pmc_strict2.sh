# issue-slot counters of the default strict kernels (2-D 4096^2, 3-D 256^3): one counter pass each, no tracing
set -e
OUT=gpurun_out/strict2
mkdir -p $OUT
export TMPDIR=/tmp FLAGS=0 NSWEEP=6 NSTEP=2
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"
timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/d2 -- python3 scripts/prof_target.py > $OUT/d2.log 2>&1
NZ=256 timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/d3 -- python3 scripts/prof_target.py > $OUT/d3.log 2>&1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t2 -- python3 scripts/prof_target.py > $OUT/t2.log 2>&1
NZ=256 timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t3 -- python3 scripts/prof_target.py > $OUT/t3.log 2>&1
python3 scripts/pmc_summary.py strict2/d2
python3 scripts/pmc_summary.py strict2/d3
find $OUT/t2 $OUT/t3 -name '*kernel_stats.csv' | xargs cat | cut -d, -f1,2,4 | cut -c1-120
