# kernel trace of the strict launch shapes (2-D 4096^2, 3-D 256^3)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/strict
DIM=2 ROUNDS=2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/strict/kt2 -o kt2 -- python3 $R/scripts/exp/kstrict.py -1:32 8:32 > $R/gpurun_out/strict/kt2.log 2>&1
DIM=3 ROUNDS=2 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/strict/kt3 -o kt3 -- python3 $R/scripts/exp/kstrict.py -1:32 2:32 > $R/gpurun_out/strict/kt3.log 2>&1
