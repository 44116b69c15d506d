# usage: bash scripts/r3_profiles.sh <tag>  -- the round's records: counter passes for the 2-D and 3-D kernels (separate
# runs, never combined with tracing), the counter record for this build, the default bench (which then reports that
# record), the same command under rocprofv3 --kernel-trace --stats, and the 512^3 bench
set -e
TAG=${1:-r03_final}   # results under gpurun_out/$TAG, to be copied into profiles/r03/
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
NSTEP=6 bash scripts/pmc.sh $TAG/pmc2d > $O/pmc2d.log 2>&1 || { tail -20 $O/pmc2d.log; exit 1; }
python3 scripts/pmc_summary.py $TAG/pmc2d > $O/pmc2d_summary.txt
NX=512 NY=512 NZ=512 NSWEEP=4 NSTEP=2 bash scripts/pmc.sh $TAG/pmc3d > $O/pmc3d.log 2>&1 || { tail -20 $O/pmc3d.log; exit 1; }
python3 scripts/pmc_summary.py $TAG/pmc3d > $O/pmc3d_summary.txt
python3 scripts/record_traffic.py $O/pmc2d_summary.txt profiles/r03/final_pmc_4096sq.txt $O/pmc3d_summary.txt profiles/r03/final_pmc_512cube.txt 512 > $O/traffic.log
cp profiles/traffic.json $O/traffic.json
timeout -k 10 600 python -u bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-300 $O/bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
find $O/prof -name '*kernel_stats.csv' | head -1 | xargs cat > $O/kernel_stats.csv; head -5 $O/kernel_stats.csv
timeout -k 10 600 python -u bench.py --nx 512 --ny 512 --nz 512 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_512cube.json 2> $O/bench_512cube.err || { tail -20 $O/bench_512cube.err; exit 1; }
cut -c1-300 $O/bench_512cube.json
