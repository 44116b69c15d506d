# usage: bash scripts/final_profiles.sh <tag>  -- the round's records: counter passes for the 2-D and 3-D sweep kernels
# (separate runs, never combined with tracing), the counter record for this build, the default bench (which then
# reports that record), the same command under rocprofv3 --kernel-trace --stats, and the 512^3 bench
set -e
TAG=${1:-final}   # e.g. r02_final: results under gpurun_out/$TAG, to be copied into profiles/<round>/
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
bash scripts/pmc.sh $TAG/pmc2d > $O/pmc2d.log 2>&1 || { tail -20 $O/pmc2d.log; exit 1; }
python3 scripts/pmc_summary.py $TAG/pmc2d > $O/pmc2d_summary.txt
python3 scripts/record_traffic.py $O/pmc2d_summary.txt profiles/r02/final_pmc_4096sq.txt > $O/traffic.log
timeout -k 10 600 python -u bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cut -c1-300 $O/bench.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
find $O/prof -name '*kernel_stats.csv' | head -1 | xargs cat > $O/kernel_stats.csv; cat $O/kernel_stats.csv
NZ=256 bash scripts/pmc.sh $TAG/pmc3d > $O/pmc3d.log 2>&1 || { tail -20 $O/pmc3d.log; exit 1; }
python3 scripts/pmc_summary.py $TAG/pmc3d > $O/pmc3d_summary.txt
grep -A22 "sweep" $O/pmc2d_summary.txt | head -60
timeout -k 10 600 python -u bench.py --nx 512 --ny 512 --nz 512 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_512cube.json 2> $O/bench_512cube.err || { tail -20 $O/bench_512cube.err; exit 1; }
cut -c1-300 $O/bench_512cube.json
