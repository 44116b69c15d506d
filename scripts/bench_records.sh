# usage: bash scripts/bench_records.sh <tag>  -- the bench lines kept under profiles/ besides the default one
set -e
O=gpurun_out/${1:-records}
mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python -u bench.py "$@" > $O/$name.json 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }; cut -c1-200 $O/$name.json; }
run bench_4096sq_1000steps --steps 1000 --warmup 20 --no-cpu-baseline
run bench_4096sq_strict --arith strict --steps 40 --warmup 5 --no-cpu-baseline
run bench_8192sq --nx 8192 --ny 8192 --steps 50 --warmup 5 --no-cpu-baseline
run bench_cavity_2048x16384_slab --case cavity --nx 2048 --ny 16384 --steps 100 --warmup 10 --no-cpu-baseline
run bench_512cube --nx 512 --ny 512 --nz 512 --steps 20 --warmup 3 --no-cpu-baseline
run bench_256cube --nx 256 --ny 256 --nz 256 --steps 100 --warmup 10 --no-cpu-baseline
