# strict kernels: bit-exactness tests, then timing of the launch shapes (same GPU, alternating)
set -e
mkdir -p gpurun_out/strict
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_steps.py tests/test_gpu_fuzz.py -x -q -m gpu --timeout 400 --timeout-method thread -k "bit_exact or strict or fused_filter or steps_filter or golden or inactive or random or lock_step" > gpurun_out/strict/tests.log 2>&1 || { tail -30 gpurun_out/strict/tests.log; exit 1; }
tail -3 gpurun_out/strict/tests.log
DIM=2 timeout -k 10 200 python scripts/exp/kstrict.py -1:32 -1:32 2>&1 | tee gpurun_out/strict/k2.log
DIM=3 N=256 timeout -k 10 300 python scripts/exp/kstrict.py -1:32 -1:32 2>&1 | tee gpurun_out/strict/k3.log
