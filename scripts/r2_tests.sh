# all GPU tests in one process, then the no-alloc test alone in a fresh process
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_tests
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -x -q --timeout 300 --timeout-method thread -m gpu > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 200 python -u -m pytest tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > $O/no_alloc.log 2>&1 || { tail -40 $O/no_alloc.log; exit 1; }
tail -2 $O/no_alloc.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
