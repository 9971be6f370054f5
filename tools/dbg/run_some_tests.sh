cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py -x -q --timeout 200 --timeout-method thread -k "${K:-group}" > gpurun_out/some_tests.log 2>&1; rc=$?
grep -v "^  File" gpurun_out/some_tests.log | tail -${TAIL:-25} | cut -c1-250
exit $rc
