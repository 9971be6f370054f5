cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -u -m pytest tests/test_gpu_engine.py tests/test_gpu_allreduce.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3_eng.log 2>&1
rc=$?
grep -v "^  File" gpurun_out/r3_eng.log | tail -12 | cut -c1-250
exit $rc
