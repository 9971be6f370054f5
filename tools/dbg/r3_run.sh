cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout -k 10 400 python -u -m pytest tests/test_gpu_engine.py -x -q --timeout 200 --timeout-method thread -k "two_ranks" > gpurun_out/r3_ar.log 2>&1
rc=$?
tail -40 gpurun_out/r3_ar.log | cut -c1-250
exit $rc
