cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3_full.log 2>&1
rc=$?
tail -15 gpurun_out/r3_full.log
exit $rc
