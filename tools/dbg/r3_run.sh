cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread -k "fp16_fp32 or errors_are_loud or empty_and" > gpurun_out/r3_wide.log 2>&1
rc=$?
grep -v "^  File" gpurun_out/r3_wide.log | tail -30 | cut -c1-300
exit $rc
