cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
time timeout -k 10 900 python -u bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err
rc=$?

tail -5 gpurun_out/r3_bench.err | cut -c1-300
exit $rc
