# usage: tools/pmc_gemm.sh <outdir> <counters...>   -- counters for the qkv M=256 decode GEMM
out=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 -u $GRAFT_REPO_ROOT/tools/bench_kernels.py --mode w8a16 --ms 256 --shapes qkv --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/$out/run.log 2>&1
echo rc=$?
