# usage: tools/pmc_gemm.sh <outdir> <ms> <shapes> <counters...>  -- one rocprofv3 --pmc pass over tools/bench_kernels.py
out=$1; ms=$2; shapes=$3; shift 3
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/$out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 -u $GRAFT_REPO_ROOT/tools/bench_kernels.py --mode w8a16 --ms $ms --shapes $shapes --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/$out/run.log 2>&1
echo rc=$?
