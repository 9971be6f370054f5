# usage: tools/pmc_prefill_run.sh <tag> <counter> [<counter> ...]   (one --pmc pass over tools/pmc_prefill.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p $R/gpurun_out/pmc_pf_$TAG
timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_pf_$TAG -- python3 -u $R/tools/pmc_prefill.py > $R/gpurun_out/pmc_pf_$TAG/run.log 2>&1
echo rc=$?
