# usage (GPU box): bash tools/dbg/traffic_ab.sh  - decode-GEMM fabric traffic with and without the XCD-aware tile orders
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/traffic_ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in on off; do
    if [ $v = off ]; then export QL_NO_XCD_SWIZZLE=1; fi
    timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -- python3 -u $R/tools/pmc_traffic.py > $O/fetch_$v.log 2>&1
    timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$v -- python3 -u $R/tools/pmc_traffic.py > $O/write_$v.log 2>&1
    (cd $R && python3 tools/pmc_traffic.py --parse $O/fetch_$v $O/write_$v > $O/traffic_$v.json)
    echo "$v done"
done
