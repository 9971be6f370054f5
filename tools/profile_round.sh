# usage (on the GPU box): bash tools/profile_round.sh <tag>
# Evidence for profiles/: kernel-trace stats of bench.py, the two PMC passes behind roofline.traffic, and two PMC passes
# over the prefill GEMM.  Every step has its own time limit; the script stops at the first failure.
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench.log 2>&1
find $O/bench -name "*kernel_trace.csv" -delete          # (hundreds of thousands of rows; the stats file is what is kept)
grep -c . $(find $O/bench -name "*kernel_stats.csv") | head -1
echo "kernel-trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench4b -- python3 $R/bench.py --model 4b --quant per_group --no-cpu-baseline --no-extra-configs > $O/bench4b.log 2>&1
find $O/bench4b -name "*kernel_trace.csv" -delete
echo "4B g128 kernel-trace done"
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 -u $R/tools/pmc_traffic.py > $O/fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 -u $R/tools/pmc_traffic.py > $O/write.log 2>&1
echo "write pass done"
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pf_sq -- python3 -u $R/tools/pmc_prefill.py > $O/pf_sq.log 2>&1
echo "prefill SQ pass done"
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pf_lds -- python3 -u $R/tools/pmc_prefill.py > $O/pf_lds.log 2>&1
echo "prefill LDS pass done"
cd $R && python3 tools/pmc_traffic.py --parse $O/fetch $O/write > $O/decode_gemm_traffic.json
echo "traffic table written"
