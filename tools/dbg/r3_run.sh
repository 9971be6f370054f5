cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
for B in 32 64 131 256; do
 for S in 1 2 4 8; do
  echo "B=$B SPW=$S"; BATCH=$B QL_ATTN_SPW=$S timeout -k 10 120 python -u tools/bench_attn.py || exit 1
 done
done > gpurun_out/r3_attn_spw.log 2>&1
echo HQ32 >> gpurun_out/r3_attn_spw.log
for B in 131 256; do
 for S in 1 8; do
  echo "B=$B SPW=$S HQ=32"; HQ=32 BATCH=$B QL_ATTN_SPW=$S timeout -k 10 120 python -u tools/bench_attn.py || exit 1
 done
done >> gpurun_out/r3_attn_spw.log 2>&1
cat gpurun_out/r3_attn_spw.log
