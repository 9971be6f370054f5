cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
for m in 0.6b 8b; do
  timeout -k 10 500 python -u tools/autotune_plans.py --model $m --mode group --out gpurun_out/r3b_plans_${m}_group.json --log gpurun_out/r3b_tune_${m}_group.jsonl > gpurun_out/r3b_tune_${m}_group.log 2>&1 || { tail -5 gpurun_out/r3b_tune_${m}_group.log; exit 1; }
  tail -1 gpurun_out/r3b_tune_${m}_group.log
done
