cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -u tools/autotune_plans.py --model 4b --mode group --out gpurun_out/r3b_plans_4b_group.json --log gpurun_out/r3b_tune_4b_group.jsonl > gpurun_out/r3b_tune_4b_group.log 2>&1 || { tail -5 gpurun_out/r3b_tune_4b_group.log; exit 1; }
tail -1 gpurun_out/r3b_tune_4b_group.log
timeout -k 10 900 python -u tools/autotune_plans.py --model 4b --mode group --prefill --out gpurun_out/r3b_plans_4b_group_prefill.json --log gpurun_out/r3b_tune_4b_group_prefill.jsonl > gpurun_out/r3b_tune_4b_group_prefill.log 2>&1 || { tail -5 gpurun_out/r3b_tune_4b_group_prefill.log; exit 1; }
tail -1 gpurun_out/r3b_tune_4b_group_prefill.log
