set -e
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout -k 10 1000 python -u tools/autotune_plans.py --model 0.6b --mode row --forms rope,silu --iters 30 --out gpurun_out/r3_plans_06b_row.json --log gpurun_out/r3_tune_06b_row.jsonl > gpurun_out/r3_tune_06b_row.log 2>&1
tail -2 gpurun_out/r3_tune_06b_row.log
