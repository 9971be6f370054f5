cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout -k 10 400 python -u bench.py --model 4b --quant per_group --no-cpu-baseline --no-extra-configs > gpurun_out/r3b_bench4b.json 2> gpurun_out/r3b_bench4b.err || { tail -5 gpurun_out/r3b_bench4b.err; exit 1; }
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r3b_bench4b.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline'].get('frac_of_roofline'))"
