cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
MS=16,32,64,96,128,192,256,384,512
for mode in bf16 w8a16 group w8a8; do
  echo "== $mode"
  timeout -k 10 200 python -u tools/bench_kernels.py --model 4b --mode $mode --ms $MS --iters 40 --tiled || exit 1
done > gpurun_out/r3_midm_calib.log 2>&1
grep -v amdgpu.ids gpurun_out/r3_midm_calib.log | python3 -c "
import sys, json
mode=None; t={}
for l in sys.stdin:
    if l.startswith('=='): mode=l.split()[1]; continue
    if not l.startswith('{'): continue
    r=json.loads(l); t.setdefault((r['shape'],r['M']),{})[mode]=r['us']
for k,v in t.items(): print(k, v)
"
