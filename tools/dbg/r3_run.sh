cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 200 --timeout-method thread -k "group or stream_k or silu or rope" > gpurun_out/r3_grp_tests.log 2>&1 || { tail -30 gpurun_out/r3_grp_tests.log; exit 1; }
tail -2 gpurun_out/r3_grp_tests.log
L=$PWD/nano-vllm-hw3_amd/nanovllm/_lib
MS=16,64,128,256,2048,16384
for lib in libqlinear_old.so libqlinear.so; do
 for model in 4b 0.6b; do
  echo "== $lib/$model"
  QLINEAR_LIB=$L/$lib timeout -k 10 300 python -u tools/bench_kernels.py --model $model --mode group --ms $MS --iters 40 --tiled || exit 1
 done
done > gpurun_out/r3_grp_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r3_grp_ab.log | python3 -c "
import sys, json
mode=None; t={}
for l in sys.stdin:
    if l.startswith('=='): mode=l.split()[1]; continue
    if not l.startswith('{'): continue
    r=json.loads(l); lib,model=mode.split('/'); t.setdefault((model,r['shape'],r['M']),{})[lib]=r['us']
for k,v in t.items(): print(k, v, round(100*(v['libqlinear.so']/v['libqlinear_old.so']-1),1),'%')
"
