cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
L=$PWD/nano-vllm-hw3_amd/nanovllm/_lib
A=${LIBA:-libqlinear.so}; B=${LIBB:-libqlinear_nopk.so}
MS=${MS:-16,64,128,256,2048,16384}
for lib in $A $B; do
 for model in ${MODELS:-4b}; do
  echo "== $lib/$model"
  QLINEAR_LIB=$L/$lib timeout -k 10 300 python -u tools/bench_kernels.py --model $model --mode ${MODE:-group} --ms $MS --iters 40 --tiled || exit 1
 done
done > gpurun_out/r3_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r3_ab.log | A=$A B=$B python3 -c "
import sys, json, os
A,B=os.environ['A'],os.environ['B']
mode=None; t={}
for l in sys.stdin:
    if l.startswith('=='): mode=l.split()[1]; continue
    if not l.startswith('{'): continue
    r=json.loads(l); lib,model=mode.split('/'); t.setdefault((model,r['shape'],r['M']),{})[lib]=r['us']
for k,v in t.items(): print(k, v, round(100*(v[B]/v[A]-1),1),'%')
"
