cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
L=$PWD/nano-vllm-hw3_amd/nanovllm/_lib
for v in ${VARS}; do
  QLINEAR_LIB=$L/libqlinear_$v.so timeout -k 10 120 python -u tools/dbg/allrows_time.py 2>&1 | grep -v amdgpu.ids || exit 1
done
