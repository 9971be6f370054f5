# usage: tools/dbg/build_variant.sh <suffix> <extra hipcc flags...>   (group translation units rebuilt with the flags)
set -e
S=$1; shift
D=/root/repo/nano-vllm-hw3_amd
mkdir -p $D/build_$S
for f in ql_gemm_inst_bf16_i8_group ql_gemm_inst_bf16_fp8_group; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-gpu-rdc "$@" -c $D/csrc/$f.hip -o $D/build_$S/$f.o &
done
wait
OBJS=$(ls $D/build/*.o | grep -v -e ql_gemm_inst_bf16_i8_group.o -e ql_gemm_inst_bf16_fp8_group.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/nanovllm/_lib/libqlinear_$S.so $OBJS $D/build_$S/*.o
ls -la $D/nanovllm/_lib/libqlinear_$S.so
