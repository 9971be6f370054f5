# A/B of library builds: tools/dbg/ab_libs.sh A C ""   (suffixes of nanovllm/_lib/libqlinear<suffix>.so; "" = the tree's build)
L=$PWD/nano-vllm-hw3_amd/nanovllm/_lib
for s in "$@"; do
  f=$L/libqlinear${s:+_$s}.so
  for mode in ${MODES:-w8a16 group w8a8}; do
    echo "== lib ${s:-tree} mode $mode"
    QLINEAR_LIB=$f timeout -k 10 170 python3 -u tools/bench_kernels.py --mode $mode --ms 16,64,128,256,512,2048,16384 --copies 4 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('%-8s M=%-6d %9.2f us %8.1f TF' % (r['shape'], r['M'], r['us'], r['TFLOPs']))
" || exit 1
  done
done
