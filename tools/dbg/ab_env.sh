# A/B of an environment switch on the decode-size GEMMs: tools/dbg/ab_env.sh VAR   (runs with VAR unset, then VAR=1)
for v in "" 1; do
  for mode in ${MODES:-w8a16}; do
    echo "== lib ${v:-off} mode $mode"
    env ${v:+$1=1} timeout -k 10 170 python3 -u tools/bench_kernels.py --mode $mode --ms ${MS:-16,32,64,96,128,192,256,384,512} --copies 0 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('%-8s M=%-6d %9.2f us %8.1f TF' % (r['shape'], r['M'], r['us'], r['TFLOPs']))
" || exit 1
  done
done
