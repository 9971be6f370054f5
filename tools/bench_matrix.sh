#!/bin/bash
# One bench line per (model, quantization, dtype) on the same workload: gpurun_out/bench_matrix.jsonl
set -e -o pipefail
out=gpurun_out/bench_matrix.jsonl
mkdir -p gpurun_out && : > $out
run() {
    echo "== $*" >&2
    timeout -k 10 280 python -u bench.py --no-cpu-baseline --no-roofline --no-extra-configs "$@" 2> gpurun_out/bench_matrix_last.err | grep '^{' >> $out
    tail -1 $out | cut -c1-160 >&2
}
run --quant per_row_weight_only --dtype int8
run --quant per_row_weight_only --dtype fp8
run --quant per_group --dtype int8
run --quant per_tensor --dtype int8
run --quant per_row --dtype int8
run --quant per_row --dtype fp8
run --quant bf16
run --model 4b --quant per_row_weight_only --dtype int8
run --model 4b --quant per_group --dtype int8
run --model 8b --quant per_row_weight_only --dtype int8
run --model 8b --quant per_group --dtype int8
