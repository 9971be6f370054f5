"""MMLU 5-shot accuracy of a quantized configuration next to the bf16 model (reference: test_mmlu.py,
experiments.py:103-172), on local files:

    python tools/mmlu.py --model DIR --test test.parquet --dev dev.parquet [--quant per_row --dtype int8 --group 128]

`--test` / `--dev` hold the "all" test and dev splits of MMLU as .parquet, .json or .jsonl with the fields
question / choices / answer; DIR needs the tokenizer files.  Prints one JSON line with both accuracies.  Neither
the dataset nor a checkpoint ships with this repository, so quality numbers have to be produced where they exist."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("--test", required=True)
    ap.add_argument("--dev", required=True)
    ap.add_argument("--quant", default="per_row_weight_only")
    ap.add_argument("--dtype", default="int8", choices=["int8", "fp8"])
    ap.add_argument("--group", type=int, default=128)
    ap.add_argument("--items", type=int, default=1000)
    ap.add_argument("--skip-bf16", action="store_true")
    args = ap.parse_args()
    from nanovllm import LLM
    from nanovllm.utils import mmlu
    test, dev = mmlu.load_records(args.test), mmlu.load_records(args.dev)
    configs = [] if args.skip_bf16 else [("bf16", {})]
    configs.append((f"{args.quant}/{args.dtype}", dict(
        quant_type=args.quant, group_size=args.group,
        linear_dtype=torch.int8 if args.dtype == "int8" else torch.float8_e4m3fn)))
    out = {}
    for name, q in configs:
        llm = LLM(args.model, enforce_eager=False, max_model_len=4096, **q)
        out[name] = mmlu.evaluate(llm, test, dev, num_items=args.items)
        del llm
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
