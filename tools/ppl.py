"""Perplexity harness: PPL of a quantized configuration next to the bf16 model on the same tokens.

    python tools/ppl.py --model DIR --tokens tokens.json [--quant per_row_weight_only --dtype int8 --group 128]
    python tools/ppl.py --model DIR --text wiki.txt          (needs the tokenizer files in DIR)
    python tools/ppl.py --model DIR --synthetic              (random-init weights + random tokens: a plumbing
                                                              check of the delta, not a quality number)

`tokens.json` is a list of token-id lists.  Prints one JSON line with both perplexities and their difference
(the reference reports the same pair from its WikiText-2 harness, experiments.py:175-260).  No dataset or
checkpoint ships with this repository, so quality numbers have to be produced where they exist."""
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
    ap.add_argument("--tokens")
    ap.add_argument("--text")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--quant", default="per_row_weight_only")
    ap.add_argument("--dtype", default="int8", choices=["int8", "fp8"])
    ap.add_argument("--group", type=int, default=128)
    ap.add_argument("--max-length", type=int, default=4096)
    args = ap.parse_args()
    from nanovllm import LLM
    from nanovllm.utils.ppl import perplexity
    if args.tokens:
        seqs = json.load(open(args.tokens))
    elif args.text:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(args.model)
        seqs = [tok.encode(t, add_special_tokens=False) for t in open(args.text).read().split("\n\n") if t.strip()]
    else:
        assert args.synthetic, "give --tokens, --text or --synthetic"
        g = torch.Generator().manual_seed(0)
        seqs = [torch.randint(0, 10000, (1024,), generator=g).tolist() for _ in range(8)]
    kw = dict(enforce_eager=True, max_model_len=args.max_length, load_format="dummy" if args.synthetic else "auto")
    out = {}
    for name, q in (("bf16", {}), (f"{args.quant}/{args.dtype}", dict(
            quant_type=args.quant, group_size=args.group,
            linear_dtype=torch.int8 if args.dtype == "int8" else torch.float8_e4m3fn))):
        llm = LLM(args.model, **kw, **q)
        out[name] = perplexity(llm.model_runner.model, seqs, args.max_length)
        del llm
        torch.cuda.empty_cache()
    names = list(out)
    out["delta"] = out[names[1]] - out[names[0]]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
