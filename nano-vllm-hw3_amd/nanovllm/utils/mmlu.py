"""MMLU 5-shot harness: the evaluation of the reference's test_mmlu.py (test_mmlu.py:11-66, repeated with
quantization switches in experiments.py:103-172) on records read from LOCAL files - 1000 test items chosen by
``random.seed(42); random.shuffle``, the first five dev items as the few-shot header, one greedy token per item,
first character of the decoded text against the answer letter.  No dataset ships with this repository."""
import json
import random
import time

LETTERS = "ABCD"
HEADER = "The following are multiple choice questions (with answers).\n\n"


def format_example(question, choices, answer):
    """One item as prompt text; ``answer`` is a letter, or "" for the item to be answered."""
    text = f"Question: {question}\n"
    for i, choice in enumerate(choices):
        text += f"{LETTERS[i]}. {choice}\n"
    return text + f"Answer: {answer}\n\n"


def load_records(path):
    """A list of {"question", "choices", "answer"} from a .json (list), .jsonl or .parquet file; ``answer`` may
    be an index 0-3 or a letter."""
    if path.endswith(".parquet"):
        import pandas as pd
        rows = pd.read_parquet(path).to_dict("records")
    elif path.endswith(".jsonl"):
        rows = [json.loads(line) for line in open(path) if line.strip()]
    else:
        rows = json.load(open(path))
    out = []
    for r in rows:
        a = r["answer"]
        a = LETTERS.index(a.strip().upper()) if isinstance(a, str) else int(a)
        out.append({"question": r["question"], "choices": list(r["choices"]), "answer": a})
    return out


def build_prompts(test_records, dev_records, num_items=1000, num_shots=5, seed=42):
    """(prompts, answer letters): shuffle the test split with ``seed``, keep ``num_items``; every prompt is the
    header + the first ``num_shots`` dev items with their answers + the item ending in "Answer:"."""
    items = list(test_records)
    random.Random(seed).shuffle(items)       # same permutation as random.seed(seed); random.shuffle(items)
    items = items[:num_items]
    shots = HEADER
    for ex in list(dev_records)[:num_shots]:
        shots += format_example(ex["question"], ex["choices"], LETTERS[ex["answer"]])
    prompts = [shots + format_example(ex["question"], ex["choices"], "").replace("Answer: \n", "Answer:") for ex in items]
    return prompts, [LETTERS[ex["answer"]] for ex in items]


def score(texts, answers):
    """Number of items whose decoded continuation starts (after stripping) with the answer letter."""
    correct = 0
    for text, want in zip(texts, answers):
        pred = text.strip()[0].upper() if text and text.strip() else ""
        correct += pred == want
    return correct


def evaluate(llm, test_records, dev_records, num_items=1000, num_shots=5, seed=42):
    """Runs the items through ``llm.generate`` (one token each) and returns accuracy (%), counts and throughput."""
    from nanovllm.sampling_params import SamplingParams
    prompts, answers = build_prompts(test_records, dev_records, num_items, num_shots, seed)
    t0 = time.time()
    outputs = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=1))
    dt = time.time() - t0
    correct = score([o["text"] for o in outputs], answers)
    return {"accuracy": 100.0 * correct / max(len(answers), 1), "correct": correct, "total": len(answers),
            "seconds": dt, "samples_per_s": len(answers) / dt if dt > 0 else 0.0}
