"""Host logic of the MMLU harness (prompt format, item selection, scoring) - known answers written out from the
format of the reference's test_mmlu.py:11-17, 37-49, 61-64."""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))

from nanovllm.utils import mmlu  # noqa: E402


def _rec(i):
    return {"question": f"q{i}?", "choices": [f"a{i}", f"b{i}", f"c{i}", f"d{i}"], "answer": i % 4}


def test_prompt_format_known_answer():
    assert mmlu.format_example("What?", ["x", "y", "z", "w"], "C") == "Question: What?\nA. x\nB. y\nC. z\nD. w\nAnswer: C\n\n"
    dev = [_rec(i) for i in range(7)]
    prompts, answers = mmlu.build_prompts([_rec(100)], dev, num_items=1000)
    want = "The following are multiple choice questions (with answers).\n\n"
    for i in range(5):                                   # only the first five dev items
        want += f"Question: q{i}?\nA. a{i}\nB. b{i}\nC. c{i}\nD. d{i}\nAnswer: {'ABCD'[i % 4]}\n\n"
    # ("Answer: \n\n" with "Answer: \n" replaced by "Answer:" keeps the second newline, as in the reference)
    want += "Question: q100?\nA. a100\nB. b100\nC. c100\nD. d100\nAnswer:\n"
    assert prompts == [want] and answers == ["A"]


def test_item_selection_is_the_seed_42_shuffle():
    test = [_rec(i) for i in range(50)]
    _, answers = mmlu.build_prompts(test, [_rec(0)] * 5, num_items=10)
    ref = list(test)
    random.seed(42)
    random.shuffle(ref)
    assert answers == ["ABCD"[r["answer"]] for r in ref[:10]]


def test_scoring_takes_the_first_character():
    assert mmlu.score([" B", "b) because", "", "   ", "Answer", "C"], ["B", "B", "A", "A", "A", "D"]) == 3


def test_load_records_json_and_jsonl(tmp_path):
    recs = [dict(_rec(i), answer="ABCD"[i % 4] if i % 2 else i % 4) for i in range(4)]
    p1, p2 = tmp_path / "t.json", tmp_path / "t.jsonl"
    p1.write_text(json.dumps(recs))
    p2.write_text("\n".join(json.dumps(r) for r in recs))
    for p in (p1, p2):
        got = mmlu.load_records(str(p))
        assert [g["answer"] for g in got] == [0, 1, 2, 3] and got[2]["choices"] == ["a2", "b2", "c2", "d2"]
