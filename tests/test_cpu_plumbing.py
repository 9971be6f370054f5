"""BASELINE configuration 1: the UNQUANTIZED model in fp16 through the whole engine (LLM -> scheduler -> runner ->
Qwen3 -> plain torch operators and torch.matmul) on the CPU, eager, four short sequences - a wiring check that needs
no GPU.  A tiny random Qwen3 checkpoint (config.json + model.safetensors) is synthesized, so the safetensors loader
and the packed q/k/v and gate/up weight loaders run too.  Quantized options are refused on the CPU: that path is HIP only."""
import json
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd"))

CFG = dict(architectures=["Qwen3ForCausalLM"], hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
           num_key_value_heads=1, head_dim=128, intermediate_size=256, vocab_size=512, rms_norm_eps=1e-6,
           rope_theta=1000000.0, max_position_embeddings=1024, tie_word_embeddings=False, attention_bias=False,
           hidden_act="silu", torch_dtype="float16", eos_token_id=511)


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    from safetensors.torch import save_file
    d = tmp_path_factory.mktemp("tiny_fp16")
    with open(d / "config.json", "w") as f:
        json.dump(CFG, f)
    g = torch.Generator().manual_seed(0)
    H, D, I, V = CFG["hidden_size"], CFG["head_dim"], CFG["intermediate_size"], CFG["vocab_size"]
    Hq, Hkv = CFG["num_attention_heads"], CFG["num_key_value_heads"]

    def rnd(*shape, std=0.05):
        return (torch.randn(*shape, generator=g) * std).to(torch.float16)
    t = {"model.embed_tokens.weight": rnd(V, H, std=0.5), "lm_head.weight": rnd(V, H), "model.norm.weight": 1 + rnd(H)}
    for i in range(CFG["num_hidden_layers"]):
        p = f"model.layers.{i}."
        t.update({p + "input_layernorm.weight": 1 + rnd(H), p + "post_attention_layernorm.weight": 1 + rnd(H),
                  p + "self_attn.q_proj.weight": rnd(Hq * D, H), p + "self_attn.k_proj.weight": rnd(Hkv * D, H),
                  p + "self_attn.v_proj.weight": rnd(Hkv * D, H), p + "self_attn.o_proj.weight": rnd(H, Hq * D),
                  p + "self_attn.q_norm.weight": 1 + rnd(D), p + "self_attn.k_norm.weight": 1 + rnd(D),
                  p + "mlp.gate_proj.weight": rnd(I, H), p + "mlp.up_proj.weight": rnd(I, H),
                  p + "mlp.down_proj.weight": rnd(H, I)})
    save_file(t, str(d / "model.safetensors"))
    return str(d), t


def _greedy_reference(t, ids, steps):
    """Greedy continuation by an independent fp32 forward straight from the checkpoint tensors."""
    H, D, Hq, Hkv = CFG["hidden_size"], CFG["head_dim"], CFG["num_attention_heads"], CFG["num_key_value_heads"]
    W = {k: v.float() for k, v in t.items()}

    def rms(v, w):
        return v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-6) * w

    def rope(x, pos):
        inv = 1.0 / (1e6 ** (torch.arange(0, D, 2).float() / D))
        f = pos[:, None].float() * inv[None]
        cos, sin = f.cos()[:, None], f.sin()[:, None]
        x1, x2 = x[..., :D // 2], x[..., D // 2:]
        return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), -1)
    ids = list(ids)
    out = []
    for _ in range(steps):
        T = len(ids)
        pos = torch.arange(T)
        x = F.embedding(torch.tensor(ids), W["model.embed_tokens.weight"])
        for i in range(CFG["num_hidden_layers"]):
            p = f"model.layers.{i}."
            h = rms(x, W[p + "input_layernorm.weight"])
            q = rope(rms((h @ W[p + "self_attn.q_proj.weight"].T).view(T, Hq, D), W[p + "self_attn.q_norm.weight"]), pos)
            k = rope(rms((h @ W[p + "self_attn.k_proj.weight"].T).view(T, Hkv, D), W[p + "self_attn.k_norm.weight"]), pos)
            v = (h @ W[p + "self_attn.v_proj.weight"].T).view(T, Hkv, D)
            o = F.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None],
                                               is_causal=True, enable_gqa=True)[0].transpose(0, 1).reshape(T, -1)
            x = x + o @ W[p + "self_attn.o_proj.weight"].T
            h = rms(x, W[p + "post_attention_layernorm.weight"])
            x = x + (F.silu(h @ W[p + "mlp.gate_proj.weight"].T) * (h @ W[p + "mlp.up_proj.weight"].T)) @ W[p + "mlp.down_proj.weight"].T
        logits = rms(x[-1], W["model.norm.weight"]) @ W["lm_head.weight"].T
        top = logits.topk(2).values
        out.append((int(logits.argmax()), float(top[0] - top[1])))
        ids.append(out[-1][0])
    return out


def test_config1_fp16_cpu_eager_four_short_sequences(ckpt):
    from nanovllm import LLM, SamplingParams
    path, tensors = ckpt
    llm = LLM(path, device="cpu", enforce_eager=True, max_model_len=256, max_num_seqs=4, max_num_batched_tokens=256)
    model = llm.model_runner.model
    assert next(model.parameters()).dtype == torch.float16 and not next(model.parameters()).is_cuda
    assert torch.equal(model.model.layers[1].mlp.gate_up_proj.weight[:256], tensors["model.layers.1.mlp.gate_proj.weight"])
    g = torch.Generator().manual_seed(1)
    prompts = [torch.randint(0, 500, (n,), generator=g).tolist() for n in (5, 17, 2, 9)]
    sps = [SamplingParams(max_tokens=m, ignore_eos=True) for m in (6, 3, 8, 5)]
    out = llm.generate(prompts, sps)
    assert [len(o["token_ids"]) for o in out] == [6, 3, 8, 5]
    assert len(llm.scheduler.block_manager.free) == llm.scheduler.block_manager.num_blocks
    # against an independent fp32 forward from the checkpoint tensors: identical greedy tokens wherever the fp32
    # reference is not nearly tied (fp16 storage of activations moves logits by ~1e-3)
    for p, o in zip(prompts, out):
        ref = _greedy_reference(tensors, p, len(o["token_ids"]))
        for got, (want, margin) in zip(o["token_ids"], ref):
            if got != want:
                assert margin < 2e-2, (got, want, margin)
                break
    # step() keeps the reference's one-step semantics: prefill returns +tokens, decode -batch
    for p in prompts[:2]:
        llm.add_request(p, SamplingParams(max_tokens=2, ignore_eos=True))
    outs, n = llm.step()
    # (counted after the sampled token was appended, as the reference's llm_engine.py:49-53 does)
    assert n == len(prompts[0]) + len(prompts[1]) + 2 and outs == []
    outs, n = llm.step()
    assert n == -2 and sorted(len(t) for _, t in outs) == [2, 2]


def test_cpu_refuses_quantized_options(ckpt):
    from nanovllm import LLM
    path, _ = ckpt
    for kw in (dict(quant_type="per_row", linear_dtype=torch.int8), dict(linear_dtype=torch.float8_e4m3fn),
               dict(quantize_lm_head=True)):
        with pytest.raises(AssertionError):
            LLM(path, device="cpu", **kw)
