"""Perplexity of a loaded model over token sequences: the evaluation loop of the reference's
test_ppl.py (test_ppl.py:36-72: windows of up to ``max_length`` tokens, next-token cross entropy summed
over all windows, exp of the mean), run through this build's prefill path (varlen attention kernel,
quantized linears) instead of a separate simple-attention model."""
import math

import torch
import torch.nn.functional as F

from nanovllm.models.qwen3 import AttnContext, set_context


@torch.inference_mode()
def total_nll(model, token_seqs, max_length: int = 4096, device="cuda"):
    """Sum of next-token negative log-likelihoods (nats) and the number of predicted tokens."""
    loss, count = 0.0, 0
    for toks in token_seqs:
        for i in range(0, len(toks) - 1, max_length):
            chunk = toks[i:min(i + max_length + 1, len(toks))]
            if len(chunk) < 2:
                continue
            ids = torch.tensor(chunk[:-1], device=device)
            tgt = torch.tensor(chunk[1:], device=device)
            set_context(AttnContext(is_prefill=True, cu_seqlens=[0, ids.numel()], slot_mapping=None))
            hidden = model(ids, torch.arange(ids.numel(), device=device))
            logits = model.compute_logits(hidden).float()
            loss += F.cross_entropy(logits, tgt, reduction="sum").item()
            count += tgt.numel()
    return loss, count


def perplexity(model, token_seqs, max_length: int = 4096):
    loss, count = total_nll(model, token_seqs, max_length)
    return math.exp(loss / max(count, 1))
