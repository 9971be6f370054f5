import os
from glob import glob

import torch


def default_weight_loader(param, loaded_weight):
    param.data.copy_(loaded_weight)


def load_model(model, path: str):
    """Read every *.safetensors under ``path`` and hand each tensor to its parameter's
    ``weight_loader`` (q/k/v and gate/up go into the packed parameters with their shard id), as the
    reference's nanovllm/utils/loader.py does."""
    from safetensors import safe_open
    packed = getattr(model, "packed_modules_mapping", {})
    params = dict(model.named_parameters())
    for file in sorted(glob(os.path.join(path, "*.safetensors"))):
        with safe_open(file, "pt", "cpu") as f:
            for name in f.keys():
                for k, (v, shard_id) in packed.items():
                    if k in name:
                        pname = name.replace(k, v)
                        if pname in params:
                            p = params[pname]
                            p.weight_loader(p, f.get_tensor(name), shard_id)
                        break
                else:
                    if name in params:
                        p = params[name]
                        getattr(p, "weight_loader", default_weight_loader)(p, f.get_tensor(name))


@torch.no_grad()
def init_dummy_weights(model, seed: int = 0, std: float = 0.02):
    """Random-initialise every parameter in place (benchmarks without a checkpoint): linear and
    embedding weights ~ N(0, std), norm weights = 1.  Same seed on every TP rank draws the FULL
    tensor's shard deterministically only at tp=1; for tp>1 each rank seeds with its rank, which is
    fine for throughput measurement (the values do not affect the work done)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    for name, p in model.named_parameters():
        if p.dim() == 1:
            p.data.fill_(1.0)
        else:
            p.data.copy_(torch.randn(p.shape, generator=g, device="cuda", dtype=torch.float32) * std)
