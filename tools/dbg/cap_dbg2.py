import os, sys, json, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-hw3_amd")); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29718")
dist.init_process_group("nccl", rank=0, world_size=1)
from nanovllm import LLM, SamplingParams
roles = sys.argv[1].split(",")
cfg = dict(architectures=["Qwen3ForCausalLM"], hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
           num_key_value_heads=2, head_dim=128, intermediate_size=512, vocab_size=1024, rms_norm_eps=1e-6,
           rope_theta=1000000.0, max_position_embeddings=2048, tie_word_embeddings=True, attention_bias=False,
           hidden_act="silu", torch_dtype="bfloat16")
d = tempfile.mkdtemp()
json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
llm = LLM(d, load_format="dummy", enforce_eager=True, max_model_len=512, max_num_seqs=16, quant_type="per_row_weight_only", linear_dtype=torch.int8)
sp = SamplingParams(max_tokens=8, ignore_eos=True)
prompts = [[1, 2, 3, 4, 5], [7, 8, 9]]
before = [o["token_ids"] for o in llm.generate(prompts, sp)]
runner = llm.model_runner; model = runner.model
if "embed" in roles: model.model.embed_tokens.tp_size = 2
if "row" in roles:
    for layer in model.model.layers:
        layer.self_attn.o_proj._init_tp(1, 0, 2); layer.mlp.down_proj._init_tp(1, 0, 2)
if "head" in roles: model.lm_head.force_exchange = True
print("eager with roles", [o["token_ids"] for o in llm.generate(prompts, sp)] == before, flush=True)
if "tl" in roles: runner.world_size = 2
print("capturing", roles, flush=True)
runner.capture_graphs()
print("captured", flush=True)
runner.enforce_eager = False
print("graph tokens equal", [o["token_ids"] for o in llm.generate(prompts, sp)] == before, flush=True)
dist.destroy_process_group()
