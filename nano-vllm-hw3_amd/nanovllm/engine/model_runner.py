"""Runs the model for one scheduler step on one GPU.  With tensor parallelism every rank is its own
process (launched by torch.distributed.run) executing the SAME control flow on the same inputs; the
only cross-rank traffic is the RCCL collectives inside the layers, so no RPC channel is needed."""
import torch
import torch.distributed as dist

from nanovllm.config import Config
from nanovllm.engine.sequence import Sequence
from nanovllm.models.qwen3 import AttnContext, Qwen3ForCausalLM, set_context
from nanovllm.utils.loader import init_dummy_weights, load_model
from nanovllm.utils.quantization import apply_quant_from_config, reserve_workspace


class ModelRunner:
    def __init__(self, config: Config, rank: int = 0):
        self.config = config
        hf = config.hf_config
        self.block_size = config.kvcache_block_size
        self.enforce_eager = config.enforce_eager
        self.world_size = config.tensor_parallel_size
        self.rank = rank
        self.on_gpu = config.device == "cuda"
        self.device = torch.device("cuda", torch.cuda.current_device()) if self.on_gpu else torch.device("cpu")
        default_dtype = torch.get_default_dtype()
        torch.set_default_dtype(hf.torch_dtype)
        torch.set_default_device(self.device)
        try:
            self.model = Qwen3ForCausalLM(hf)
            if config.load_format == "quantized":
                # a pre-quantized checkpoint: no bf16 read, no re-quantization
                from nanovllm.utils.checkpoint import load_quantized
                qc = load_quantized(self.model, config.model)
                config.quant_type, config.group_size = qc["quant_type"], qc["group_size"] or config.group_size
                config.linear_dtype = getattr(torch, qc["linear_dtype"])
            else:
                if config.load_format == "dummy":
                    init_dummy_weights(self.model, hf, seed=0)
                else:
                    load_model(self.model, config.model)
                # quantization happens after loading, before warm-up and graph capture (reference:
                # nanovllm/engine/model_runner.py:36-57)
                apply_quant_from_config(self.model, config)
            self.graphs = {}
            if not self.on_gpu:
                # plumbing mode: unquantized model, plain torch operators, eager, no staging buffers
                self.allocate_kv_cache()
                self._stage_max_bs = self._stage_max_blocks = 0
                self._stage_rows = []
                self._tok_host = [torch.zeros(0, dtype=torch.int64)]
                return
            self._reserve_gemm_workspace()
            self.allocate_kv_cache()
            from nanovllm import ops
            ops.reserve_attn_workspace(min(config.max_num_seqs, 512), hf.num_attention_heads // self.world_size,
                                       hf.num_key_value_heads // self.world_size, self.block_size,
                                       (config.max_model_len + self.block_size - 1) // self.block_size, self.device)
            self._init_decode_staging(min(config.max_num_seqs, 512),
                                      (config.max_model_len + self.block_size - 1) // self.block_size)
            if not self.enforce_eager:
                # no eager fallback: a capture that fails (e.g. a collective this RCCL build cannot capture)
                # raises, and the caller decides (enforce_eager=True)
                self.capture_graphs()
                from nanovllm import _qlinear
                _qlinear.workspace.pin(self.device)        # the graphs address these buffers: never freed
                _qlinear.attn_workspace.pin(self.device)
                _qlinear.head_workspace.pin(self.device)
        finally:
            torch.set_default_device("cpu")
            torch.set_default_dtype(default_dtype)

    # ------------------------------------------------------------------ setup
    def _linear_shapes(self):
        shapes = set()
        for l in self.model.model.layers:
            for m in (l.self_attn.qkv_proj, l.self_attn.o_proj, l.mlp.gate_up_proj, l.mlp.down_proj):
                w = getattr(m, "qw", None)
                if w is not None:
                    shapes.add((w.shape[0], w.shape[1]))
        head = getattr(self.model.lm_head, "quantized", None)
        if head is not None:
            shapes.add((head.qw.shape[0], head.qw.shape[1]))
        return sorted(shapes)

    def _reserve_gemm_workspace(self):
        # over every row count a call can bring: decode buckets AND eager prefill batches.  The decode hipGraphs
        # bake the buffer's address in; sized here it is never outgrown by this engine's own calls, and
        # _Workspace.pin keeps it alive should another engine of the process need a larger one.
        shapes = self._linear_shapes()
        if shapes:
            reserve_workspace(max(self.config.max_num_batched_tokens, self.config.max_num_seqs), shapes, self.device)

    def allocate_kv_cache(self):
        cfg, hf = self.config, self.config.hf_config
        num_kv_heads = hf.num_key_value_heads // self.world_size
        block_bytes = 2 * hf.num_hidden_layers * self.block_size * num_kv_heads * hf.head_dim * hf.torch_dtype.itemsize
        if self.on_gpu:
            free, total = torch.cuda.mem_get_info()
            # activations of the largest prefill batch stay well under 10 GiB for every supported size
            budget = int(free - (1.0 - cfg.gpu_memory_utilization) * total) - (10 << 30)
        else:
            budget = 1 << 30                                     # host plumbing runs: a small cache is plenty
        max_needed = cfg.max_num_seqs * ((cfg.max_model_len + self.block_size - 1) // self.block_size)
        cfg.num_kvcache_blocks = max(1, min(budget // block_bytes, max_needed))
        if self.world_size > 1:
            # every rank runs its own copy of the scheduler: they must agree on the block budget
            n = torch.tensor([cfg.num_kvcache_blocks], dtype=torch.int64, device="cpu" if dist.get_backend() == "gloo" else self.device)
            dist.all_reduce(n, op=dist.ReduceOp.MIN)
            cfg.num_kvcache_blocks = int(n.item())
        assert cfg.num_kvcache_blocks > 0
        self.kv_cache = torch.empty(2, hf.num_hidden_layers, cfg.num_kvcache_blocks, self.block_size, num_kv_heads,
                                    hf.head_dim)
        for i, layer in enumerate(self.model.model.layers):
            layer.self_attn.k_cache = self.kv_cache[0, i]
            layer.self_attn.v_cache = self.kv_cache[1, i]

    # ------------------------------------------------------------------ per-step input preparation
    def _dev(self, data, dtype):
        if not self.on_gpu:
            return torch.tensor(data, dtype=dtype)
        return torch.tensor(data, dtype=dtype, pin_memory=True).to(self.device, non_blocking=True)

    def prepare_prefill(self, seqs):
        input_ids, positions, slot_mapping, cu = [], [], [], [0]
        # (re)prefilled sequences may own different blocks than the decode staging rows remember
        self._stage_rows = [[None] * self._stage_max_bs for _ in range(2)] if self.on_gpu else []
        for seq in seqs:
            n = len(seq)
            input_ids.extend(seq.token_ids)
            positions.extend(range(n))
            cu.append(cu[-1] + n)
            for i, blk in enumerate(seq.block_table):
                start = blk * self.block_size
                slot_mapping.extend(range(start, start + seq.tokens_in_block(i, self.block_size)))
        if not slot_mapping:           # warm-up sequences own no blocks
            slot = None
        else:
            slot = self._dev(slot_mapping, torch.int32)
        set_context(AttnContext(is_prefill=True, cu_seqlens=cu, slot_mapping=slot,
                                cu_seqlens_dev=self._dev(cu, torch.int32), max_seqlen=max(len(s) for s in seqs)))
        return self._dev(input_ids, torch.int64), self._dev(positions, torch.int64), cu

    def _init_decode_staging(self, max_bs, max_blocks):
        """All per-step decode inputs live in ONE int32 device buffer (ids and positions as int64 pairs,
        slot mapping, context lengths, row permutation, block tables) mirrored by pinned host buffers: a
        decode step costs a single host-to-device copy and no fill or scatter launches.  The hipGraph
        inputs are views of the device buffer.  There are TWO host mirrors, used alternately, because a
        step is prepared while the copy of the step before it may still be queued behind the GPU's work."""
        words = max_bs * (2 + 2 + 1 + 1 + 1 + max_blocks)
        self._stage_dev = torch.zeros(words, dtype=torch.int32, device=self.device)

        def views(buf):
            o = 0
            ids = buf[o:o + 2 * max_bs].view(torch.int64); o += 2 * max_bs
            pos = buf[o:o + 2 * max_bs].view(torch.int64); o += 2 * max_bs
            slots = buf[o:o + max_bs]; o += max_bs
            ctx = buf[o:o + max_bs]; o += max_bs
            perm = buf[o:o + max_bs]; o += max_bs
            tables = buf[o:o + max_bs * max_blocks].view(max_bs, max_blocks)
            return ids, pos, slots, ctx, tables, perm
        self._stage_max_bs, self._stage_max_blocks = max_bs, max_blocks
        self._stage_host, self._hvs, self._stage_rows, self._stage_prev_bs = [], [], [], []
        for _ in range(2):
            host = torch.zeros(words, dtype=torch.int32, device="cpu", pin_memory=True)
            hv = tuple(v.numpy() for v in views(host))
            hv[2].fill(-1)
            self._stage_host.append(host)
            self._hvs.append(hv)
            self._stage_rows.append([None] * max_bs)   # (seq_id, len(block_table)) whose table each row holds
            self._stage_prev_bs.append(0)
        self._dv = views(self._stage_dev)
        self._stage_turn = 0
        # sampled tokens travel device -> host through pinned buffers, alternately as well
        self._tok_host = [torch.zeros(max(max_bs, self.config.max_num_seqs), dtype=torch.int64, device="cpu",
                                      pin_memory=True) for _ in range(2)]
        self._tok_turn = 0

    def can_stage(self, seqs):
        """True when a decode step over ``seqs`` fits the staging buffers (what pipelined decode needs)."""
        return (self.on_gpu and len(seqs) <= self._stage_max_bs and
                all(len(s.block_table) + 1 <= self._stage_max_blocks for s in seqs))

    def prepare_decode(self, seqs, perm=None):
        """Fills a pinned mirror and issues the one copy; returns device views of the first len(seqs)
        rows.  Rows past the batch keep slot -1 / context 0 (what a padded graph bucket needs).
        ``perm[r]`` = row that sequence r had in the step whose tokens are still on the device."""
        bs = len(seqs)
        if bs > self._stage_max_bs or max(len(s.block_table) for s in seqs) > self._stage_max_blocks:
            assert perm is None
            return self._prepare_decode_unstaged(seqs)
        turn = self._stage_turn
        self._stage_turn = turn ^ 1
        h_ids, h_pos, h_slots, h_ctx, h_tab, h_perm = self._hvs[turn]
        bsz = self.block_size
        lens = [len(s) for s in seqs]
        h_ids[:bs] = [s.last_token for s in seqs]
        h_ctx[:bs] = lens
        h_pos[:bs] = h_ctx[:bs]
        h_pos[:bs] -= 1
        h_slots[:bs] = [s.block_table[-1] * bsz + (n - 1) % bsz for s, n in zip(seqs, lens)]
        if perm is not None:
            h_perm[:bs] = perm
        prev = self._stage_prev_bs[turn]
        if prev > bs:
            h_slots[bs:prev] = -1
            h_ctx[bs:prev] = 0
        self._stage_prev_bs[turn] = bs
        rows = self._stage_rows[turn]
        for i, s in enumerate(seqs):
            bt = s.block_table
            key = (s.seq_id, len(bt))
            if rows[i] != key:
                rows[i] = key
                h_tab[i, :len(bt)] = bt
        self._stage_dev.copy_(self._stage_host[turn], non_blocking=True)
        d = self._dv
        return d[0][:bs], d[1][:bs], d[2][:bs], d[3][:bs], d[4][:bs]

    def _prepare_decode_unstaged(self, seqs):
        input_ids = [s.last_token for s in seqs]
        positions = [len(s) - 1 for s in seqs]
        context_lens = [len(s) for s in seqs]
        slot_mapping = [s.block_table[-1] * self.block_size + (len(s) - 1) % self.block_size for s in seqs]
        width = max(len(s.block_table) for s in seqs)
        tables = [s.block_table + [0] * (width - len(s.block_table)) for s in seqs]
        return (self._dev(input_ids, torch.int64), self._dev(positions, torch.int64),
                self._dev(slot_mapping, torch.int32), self._dev(context_lens, torch.int32),
                self._dev(tables, torch.int32))

    # ------------------------------------------------------------------ execution
    @torch.inference_mode()
    def launch(self, seqs, is_prefill: bool, prev=None):
        """Enqueue one step and the copy of its sampled tokens to the host; returns a handle for wait().
        ``prev`` = handle of the decode step before this one when its tokens have not been read on the host
        yet: the input ids of this step are then gathered on the device from that step's output (every
        sequence of this step must have been part of it), so the host never waits between two steps."""
        tokens = logits = None
        if is_prefill:
            assert prev is None
            ids, pos, cu = self.prepare_prefill(seqs)
            h = self.model(ids, pos)
            last = torch.tensor(cu[1:], device=self.device) - 1
            rows = h[last].contiguous()
            if self.on_gpu and self.model.lm_head.fused_sampler(rows):
                tokens = self.model.compute_tokens(rows)          # sampler in the projection's epilogue
            else:
                logits = self.model.compute_logits(rows)
        else:
            perm = None
            if prev is not None:
                row = {s.seq_id: i for i, s in enumerate(prev["seqs"])}
                perm = [row[s.seq_id] for s in seqs]
            ids, pos, slots, ctx_lens, tables = self.prepare_decode(seqs, perm)
            bs = ids.numel()
            if prev is not None:
                torch.index_select(prev["tokens"], 0, self._dv[5][:bs], out=ids)
            if self.enforce_eager or bs > self.max_graph_bs:
                set_context(AttnContext(False, None, slots, ctx_lens, tables))
                h = self.model(ids, pos)
                if self.on_gpu and self.model.lm_head.fused_sampler(h):
                    tokens = self.model.compute_tokens(h)
                else:
                    logits = self.model.compute_logits(h)
            else:
                gbs = next(x for x in self.graph_bs if x >= bs)
                gv = self.graph_vars
                self.graphs[gbs].replay()
                if "tokens" in gv:
                    # the graph's own buffer: its readers (the copy to the host below, the gather into the next
                    # step's input ids) are enqueued on this stream before the next replay overwrites it
                    tokens = gv["tokens"][:bs]
                else:
                    logits = gv["logits"][:bs]
        # greedy sampler (layers/sampler.py:10-12)
        if not self.on_gpu:
            return dict(seqs=list(seqs), tokens=logits.float().argmax(dim=-1), host=None, event=None)
        if tokens is not None:
            pass
        elif logits.dtype == torch.bfloat16 and logits.shape[-1] % 8 == 0 and logits.is_contiguous():
            from nanovllm import ops
            tokens = ops.argmax_rows(logits)
        else:
            tokens = logits.argmax(dim=-1)
        n = tokens.numel()
        if n <= self._tok_host[0].numel():
            host = self._tok_host[self._tok_turn][:n]
            self._tok_turn ^= 1
            host.copy_(tokens, non_blocking=True)
        else:
            host = None
            tokens = tokens.clone()          # read at wait(): must not alias a buffer a later step rewrites
        ev = torch.cuda.Event()
        ev.record()
        return dict(seqs=list(seqs), tokens=tokens, host=host, event=ev)

    @staticmethod
    def wait(handle):
        """The token ids of a launched step (blocks until the GPU has produced them)."""
        if handle["host"] is None:
            return handle["tokens"].tolist()
        handle["event"].synchronize()
        return handle["host"].tolist()

    def run(self, seqs, is_prefill: bool):
        return self.wait(self.launch(seqs, is_prefill))

    @torch.no_grad()
    def capture_graphs(self):
        # (no_grad, not inference_mode: the first capture of a process creates the RNG bookkeeping tensors that
        # every later capture updates in place, and as inference tensors they would make a capture that anybody
        # starts outside inference mode fail for as long as these graphs live)
        # (the buffers the graphs address are created here: on this runner's device and in the model's dtype, whoever
        # calls - the constructor, or a caller that re-captures after changing the model)
        prev_dtype, prev_device = torch.get_default_dtype(), torch.get_default_device()
        torch.set_default_dtype(self.config.hf_config.torch_dtype)
        torch.set_default_device(self.device)
        try:
            self._capture_graphs()
        finally:
            torch.set_default_device(prev_device)
            torch.set_default_dtype(prev_dtype)

    def _capture_graphs(self):
        cfg, hf = self.config, self.config.hf_config
        max_bs = min(cfg.max_num_seqs, 512)
        self.max_graph_bs = max_bs
        max_blocks = (cfg.max_model_len + self.block_size - 1) // self.block_size
        d = self._dv
        assert self._stage_max_bs >= max_bs and self._stage_max_blocks >= max_blocks
        gv = dict(input_ids=d[0], positions=d[1], slot_mapping=d[2], context_lens=d[3], block_tables=d[4])
        probe = torch.zeros(1, hf.hidden_size)
        fused = self.model.lm_head.fused_sampler(probe)
        if fused:
            gv["tokens"] = torch.zeros(max_bs, dtype=torch.int64)     # the graphs end in the sampler
            from nanovllm import ops
            ops.reserve_head_workspace(max_bs, self.model.lm_head.weight.shape[0], self.device)
        else:
            gv["logits"] = torch.zeros(max_bs, hf.vocab_size)

        def step(bs):
            h = self.model(gv["input_ids"][:bs], gv["positions"][:bs])
            if fused:
                self.model.compute_tokens(h, out=gv["tokens"][:bs])
            else:
                gv["logits"][:bs] = self.model.compute_logits(h)

        self.graph_bs = [1, 2, 4, 8] + list(range(16, max_bs + 1, 16))
        pool = None
        for bs in reversed(self.graph_bs):
            set_context(AttnContext(False, None, gv["slot_mapping"][:bs], gv["context_lens"][:bs], gv["block_tables"][:bs]))
            step(bs)                                                  # warm-up
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # tensor-parallel ranks: RCCL's watchdog thread polls events while this thread captures; a capture in
            # "thread_local" mode is not invalidated by what other threads of the process do
            mode = "thread_local" if self.world_size > 1 else "global"
            with torch.cuda.graph(g, pool, capture_error_mode=mode):
                step(bs)
            if pool is None:
                pool = g.pool()
            self.graphs[bs] = g
        torch.cuda.synchronize()
        self.graph_vars = gv
