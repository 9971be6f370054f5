"""Tensor-parallel host logic on CPU with world_size-2/4 gloo process groups: weight-loader sharding
against the reference's own loaders (golden vectors), and the sharded forward of the bf16 and the
quantized modules (column-parallel = concatenation, row-parallel = all-reduce of partials, bias on
rank 0 only) against the unsharded result.

The quantized modules have no CPU implementation (the product path is HIP only), so in these
processes the two operator entry points they call are replaced by oracle-backed TEST DOUBLES; what is
under test here is the sharding / collective logic around them, not the arithmetic."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, "nano-vllm-hw3_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _install_oracle_doubles():
    """CPU stand-ins for the HIP operators, built on the oracle (test infrastructure only)."""
    from oracle import qlinear_oracle as O
    from nanovllm.utils import quantization as NQ

    def bits(t):
        return t.contiguous().view(torch.int16).numpy().view(np.uint16)

    def per_row_quant(x, dtype):
        q, s = O.per_row_quant(bits(x), "int8")
        return torch.from_numpy(q), torch.from_numpy(s)

    def w8a16(x, qw, sw, bias, dtype, silu=False, out=None):
        assert not silu
        x2 = x.reshape(-1, x.shape[-1])
        y = O.w8a16_rowscale_matmul(bits(x2), qw.numpy(), sw.numpy(), None if bias is None else bits(bias), "int8")
        y = torch.from_numpy(y.view(np.int16)).view(torch.bfloat16)
        if out is not None:
            out.copy_(y)
            return out
        return y.reshape(*x.shape[:-1], qw.shape[0])

    NQ.per_row_quant = per_row_quant
    NQ.per_row_weight_only_matmul = w8a16


def _worker_layout(rank, world, port, gold_path):
    _setup(rank, world, port)
    try:
        from nanovllm.layers import linear as L
        g = np.load(gold_path)
        hidden, head, nh, nkv, inter = 32, 8, 8, 4, 48
        full = {k: torch.from_numpy(g[f"tp/full/{k}"].astype(np.float32)) for k in ("q", "k", "v", "o", "gate", "up", "down")}
        qkv = L.QKVParallelLinear(hidden, head, nh, nkv)
        for sid in "qkv":
            qkv.weight_loader(qkv.weight, full[sid], sid)
        gu = L.MergedColumnParallelLinear(hidden, [inter, inter])
        gu.weight_loader(gu.weight, full["gate"], 0)
        gu.weight_loader(gu.weight, full["up"], 1)
        o = L.RowParallelLinear(nh * head, hidden)
        o.weight_loader(o.weight, full["o"])
        dn = L.RowParallelLinear(inter, hidden)
        dn.weight_loader(dn.weight, full["down"])
        for nm, mod in (("qkv", qkv), ("gate_up", gu), ("o", o), ("down", dn)):
            want = g[f"tp/{world}/{rank}/{nm}"]
            assert np.array_equal(mod.weight.data.numpy().astype(np.int32), want), (nm, rank)
        assert (qkv.tp_dim, gu.tp_dim, o.tp_dim, dn.tp_dim) == (0, 0, 1, 1)
    finally:
        dist.destroy_process_group()


def _worker_forward(rank, world, port):
    _setup(rank, world, port)
    try:
        from nanovllm.layers import linear as L
        from nanovllm.utils import quantization as NQ
        _install_oracle_doubles()
        torch.manual_seed(0)                      # same full tensors on every rank
        hidden, inter, M = 64, 128, 5
        w_up = (torch.randn(2 * inter, hidden) * 0.1).to(torch.bfloat16)
        w_dn = (torch.randn(hidden, inter) * 0.1).to(torch.bfloat16)
        b_dn = torch.randn(hidden).to(torch.bfloat16)
        x = torch.randn(M, hidden).to(torch.bfloat16)
        torch.set_default_dtype(torch.bfloat16)
        col = L.MergedColumnParallelLinear(hidden, [inter, inter])
        col.weight_loader(col.weight, w_up[:inter], 0)
        col.weight_loader(col.weight, w_up[inter:], 1)
        row = L.RowParallelLinear(inter, hidden, bias=True)
        row.weight_loader(row.weight, w_dn)
        row.bias.data.copy_(b_dn)
        torch.set_default_dtype(torch.float32)

        # --- bf16 reference semantics (linear.py:72-73, 149-153)
        y_col = col(x)                                          # [M, 2*inter/tp]
        parts = [torch.empty_like(y_col) for _ in range(world)]
        dist.all_gather(parts, y_col)
        gate = torch.cat([p[:, : inter // world] for p in parts], 1)
        up = torch.cat([p[:, inter // world:] for p in parts], 1)
        full = torch.nn.functional.linear(x, w_up)
        assert torch.allclose(torch.cat([gate, up], 1).float(), full.float(), atol=1e-2, rtol=2e-2)
        h = torch.randn(M, inter, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16)
        h_shard = h[:, rank * inter // world:(rank + 1) * inter // world].contiguous()
        y_row = row(h_shard)
        ref = torch.nn.functional.linear(h, w_dn, b_dn)
        assert torch.allclose(y_row.float(), ref.float(), atol=3e-2, rtol=3e-2)

        # --- quantized modules converted from the sharded linears
        qcol = NQ.QLinear.from_linear(col, torch.int8, weight_only=True)
        qrow = NQ.QLinear.from_linear(row, torch.int8, weight_only=True)
        assert (qcol.tp_dim, qcol.tp_size, qrow.tp_dim, qrow.tp_rank) == (0, world, 1, rank)
        assert qcol.qw.shape == (2 * inter // world, hidden) and qrow.qw.shape == (hidden, inter // world)
        # column-parallel: per-row scales shard with the rows, so the shard's quantization equals the
        # corresponding rows of the unsharded quantization -> concatenated outputs are IDENTICAL
        q_full, s_full = NQ.per_row_quant(w_up, torch.int8)
        rows = torch.cat([torch.arange(rank * inter // world, (rank + 1) * inter // world),
                          inter + torch.arange(rank * inter // world, (rank + 1) * inter // world)])
        assert torch.equal(qcol.qw, q_full[rows]) and torch.equal(qcol.sw, s_full[rows])
        yq = qcol(x)
        assert torch.equal(yq, NQ.per_row_weight_only_matmul(x, q_full, s_full, None, torch.int8)[:, rows])
        # row-parallel: every rank quantizes its K-shard (scale over the local part of the row), adds
        # the bias on rank 0 only, and the partial outputs are summed: same on every rank, and close
        # to the unsharded quantized result (different scales -> quantization-level difference only)
        yr = qrow(h_shard)
        gathered = [torch.empty_like(yr) for _ in range(world)]
        dist.all_gather(gathered, yr)
        assert all(torch.equal(gathered[0], t) for t in gathered)
        q1, s1 = NQ.per_row_quant(w_dn, torch.int8)
        y1 = NQ.per_row_weight_only_matmul(h, q1, s1, b_dn, torch.int8)
        err = (yr.float() - y1.float()).abs().max() / y1.float().abs().max()
        assert err < 0.03, float(err)
        assert (yr.float() - ref.float()).abs().max() / ref.float().abs().max() < 0.05
        # a prefill-sized input goes through in row slabs, each reduced as soon as it exists: same result
        # (slab limits lowered so that 5 rows make three slabs), for the quantized and the bf16 module
        L.TP_SLAB_ROWS, L.TP_SLAB_MIN_ROWS = 2, 4
        assert L.tp_slab_bounds(5) == [(0, 2), (2, 4), (4, 5)] and L.tp_slab_bounds(3) is None
        assert torch.equal(qrow(h_shard), yr)
        assert torch.equal(qrow(h_shard.view(1, M, -1)), yr.view(1, M, -1))
        assert torch.equal(row(h_shard), y_row)
        L.TP_SLAB_ROWS, L.TP_SLAB_MIN_ROWS = 2048, 4096
        # bias really is applied once: removing it changes the result by exactly one bias (to rounding)
        qrow.bias = None
        yr_nb = qrow(h_shard)
        assert torch.allclose((yr.float() - yr_nb.float()), b_dn.float().expand(M, -1), atol=0.06)
    finally:
        dist.destroy_process_group()


def _spawn(fn, world, *args):
    port = 29650 + (os.getpid() % 200)
    mp.spawn(fn, args=(world, port) + args, nprocs=world, join=True)


@pytest.mark.parametrize("world", [2, 4])
def test_weight_loader_sharding_matches_reference(world):
    _spawn(_worker_layout, world, os.path.join(ROOT, "tests", "golden", "quant_golden.npz"))


def test_tp2_forward_bf16_and_quantized():
    _spawn(_worker_forward, 2)


def _worker_checkpoint_shards(rank, world, port):
    """Shards of a pre-quantized checkpoint: cut by the modules' own weight loaders, they must equal
    explicit slices of the unsharded tensors, and the sharded W8A16 forward (oracle test double) must
    reproduce the unsharded one exactly for column-parallel and up to the bf16 all-reduce for row-parallel."""
    _setup(rank, world, port)
    try:
        from oracle import qlinear_oracle as O
        from nanovllm.layers import linear as L
        from nanovllm.utils import checkpoint as C, quantization as NQ
        _install_oracle_doubles()
        g = torch.Generator().manual_seed(3)
        hidden, head, nh, nkv, inter, G = 256, 8, 8, 4, 128, 32

        def rq(n, k):
            return torch.randint(-127, 128, (n, k), dtype=torch.int8, generator=g)

        def rs(*shape):
            return torch.rand(*shape, generator=g) * 1e-2 + 1e-3
        qkv = L.QKVParallelLinear(hidden, head, nh, nkv)
        gu = L.MergedColumnParallelLinear(hidden, [inter, inter])
        dn = L.RowParallelLinear(inter, hidden)
        # --- per row
        Nq, Nk = nh * head, nkv * head
        q_full, s_full = rq(Nq + 2 * Nk, hidden), rs(Nq + 2 * Nk)
        q, s, _ = C.shard_quantized(qkv, q_full, s_full, None, "per_row")
        rows = torch.cat([torch.arange(Nq // world) + rank * Nq // world,
                          Nq + torch.arange(Nk // world) + rank * Nk // world,
                          Nq + Nk + torch.arange(Nk // world) + rank * Nk // world])
        assert torch.equal(q, q_full[rows]) and torch.equal(s, s_full[rows])
        q_full, s_full = rq(2 * inter, hidden), rs(2 * inter)
        q, s, _ = C.shard_quantized(gu, q_full, s_full, None, "per_row")
        rows = torch.cat([torch.arange(inter // world) + rank * inter // world,
                          inter + torch.arange(inter // world) + rank * inter // world])
        assert torch.equal(q, q_full[rows]) and torch.equal(s, s_full[rows])
        q_full, s_full, b_full = rq(hidden, inter), rs(hidden), torch.randn(hidden, generator=g).to(torch.bfloat16)
        q, s, b = C.shard_quantized(dn, q_full, s_full, b_full, "per_row")
        k = inter // world
        assert torch.equal(q, q_full[:, rank * k:(rank + 1) * k]) and torch.equal(s, s_full) and torch.equal(b, b_full)
        # the sharded row-parallel module multiplies the SAME quantized weights as the unsharded one
        x = (torch.randn(5, inter, generator=g) * 0.5).to(torch.bfloat16)
        mod = NQ.QLinear.from_quantized(dn, torch.int8, q, s, b, weight_only=True)
        y = mod(x[:, rank * k:(rank + 1) * k].contiguous())
        ref = NQ.per_row_weight_only_matmul(x, q_full, s_full, b_full, torch.int8)
        # (each rank's partial is rounded to bf16 and the all-reduce adds in bf16)
        assert ((y.float() - ref.float()).abs() <= 2.0 ** -6 * ref.float().abs() + 4e-2).all()
        # --- per group: a row-parallel shard takes its K/g columns of the scales
        sg_full = rs(hidden, inter // G)
        q, s, _ = C.shard_quantized(dn, q_full, sg_full, None, "per_group")
        assert torch.equal(s, sg_full[:, rank * k // G:(rank + 1) * k // G]) and q.shape == (hidden, k)
        sg_full = rs(2 * inter, hidden // G)
        q, s, _ = C.shard_quantized(gu, rq(2 * inter, hidden), sg_full, None, "per_group")
        assert torch.equal(s, sg_full[rows])
        # --- per tensor: the scalar travels unchanged
        q, s, _ = C.shard_quantized(dn, q_full, torch.tensor([0.01]), None, "per_tensor")
        assert s.numel() == 1 and q.shape == (hidden, k)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 4])
def test_quantized_checkpoint_sharding(world):
    _spawn(_worker_checkpoint_shards, world)


def test_gate_up_interleave_round_trip():
    from nanovllm.utils import checkpoint as C, quantization as NQ
    t = torch.arange(2 * 96 * 3).reshape(2 * 96, 3)
    il = NQ.interleave_gate_up(t)
    assert torch.equal(il[:32], t[:32]) and torch.equal(il[32:64], t[96:128]) and torch.equal(il[64:96], t[32:64])
    assert torch.equal(C.deinterleave_gate_up(il), t)
    v = torch.arange(2 * 64)
    assert torch.equal(C.deinterleave_gate_up(NQ.interleave_gate_up(v)), v)
