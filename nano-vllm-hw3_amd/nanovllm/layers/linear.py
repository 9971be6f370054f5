"""bf16 linear layers with Megatron-style tensor parallelism: the layout contract the quantized
modules are converted from.  Mirrors the reference's ``nanovllm/layers/linear.py`` (class names,
constructor arguments, ``weight_loader`` shard ids), written for one process per GPU on RCCL.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn


def _tp_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def divide(a: int, b: int) -> int:
    assert a % b == 0, f"{a} is not divisible by {b}"
    return a // b


class _SideStreamAllReduce:
    """Row-parallel partial sums are reduced with RCCL on a dedicated HIP stream.  A decode-sized output goes in
    one piece (fork, all-reduce, join: nothing else can run before the sum exists).  A prefill-sized output is
    produced in row slabs (reduce_slabs): slab i is all-reduced on the side stream while the compute stream
    already runs the GEMM of slab i+1, so only the last slab's collective is exposed.  Both forms only record
    events and enqueue work, so they can be captured into a hipGraph."""

    def __init__(self):
        self.stream = None

    def side(self, device):
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=device)
        return self.stream

    @staticmethod
    def _host_reduce(y):
        # test transport (several ranks sharing one GPU): gloo moves host memory
        h = y.cpu()
        dist.all_reduce(h)
        y.copy_(h)

    def reduce_(self, y: torch.Tensor):
        if not y.is_cuda:
            dist.all_reduce(y)
            return y
        if dist.get_backend() == "gloo":
            self._host_reduce(y)
            return y
        main = torch.cuda.current_stream(y.device)
        side = self.side(y.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dist.all_reduce(y)
        main.wait_stream(side)
        return y

    def reduce_slabs(self, y: torch.Tensor, bounds, produce):
        """``produce(a, b)`` enqueues the computation of rows [a, b) of ``y`` on the current stream; every slab
        is summed over the TP group as soon as it exists, overlapping the production of the next one."""
        if not y.is_cuda or dist.get_backend() == "gloo":
            for a, b in bounds:
                produce(a, b)
                if y.is_cuda:
                    self._host_reduce(y[a:b])
                else:
                    dist.all_reduce(y[a:b])
            return y
        main = torch.cuda.current_stream(y.device)
        side = self.side(y.device)
        for a, b in bounds:
            produce(a, b)
            ready = torch.cuda.Event()
            ready.record(main)
            side.wait_event(ready)                     # the side stream waits for THIS slab only
            with torch.cuda.stream(side):
                dist.all_reduce(y[a:b])
        main.wait_stream(side)
        return y


_allreduce = _SideStreamAllReduce()

# rows per slab of a pipelined row-parallel output, and the row count from which slabs are used
TP_SLAB_ROWS = 2048
TP_SLAB_MIN_ROWS = 4096


def tp_all_reduce(y: torch.Tensor) -> torch.Tensor:
    """Sum a row-parallel partial output over the TP group (reference: linear.py:151-152)."""
    return _allreduce.reduce_(y)


def tp_slab_bounds(m: int):
    """Row ranges of a pipelined row-parallel output, or None when the output goes in one piece."""
    if m < TP_SLAB_MIN_ROWS:
        return None
    return [(a, min(a + TP_SLAB_ROWS, m)) for a in range(0, m, TP_SLAB_ROWS)]


def tp_all_reduce_slabs(y: torch.Tensor, bounds, produce) -> torch.Tensor:
    return _allreduce.reduce_slabs(y, bounds, produce)


class LinearBase(nn.Module):
    """weight[out, in] (+ bias[out]); records which dim is sharded (reference: linear.py:12-34)."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False, tp_dim=None):
        super().__init__()
        self.tp_dim = tp_dim
        self.tp_rank, self.tp_size = _tp_world()
        self.weight = nn.Parameter(torch.empty(output_size, input_size), requires_grad=False)
        self.weight.weight_loader = self.weight_loader
        if bias:
            self.bias = nn.Parameter(torch.empty(output_size), requires_grad=False)
            self.bias.weight_loader = self.weight_loader
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        raise NotImplementedError


class ReplicatedLinear(LinearBase):
    def __init__(self, input_size, output_size, bias=False):
        super().__init__(input_size, output_size, bias)

    def weight_loader(self, param, loaded_weight):
        param.data.copy_(loaded_weight)

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class ColumnParallelLinear(LinearBase):
    """Shards the output features (dim 0): rank r owns rows [r*N/tp, (r+1)*N/tp)."""

    def __init__(self, input_size, output_size, bias=False):
        _, tp = _tp_world()
        super().__init__(input_size, divide(output_size, tp), bias, 0)

    def weight_loader(self, param, loaded_weight):
        n = param.data.size(0)
        param.data.copy_(loaded_weight.narrow(0, self.tp_rank * n, n))

    def forward(self, x):
        return F.linear(x, self.weight, self.bias)


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel matrices stacked along dim 0 (gate, up); each is sharded on its own,
    so a rank's rows are [gate shard | up shard]."""

    def __init__(self, input_size, output_sizes, bias=False):
        self.output_sizes = list(output_sizes)
        super().__init__(input_size, sum(output_sizes), bias)

    def weight_loader(self, param, loaded_weight, loaded_shard_id: int):
        offset = sum(self.output_sizes[:loaded_shard_id]) // self.tp_size
        size = self.output_sizes[loaded_shard_id] // self.tp_size
        dst = param.data.narrow(0, offset, size)
        dst.copy_(loaded_weight.narrow(0, self.tp_rank * size, size))


class QKVParallelLinear(ColumnParallelLinear):
    """q, k, v projections packed as [q heads | k heads | v heads] of this rank."""

    def __init__(self, hidden_size, head_size, total_num_heads, total_num_kv_heads=None, bias=False):
        _, tp = _tp_world()
        total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.head_size = head_size
        self.num_heads = divide(total_num_heads, tp)
        self.num_kv_heads = divide(total_num_kv_heads, tp)
        super().__init__(hidden_size, (total_num_heads + 2 * total_num_kv_heads) * head_size, bias)

    def weight_loader(self, param, loaded_weight, loaded_shard_id: str):
        assert loaded_shard_id in ("q", "k", "v")
        q_rows = self.num_heads * self.head_size
        kv_rows = self.num_kv_heads * self.head_size
        offset, size = {"q": (0, q_rows), "k": (q_rows, kv_rows), "v": (q_rows + kv_rows, kv_rows)}[loaded_shard_id]
        dst = param.data.narrow(0, offset, size)
        dst.copy_(loaded_weight.narrow(0, self.tp_rank * size, size))


class RowParallelLinear(LinearBase):
    """Shards the input features (dim 1); every rank produces a full-width partial that is summed
    over the group, bias on rank 0 only."""

    def __init__(self, input_size, output_size, bias=False):
        _, tp = _tp_world()
        super().__init__(divide(input_size, tp), output_size, bias, 1)

    def weight_loader(self, param, loaded_weight):
        k = param.data.size(1)
        param.data.copy_(loaded_weight.narrow(1, self.tp_rank * k, k))

    def forward(self, x):
        bias = self.bias if self.tp_rank == 0 else None
        if self.tp_size > 1:
            x2 = x.reshape(-1, x.shape[-1])
            bounds = tp_slab_bounds(x2.shape[0])
            if bounds is not None:
                y = torch.empty((x2.shape[0], self.weight.shape[0]), dtype=x.dtype, device=x.device)

                def produce(a, b):
                    y[a:b] = F.linear(x2[a:b], self.weight, bias)
                return tp_all_reduce_slabs(y, bounds, produce).view(*x.shape[:-1], -1)
        y = F.linear(x, self.weight, bias)
        if self.tp_size > 1:
            y = tp_all_reduce(y)
        return y
