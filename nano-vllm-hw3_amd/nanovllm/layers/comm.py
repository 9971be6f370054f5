"""One-shot all-reduce of the row-parallel partial outputs over peer-mapped buffers (ql_allreduce_add_rmsnorm,
include/qlinear.h): the decode-sized all-reduce of ``RowParallelLinear.forward`` (reference: layers/linear.py:149-153)
together with the residual add + RMSNorm that consumes it, in one launch, without the collective library.

Opt-in (``NANOVLLM_ONESHOT_AR=1``): the RCCL all-reduce stays the default and the reference-numerics path (RCCL sums
bf16 partials hop by hop in ring order; this kernel sums them once, in rank order, in f32).  One process per GPU; the
buffers' IPC handles travel once through the process group that already exists.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from nanovllm import _qlinear as Q



def enabled():
    return os.environ.get("NANOVLLM_ONESHOT_AR") == "1"


MAX_ROWS = int(os.environ.get("NANOVLLM_ONESHOT_AR_ROWS", "512"))     # larger messages stay with the library collective
SPIN_LIMIT = int(os.environ.get("NANOVLLM_ONESHOT_AR_SPIN", "0"))     # polls per wait (0: the library's default)


class OneShotAllReduce:
    """A communicator for [T <= max_rows, hidden] bf16 partial sums of one tensor-parallel group."""

    def __init__(self, hidden, max_rows=MAX_ROWS, device=None, group=None):
        assert dist.is_initialized(), "the handles are exchanged through torch.distributed"
        self.lib = Q.load()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.hidden, self.max_rows = int(hidden), int(max_rows)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        nbytes = self.lib.ql_comm_buffer_bytes(self.world, self.max_rows, self.hidden)
        if nbytes == 0:
            raise Q.QLinearError(f"one-shot all-reduce: unsupported shape (world={self.world}, hidden={hidden})")
        with torch.cuda.device(self.device):
            own = ctypes.c_void_p()
            Q.check(self.lib.ql_comm_alloc(nbytes, ctypes.byref(own)))
            handle = ctypes.create_string_buffer(64)
            Q.check(self.lib.ql_comm_export(own, handle))
            handles = [None] * self.world
            dist.all_gather_object(handles, bytes(handle.raw), group=group)
            self.own = own.value
            self.bufs = (ctypes.c_void_p * self.world)()
            self.peers = []
            for r, h in enumerate(handles):
                if r == self.rank:
                    self.bufs[r] = self.own
                    continue
                p = ctypes.c_void_p()
                Q.check(self.lib.ql_comm_import(ctypes.create_string_buffer(h, 64), ctypes.byref(p)))
                self.bufs[r] = p.value
                self.peers.append(p.value)
        dist.barrier(group=group)          # every rank has mapped every buffer before the first call

    def fits(self, x):
        return x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and 0 < x.shape[0] <= self.max_rows \
            and x.shape[1] == self.hidden

    def reduce_add_rmsnorm(self, partial, residual, weight, eps):
        """(normed, new_residual) = add_rmsnorm(all_reduce(partial), residual, weight, eps), one launch."""
        Q.require_cuda(partial, torch.bfloat16, "partial")
        T, H = partial.shape
        out = torch.empty_like(partial)
        res_out = torch.empty_like(partial) if residual is not None else None
        y = None if residual is not None else torch.empty_like(partial)     # (no residual: the sum itself is the new one)
        Q.check(self.lib.ql_allreduce_add_rmsnorm(
            self.rank, self.world, self.bufs, self.max_rows, self.hidden, partial.data_ptr(),
            residual.data_ptr() if residual is not None else None, weight.data_ptr(), eps,
            y.data_ptr() if y is not None else None, out.data_ptr(), res_out.data_ptr() if res_out is not None else None,
            T, SPIN_LIMIT, Q.stream_ptr(partial.device)))
        return out, (res_out if residual is not None else y)

    def reduce(self, partial):
        """all_reduce(partial) (sum, bf16 result of an f32 sum in rank order)."""
        Q.require_cuda(partial, torch.bfloat16, "partial")
        y = torch.empty_like(partial)
        Q.check(self.lib.ql_allreduce_add_rmsnorm(
            self.rank, self.world, self.bufs, self.max_rows, self.hidden, partial.data_ptr(), None, None, 0.0,
            y.data_ptr(), None, None, partial.shape[0], SPIN_LIMIT, Q.stream_ptr(partial.device)))
        return y

    def status(self):
        """(calls completed, waits that timed out) - synchronizes the device."""
        e, t = ctypes.c_uint(), ctypes.c_uint()
        Q.check(self.lib.ql_comm_status(self.own, ctypes.byref(e), ctypes.byref(t)))
        return e.value, t.value

    def close(self):
        for p in self.peers:
            self.lib.ql_comm_unimport(p)
        self.peers = []
        if self.own:
            self.lib.ql_comm_free(self.own)
            self.own = None


class TPPartial:
    """A row-parallel shard's partial output whose all-reduce is left to the residual add + RMSNorm launch."""

    def __init__(self, y, comm):
        self.tp_partial, self.comm, self.shape = y, comm, y.shape

    def finish(self, residual, weight, eps):
        return self.comm.reduce_add_rmsnorm(self.tp_partial, residual, weight, eps)


_comms = {}


def communicator(hidden, device):
    """The process-wide communicator for a hidden size (created on first use, outside graph capture), or None when the
    one-shot path is off or this process is not tensor-parallel."""
    if not enabled() or not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    key = (int(hidden), torch.device(device).index)
    if key not in _comms:
        if torch.cuda.is_current_stream_capturing():
            return None
        _comms[key] = OneShotAllReduce(hidden, device=device)
    return _comms[key]


def statuses():
    """{(hidden, device index): (calls completed, waits timed out)} of this process's communicators (synchronizes)."""
    return {k: c.status() for k, c in _comms.items()}
