"""The one-shot all-reduce over peer-mapped buffers (ql_allreduce_add_rmsnorm) with TWO processes sharing the one GPU
(hipIpc handles work on one device): against the CPU oracle of RowParallelLinear.forward's all-reduce
(oracle.row_parallel_reduce, reference layers/linear.py:149-153) followed by the library's own residual add + RMSNorm,
bit for bit, eager and replayed from a hipGraph; identical bits on both ranks; and a lone rank whose peer never
arrives gives up after its spin bound (a timeout is counted, nothing hangs)."""
import os
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(1, 1024), (37, 1024), (256, 4096), (130, 2560), (5, 8192)]      # (rows, hidden): NV = 1, 2, 4


def _partial(rank, T, H, seed):
    g = torch.Generator().manual_seed(1000 * seed + rank)
    return (torch.randn(T, H, generator=g) * (1.0 + rank)).to(torch.bfloat16)


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "nano-vllm-hw3_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from nanovllm import ops
    from nanovllm.layers.comm import OneShotAllReduce
    res = {}
    with torch.inference_mode():
        for si, (T, H) in enumerate(SHAPES):
            comm = OneShotAllReduce(H, max_rows=256, device="cuda:0")
            x = _partial(rank, T, H, si).cuda()
            g = torch.Generator().manual_seed(77 + si)
            resid = torch.randn(T, H, generator=g).to(torch.bfloat16).cuda()
            w = (1 + 0.1 * torch.randn(H, generator=g)).to(torch.bfloat16).cuda()
            y = comm.reduce(x)
            out, new_res = comm.reduce_add_rmsnorm(x, resid, w, 1e-6)
            out0, res0 = comm.reduce_add_rmsnorm(x, None, w, 1e-6)
            # the same call captured into a hipGraph and replayed: the sequence number lives in the buffer
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            xs = x.clone()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    g_out, g_res = comm.reduce_add_rmsnorm(xs, resid, w, 1e-6)
            replays = []
            for it in range(3):
                xs.copy_(_partial(rank, T, H, 100 + it).cuda() if it else x)
                dist.barrier()
                graph.replay()
                torch.cuda.synchronize()
                replays.append((g_out.cpu().clone(), g_res.cpu().clone()))
            epoch, timeouts = comm.status()
            res[(T, H)] = dict(y=y.cpu(), out=out.cpu(), res=new_res.cpu(), out0=out0.cpu(), res0=res0.cpu(), replays=replays,
                               epoch=epoch, timeouts=timeouts)
            # what the fused launch must equal: the library's norm on the reduced tensor
            res[(T, H)]["norm_of_y"] = tuple(t.cpu() for t in ops.add_rmsnorm(y, resid, w, 1e-6))
            dist.barrier()
            comm.close()
    torch.save(res, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def _bits(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def test_one_shot_all_reduce_two_processes_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    from oracle import qlinear_oracle as O
    out_path = str(tmp_path / "ar")
    ctx = mp.spawn(_worker, args=(2, 29931 + os.getpid() % 50, out_path), nprocs=2, join=False)
    deadline = time.time() + 140
    done = False
    while time.time() < deadline and not done:
        done = ctx.join(timeout=5)          # (a failed worker raises here, with its traceback)
    if not done:
        for pr in ctx.processes:
            pr.kill()
        pytest.fail("the two ranks did not finish within 140 s")
    r0, r1 = (torch.load(f"{out_path}.{r}", weights_only=False) for r in (0, 1))
    for si, (T, H) in enumerate(SHAPES):
        a, b = r0[(T, H)], r1[(T, H)]
        assert a["timeouts"] == 0 and b["timeouts"] == 0
        assert a["epoch"] == b["epoch"] == 6                    # three eager calls, three replays
        parts = [_bits(_partial(r, T, H, si)) for r in (0, 1)]
        want = O.row_parallel_reduce(parts)                      # f32 sum in rank order, one rounding
        for r in (a, b):
            assert np.array_equal(_bits(r["y"]), want), (T, H, "all-reduce vs oracle")
            assert torch.equal(r["out"], r["norm_of_y"][0]) and torch.equal(r["res"], r["norm_of_y"][1]), (T, H, "fused norm")
            assert torch.equal(r["res0"], r["y"])                # no residual: the sum is the new residual
        for k in ("y", "out", "res", "out0"):
            assert torch.equal(a[k], b[k]), (T, H, k, "ranks differ")
        # replays: the first on the captured input (= the eager result), the others on new inputs, ranks identical
        assert torch.equal(a["replays"][0][0], a["out"]) and torch.equal(a["replays"][0][1], a["res"])
        for it in range(3):
            assert torch.equal(a["replays"][it][0], b["replays"][it][0]) and torch.equal(a["replays"][it][1], b["replays"][it][1])
        for it in (1, 2):
            parts = [_bits(_partial(r, T, H, 100 + it)) for r in (0, 1)]
            y = torch.from_numpy(O.row_parallel_reduce(parts).view(np.int16)).view(torch.bfloat16)
            g = torch.Generator().manual_seed(77 + si)
            resid = torch.randn(T, H, generator=g).to(torch.bfloat16)
            assert torch.equal(a["replays"][it][1], (y.float() + resid.float()).to(torch.bfloat16)), (T, H, it, "replayed residual")


def test_one_shot_all_reduce_gives_up_on_a_missing_peer():
    """world = 2 with the 'peer' buffer a second local allocation that nobody ever writes: the wait must end at its
    spin bound, count the timeouts and leave the device usable."""
    import ctypes
    from nanovllm import _qlinear as Q
    lib = Q.load()
    T, H = 4, 1024
    nbytes = lib.ql_comm_buffer_bytes(2, 16, H)
    bufs = (ctypes.c_void_p * 2)()
    for i in range(2):
        p = ctypes.c_void_p()
        Q.check(lib.ql_comm_alloc(nbytes, ctypes.byref(p)))
        bufs[i] = p.value
    x = torch.randn(T, H, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x)
    t0 = time.time()
    Q.check(lib.ql_allreduce_add_rmsnorm(0, 2, bufs, 16, H, x.data_ptr(), None, None, 0.0, y.data_ptr(), None, None, T,
                                         20000, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert time.time() - t0 < 30
    e, t = ctypes.c_uint(), ctypes.c_uint()
    Q.check(lib.ql_comm_status(bufs[0], ctypes.byref(e), ctypes.byref(t)))
    assert e.value == 1 and t.value == T                       # one wait per row timed out
    assert torch.equal(y, x)                                    # (the absent peer's slot is still all zeros)
    # argument checks: more rows than the buffers hold, a hidden size the kernel does not take
    assert lib.ql_allreduce_add_rmsnorm(0, 2, bufs, 16, H, x.data_ptr(), None, None, 0.0, y.data_ptr(), None, None, 17, 0, None) == -2
    assert lib.ql_allreduce_add_rmsnorm(0, 2, bufs, 16, 1028, x.data_ptr(), None, None, 0.0, y.data_ptr(), None, None, 1, 0, None) == -2
    assert torch.equal(torch.ones(8, device="cuda") * 2, torch.full((8,), 2.0, device="cuda"))
    for i in range(2):
        Q.check(lib.ql_comm_free(bufs[i]))
