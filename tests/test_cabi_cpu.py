"""CPU-side checks of the C ABI: the built library loads, exports every function the header
declares (no compute is called here), and the Python binding types all of them."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "qlinear.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ql_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    names = declared_functions()
    for must in ("ql_quant_per_row", "ql_quant_per_tensor", "ql_quant_per_group", "ql_fake_quant",
                 "ql_gemm_w8a16", "ql_gemm_w8a8", "ql_gemm_workspace_bytes", "ql_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from nanovllm import _qlinear as Q
    assert os.path.exists(Q.LIB_PATH), "build libqlinear.so first: python nano-vllm-hw3_amd/build.py"
    lib = ctypes.CDLL(Q.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in qlinear.h but not exported"


def test_binding_covers_header():
    from nanovllm import _qlinear as Q
    assert sorted(Q.SIGNATURES) == declared_functions()
    lib = Q.load()
    assert lib.ql_version() >= 1


def test_argument_validation_needs_no_gpu():
    """Rejections happen before any HIP call, so they can be exercised on a CPU-only host."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    one = ctypes.c_void_p(16)   # never dereferenced: the calls below are rejected first
    assert lib.ql_gemm_w8a16(7, 0, one, one, one, None, one, 4, 64, 128, 0, None, 0, None) == -1
    assert b"bad argument" in lib.ql_last_error()
    assert lib.ql_gemm_w8a16(0, 0, one, one, one, None, one, 4, 64, 100, 0, None, 0, None) == -2
    assert b"multiple of 32" in lib.ql_last_error()
    assert lib.ql_gemm_w8a16(0, 2, one, one, one, None, one, 4, 64, 256, 96, None, 0, None) == -2    # K % group
    assert lib.ql_gemm_w8a16(0, 2, one, one, one, None, one, 4, 64, 240, 48, None, 0, None) == -2    # K % 32
    # (a group size that is not a multiple of 32 is valid for the plain GEMM; its fused forms do not exist)
    assert lib.ql_gemm_w8a16_silu(0, 2, one, one, one, None, one, 4, 64, 192, 48, None, 0, None) == -2
    assert b"multiple of 32" in lib.ql_last_error()
    assert lib.ql_gemm_w8a16(0, 2, one, one, one, None, one, 4, 64, 192, 0, None, 0, None) == -2     # group <= 0
    assert lib.ql_quant_per_group(0, one, 4, 256, 96, one, one, None) == -2
    assert lib.ql_quant_per_row(0, one, 4, 100, one, one, None) == -2
    assert lib.ql_gemm_w8a8(0, one, one, one, None, None, one, 4, 64, 128, None, 0, None) == -1
    assert lib.ql_gemm_force_config(3, 4, 1, 4, 0) == -1
    assert lib.ql_gemm_force_config(0, 0, 0, 0, 0) == 0
    # workspace sizing is pure host arithmetic: 64 KiB of counters + split x tile slabs, or nothing
    assert lib.ql_gemm_force_config(1, 1, 4, 2, 4) == 0
    assert lib.ql_gemm_workspace_bytes(16, 64, 2048, 0) == 65536 + 2 * 4 * (1 * 1 * 16 * 64 * 4)
    assert lib.ql_gemm_force_config(0, 0, 0, 0, 0) == 0
    assert lib.ql_gemm_workspace_bytes(4096, 4096, 1024, 0) == 0


def test_workspace_max_covers_every_row_count():
    """ql_gemm_workspace_max_bytes(max_M) is the maximum of ql_gemm_workspace_bytes over 1..max_M (brute force), so
    one reservation covers decode buckets and eager prefill batches alike; includes the Qwen3-4B down_proj shape
    whose M = 600 call needs more than any M <= 512."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    for (N, K) in [(1024, 3072), (2560, 9728), (2048, 6144), (4096, 1024), (64, 256)]:
        for gran in (0, 2):
            brute = max(lib.ql_gemm_workspace_bytes(m, N, K, gran) for m in range(1, 2101))
            assert lib.ql_gemm_workspace_max_bytes(2100, N, K, gran) == brute
    assert lib.ql_gemm_workspace_max_bytes(16384, 2560, 9728, 0) >= lib.ql_gemm_workspace_bytes(600, 2560, 9728, 0)
    # (the built-in heuristics alone: measured plans of the shipped table may split the decode sizes further)
    lib.ql_gemm_plan_clear()
    try:
        assert lib.ql_gemm_workspace_max_bytes(16384, 2560, 9728, 0) >= lib.ql_gemm_workspace_bytes(600, 2560, 9728, 0) > \
            lib.ql_gemm_workspace_max_bytes(512, 2560, 9728, 0)
    finally:
        Q.load_plan_table(Q.PLAN_TABLE)
    assert lib.ql_gemm_workspace_max_bytes(0, 64, 256, 0) == 0


def test_ops_refuse_cpu_tensors():
    """No silent CPU fallback: the operator layer raises on non-GPU tensors."""
    import torch
    from nanovllm import _qlinear as Q
    from nanovllm.utils import quantization as NQ
    x = torch.zeros(4, 128, dtype=torch.bfloat16)
    with pytest.raises(Q.QLinearError):
        NQ.per_row_quant(x, torch.int8)
    with pytest.raises(Q.QLinearError):
        NQ.per_group_quant(x, 64, torch.int8)
