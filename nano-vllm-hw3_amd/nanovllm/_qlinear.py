"""ctypes binding of libqlinear.so (the C ABI declared in include/qlinear.h).

There is no CPU fallback: if the shared library is missing, or an op is asked to run on a tensor
that is not a contiguous CUDA tensor of the expected dtype, this module raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QLINEAR_LIB") or os.path.join(_HERE, "_lib", "libqlinear.so")   # env override: A/B builds

QL_INT8, QL_FP8 = 0, 1
QL_TILED = 16
QL_PER_ROW, QL_PER_TENSOR, QL_PER_GROUP = 0, 1, 2
QL_PLAN_GROUP, QL_PLAN_SILU, QL_PLAN_ROPE, QL_PLAN_W8A8, QL_PLAN_PARTIAL = 1, 2, 4, 8, 16
PLAN_TABLE = os.environ.get("QLINEAR_PLANS") or os.path.join(_HERE, "_plans", "gfx950.json")

_c_vp, _c_i64, _c_int, _c_sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t

# name -> (restype, argtypes); must list every function of include/qlinear.h
SIGNATURES = {
    "ql_version": (_c_int, []),
    "ql_last_error": (ctypes.c_char_p, []),
    "ql_quant_per_row": (_c_int, [_c_int, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_vp]),
    "ql_quant_per_row_wide": (_c_int, [_c_int, _c_int, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_vp]),
    "ql_quant_per_tensor": (_c_int, [_c_int, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp]),
    "ql_quant_per_group": (_c_int, [_c_int, _c_vp, _c_i64, _c_i64, _c_int, _c_vp, _c_vp, _c_vp]),
    "ql_fake_quant": (_c_int, [_c_int, _c_int, _c_vp, _c_i64, _c_i64, _c_int, _c_vp, _c_vp, _c_vp]),
    "ql_tile_weight": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    "ql_gemm_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_int]),
    "ql_gemm_workspace_max_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, _c_int]),
    "ql_gemm_w8a16": (_c_int, [_c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                               _c_int, _c_vp, _c_sz, _c_vp]),
    "ql_gemm_w8a16_noalloc": (_c_int, [_c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                                       _c_int, _c_vp]),
    "ql_gemm_w8a8": (_c_int, [_c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                              _c_vp, _c_sz, _c_vp]),
    "ql_gemm_w8a16_silu": (_c_int, [_c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64,
                                    _c_int, _c_vp, _c_sz, _c_vp]),
    "ql_gemm_w8a8_silu": (_c_int, [_c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_sz,
                                   _c_vp]),
    "ql_gemm_partial_slices": (_c_int, [_c_i64, _c_i64, _c_i64, _c_int]),
    "ql_gemm_w8a16_partial": (_c_int, [_c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_vp, _c_int, _c_vp]),
    "ql_add_rmsnorm_partial": (_c_int, [_c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_vp, _c_vp,
                                        _c_i64, _c_i64, _c_vp]),
    "ql_gemm_w8a16_rope": (_c_int, [_c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_int, _c_vp, _c_vp,
                                    _c_vp, _c_vp, ctypes.c_float, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int,
                                    _c_int, _c_vp, _c_sz, _c_vp]),
    "ql_gemm_w8a8_rope": (_c_int, [_c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_vp,
                                   _c_vp, _c_vp, ctypes.c_float, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_int, _c_int,
                                   _c_int, _c_vp, _c_sz, _c_vp]),
    "ql_gemm_force_config": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int]),
    "ql_gemm_plan_set": (_c_int, [_c_int, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "ql_gemm_plan_clear": (None, []),
    "ql_gemm_plan_count": (_c_int, []),
    "ql_gemm_plan_query": (_c_int, [_c_int, _c_i64, _c_i64, _c_i64, ctypes.POINTER(_c_int)]),
    "ql_add_rmsnorm": (_c_int, [_c_vp, _c_vp, _c_vp, ctypes.c_float, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    "ql_add_rmsnorm_quant": (_c_int, [_c_int, _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    "ql_silu_mul": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    "ql_qknorm_rope_store": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_vp, _c_vp, _c_vp, _c_vp,
                                      _c_vp, _c_vp, _c_i64, _c_int, _c_int, _c_int, _c_vp]),
    "ql_argmax_rows": (_c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    "ql_row_absmax_parts": (_c_int, [_c_vp, _c_i64, _c_i64, _c_int, _c_vp, _c_vp]),
    "ql_gemm_w8a8_partial": (_c_int, [_c_int, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_int, _c_vp]),
    "ql_add_rmsnorm_partial_w8a8": (_c_int, [_c_int, _c_vp, _c_int, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_float,
                                             _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_vp]),
    "ql_gemm_w8a8_silu_rowmax": (_c_int, [_c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp,
                                          _c_sz, _c_vp]),
    "ql_gemm_w8a8_dynq": (_c_int, [_c_int, _c_vp, _c_vp, _c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp,
                                   _c_sz, _c_vp]),
    "ql_comm_buffer_bytes": (_c_sz, [_c_int, _c_i64, _c_i64]),
    "ql_comm_alloc": (_c_int, [_c_sz, ctypes.POINTER(_c_vp)]),
    "ql_comm_free": (_c_int, [_c_vp]),
    "ql_comm_export": (_c_int, [_c_vp, _c_vp]),
    "ql_comm_import": (_c_int, [_c_vp, ctypes.POINTER(_c_vp)]),
    "ql_comm_unimport": (_c_int, [_c_vp]),
    "ql_comm_status": (_c_int, [_c_vp, ctypes.POINTER(ctypes.c_uint), ctypes.POINTER(ctypes.c_uint)]),
    "ql_allreduce_add_rmsnorm": (_c_int, [_c_int, _c_int, ctypes.POINTER(_c_vp), _c_i64, _c_i64, _c_vp, _c_vp, _c_vp,
                                          ctypes.c_float, _c_vp, _c_vp, _c_vp, _c_i64, ctypes.c_uint, _c_vp]),
    "ql_lmhead_logits": (_c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp]),
    "ql_lmhead_workspace_bytes": (_c_sz, [_c_i64, _c_i64]),
    "ql_lmhead_argmax": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_sz, _c_vp]),
    "ql_attn_prefill": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_int, _c_int, _c_int, _c_int,
                                 _c_int, _c_vp]),
    "ql_paged_attn_workspace_bytes": (_c_sz, [_c_i64, _c_int, _c_int, _c_int, _c_int]),
    "ql_paged_attn_decode": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_i64, _c_int, _c_int,
                                      _c_int, _c_int, _c_int, _c_vp, _c_sz, _c_vp]),
    "ql_paged_attn_decode_quant": (_c_int, [_c_int, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_float,
                                            _c_i64, _c_int, _c_int, _c_int, _c_int, _c_int, _c_vp, _c_sz, _c_vp]),
    "ql_paged_attn_decode_rowmax": (_c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_float, _c_i64, _c_int,
                                             _c_int, _c_int, _c_int, _c_int, _c_vp, _c_sz, _c_vp]),
}

_lib = None


class QLinearError(RuntimeError):
    pass


def load():
    """Load libqlinear.so and type its entry points.  Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QLinearError(
            f"{LIB_PATH} not found: build it with `python nano-vllm-hw3_amd/build.py` "
            "(or __graft_entry__.build()); there is no CPU fallback for the quantized-linear path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    if not os.environ.get("QLINEAR_NO_PLANS"):
        load_plan_table(PLAN_TABLE)
    return lib


def load_plan_table(path, replace=False, strict=False):
    """Register the measured GEMM plans of a JSON file written by tools/autotune_plans.py:
    {"plans": [{"form", "N", "K", "m_lo", "m_hi", "tile": [mt, wn, wk, cp], "split"}, ...]}.  Returns the number
    of entries set; a missing file is no error (the built-in heuristics then apply to every shape).  An entry the
    library rejects (a tile this build does not have, an overlap) is skipped with a warning - a plan is an
    optimisation, never a reason not to start - unless ``strict``."""
    import json
    import warnings
    lib = load()
    if replace:
        lib.ql_gemm_plan_clear()
    if not os.path.exists(path):
        return 0
    with open(path) as f:
        doc = json.load(f)
    n = 0
    for e in doc.get("plans", []):
        mt, wn, wk, cp = e["tile"]
        rc = lib.ql_gemm_plan_set(int(e["form"]), int(e["m_lo"]), int(e["m_hi"]), int(e["N"]), int(e["K"]), mt, wn, wk, cp,
                                  int(e["split"]))
        if rc != 0:
            if strict:
                check(rc)
            warnings.warn(f"{path}: plan entry skipped ({lib.ql_last_error().decode()})")
            continue
        n += 1
    return n


def plan_query(form, M, N, K):
    """(mt, wn, wk, cp, split) of the plan a GEMM call of this form and shape would run."""
    out = (_c_int * 5)()
    check(load().ql_gemm_plan_query(form, M, N, K, out))
    return tuple(out)


def check(rc):
    if rc != 0:
        raise QLinearError(f"libqlinear error {rc}: {load().ql_last_error().decode()}")


def dtype_code(dtype):
    if dtype == torch.int8:
        return QL_INT8
    if dtype == torch.float8_e4m3fn:
        return QL_FP8
    raise AssertionError(f"quantized dtype must be torch.int8 or torch.float8_e4m3fn, got {dtype}")


def stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def require_cuda(t, dtype, name):
    if not t.is_cuda:
        raise QLinearError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise QLinearError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise QLinearError(f"{name} must be contiguous")
    return t


class _Workspace:
    """One growable scratch buffer per device for split-K slabs.  It is sized outside graph capture
    (reserve()); during capture a too-small workspace is an error instead of a hidden allocation.
    Captured hipGraphs hold the buffer by ADDRESS: once pin() has been called for a device, a buffer that is
    outgrown is kept alive for the rest of the process (the graphs keep using it, correctly: it fits every
    shape they captured) and only later eager calls move to the larger one."""

    def __init__(self):
        self.buf = {}
        self.pinned = {}      # device -> buffers captured graphs may still address

    @staticmethod
    def _key(device):
        device = torch.device(device)
        return device.index if device.index is not None else torch.cuda.current_device()

    def pin(self, device):
        key = self._key(device)
        cur = self.buf.get(key)
        if cur is not None and not any(cur is b for b in self.pinned.setdefault(key, [])):
            self.pinned[key].append(cur)

    def reserve(self, device, nbytes):
        device = torch.device(device)
        key = self._key(device)
        cur = self.buf.get(key)
        if cur is None or cur.numel() < nbytes:
            if torch.cuda.is_current_stream_capturing():
                raise QLinearError(
                    f"split-K workspace of {nbytes} bytes requested during graph capture; call "
                    "reserve_workspace() for the largest batch before capturing")
            # zero-filled: the head of the workspace holds the split-K ticket counters, which the
            # kernels leave at zero after every call
            self.buf[key] = torch.zeros(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        return self.buf[key]

    def get(self, device, nbytes):
        if nbytes == 0:
            return None, 0
        b = self.reserve(device, nbytes)
        return b.data_ptr(), b.numel()


workspace = _Workspace()        # split-K slabs of the GEMMs
attn_workspace = _Workspace()   # per-KV-block partial states of decode attention
head_workspace = _Workspace()   # per-column-tile row maxima of the output projection's sampler epilogue
