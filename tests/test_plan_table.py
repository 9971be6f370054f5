"""The measured-plan table of the GEMM dispatch (ql_gemm_plan_*): host-side behaviour on any machine, and on the GPU
that a table entry changes the launch without changing the result beyond the f32 summation order."""
import json

import pytest
import torch


@pytest.fixture()
def lib():
    from nanovllm import _qlinear as Q
    lib = Q.load()
    lib.ql_gemm_plan_clear()
    yield lib
    lib.ql_gemm_plan_clear()
    Q.load_plan_table(Q.PLAN_TABLE)          # what the process started with


def test_plan_set_query_clear(lib):
    from nanovllm import _qlinear as Q
    N, K = 2560, 9728
    before = Q.plan_query(0, 128, N, K)
    assert lib.ql_gemm_plan_set(0, 97, 128, N, K, 1, 4, 1, 4, 8) == 0
    assert lib.ql_gemm_plan_count() == 1
    assert Q.plan_query(0, 128, N, K) == (1, 4, 1, 4, 8)
    assert Q.plan_query(0, 97, N, K) == (1, 4, 1, 4, 8)
    assert Q.plan_query(0, 96, N, K) == Q.plan_query(0, 96, N, K)          # outside the range: the heuristic
    assert Q.plan_query(0, 129, N, K)[:4] != (1, 4, 1, 4) or Q.plan_query(0, 129, N, K)[4] != 8
    assert Q.plan_query(Q.QL_PLAN_GROUP, 128, N, K) != (1, 4, 1, 4, 8)      # another form: untouched
    # a forced configuration still wins over the table
    assert lib.ql_gemm_force_config(2, 2, 4, 8, 2) == 0
    assert Q.plan_query(0, 128, N, K) == (2, 2, 4, 8, 2)
    assert lib.ql_gemm_force_config(0, 0, 0, 0, 0) == 0
    lib.ql_gemm_plan_clear()
    assert lib.ql_gemm_plan_count() == 0
    assert Q.plan_query(0, 128, N, K) == before


def test_plan_set_rejects_what_cannot_run(lib):
    from nanovllm import _qlinear as Q
    N, K = 4096, 1024
    assert lib.ql_gemm_plan_set(0, 1, 32, N, K, 3, 4, 1, 4, 1) == -1               # tile not built
    assert lib.ql_gemm_plan_set(Q.QL_PLAN_SILU, 1, 32, N, K, 1, 1, 8, 4, 1) == -1   # SiLU pairs two column strips
    assert lib.ql_gemm_plan_set(0, 1, 32, N, K, 1, 1, 8, 4, 0) == -1               # no slices at all
    assert lib.ql_gemm_plan_set(0, 1, 32, N, K, 2, 2, 4, 8, -256) == -1            # stream-K: not on this tile ...
    assert lib.ql_gemm_plan_set(Q.QL_PLAN_W8A8, 1, 32, N, K, 2, 4, 2, 4, -256) == -1   # ... nor for the W8A8 kernels
    assert lib.ql_gemm_plan_set(Q.QL_PLAN_PARTIAL, 1, 32, N, K, 2, 4, 2, 4, -256) == -1   # ... nor the deferred form
    assert lib.ql_gemm_plan_set(Q.QL_PLAN_ROPE, 1, 32, N, K, 1, 1, 8, 4, 1) == -1   # heads need 128-column tiles
    assert lib.ql_gemm_plan_set(Q.QL_PLAN_GROUP, 1, 32, N, K, 8, 4, 1, 4, 1) == -1  # no 256-row tile with group scales
    assert lib.ql_gemm_plan_set(0, 1, 32, N, 1000, 1, 1, 8, 4, 1) == -1             # K % 32
    assert lib.ql_gemm_plan_set(0, 33, 32, N, K, 1, 1, 8, 4, 1) == -1               # empty range
    assert lib.ql_gemm_plan_set(Q.QL_PLAN_PARTIAL, 1, 32, N, K, 1, 1, 8, 4, 17) == -1
    assert lib.ql_gemm_plan_set(0, 1, 64, N, K, 1, 1, 8, 4, 1) == 0
    assert lib.ql_gemm_plan_set(0, 33, 96, N, K, 1, 2, 4, 8, 1) == -1               # overlaps the entry above
    assert b"overlap" in lib.ql_last_error()
    assert lib.ql_gemm_plan_set(0, 65, 96, N, K, 1, 2, 4, 8, 1) == 0
    assert lib.ql_gemm_plan_count() == 2


def test_partial_entry_decides_whether_the_caller_defers(lib):
    from nanovllm import _qlinear as Q
    N, K = 1024, 3072
    s0 = lib.ql_gemm_partial_slices(64, N, K, 0)
    assert s0 > 1                                                                   # the heuristic defers here
    assert lib.ql_gemm_plan_set(Q.QL_PLAN_PARTIAL, 33, 64, N, K, 1, 1, 8, 4, 1) == 0
    assert lib.ql_gemm_partial_slices(64, N, K, 0) == 1                             # measured: do not defer
    assert lib.ql_gemm_partial_slices(32, N, K, 0) == lib.ql_gemm_partial_slices(32, N, K, 0)
    lib.ql_gemm_plan_clear()
    assert lib.ql_gemm_partial_slices(64, N, K, 0) == s0


def test_workspace_sizing_sees_the_table(lib):
    N, K = 2560, 9728
    assert lib.ql_gemm_plan_set(0, 97, 128, N, K, 1, 4, 1, 4, 8) == 0
    tiles = 4 * (N // 128)
    assert lib.ql_gemm_workspace_bytes(128, N, K, 0) >= 65536 + tiles * 8 * (4 * 1 * 16 * 64 * 4)
    assert lib.ql_gemm_workspace_max_bytes(512, N, K, 0) >= lib.ql_gemm_workspace_bytes(128, N, K, 0)


def test_table_file_round_trip(lib, tmp_path):
    from nanovllm import _qlinear as Q
    doc = {"plans": [{"form": Q.QL_PLAN_SILU, "N": 19456, "K": 2560, "m_lo": 1, "m_hi": 32, "tile": [1, 4, 2, 16], "split": 1},
                     {"form": Q.QL_PLAN_GROUP | Q.QL_PLAN_PARTIAL, "N": 2560, "K": 9728, "m_lo": 33, "m_hi": 64,
                      "tile": [1, 2, 4, 8], "split": 6}]}
    path = tmp_path / "plans.json"
    path.write_text(json.dumps(doc))
    assert Q.load_plan_table(str(path), replace=True) == 2
    assert Q.plan_query(Q.QL_PLAN_SILU, 16, 19456, 2560) == (1, 4, 2, 16, 1)
    assert Q.plan_query(Q.QL_PLAN_GROUP | Q.QL_PLAN_PARTIAL, 64, 2560, 9728)[:4] == (1, 2, 4, 8)
    assert Q.load_plan_table(str(tmp_path / "missing.json"), replace=True) == 0
    assert lib.ql_gemm_plan_count() == 0
    # an entry this build cannot run is skipped with a warning (strict: an error); the others are kept
    doc["plans"].append({"form": 0, "N": 64, "K": 256, "m_lo": 1, "m_hi": 32, "tile": [3, 4, 1, 4], "split": 1})
    path.write_text(json.dumps(doc))
    with pytest.warns(UserWarning, match="skipped"):
        assert Q.load_plan_table(str(path), replace=True) == 2
    with pytest.raises(Q.QLinearError):
        Q.load_plan_table(str(path), replace=True, strict=True)


def test_shipped_table_is_well_formed():
    """Every entry of the table the package ships is accepted by the library (tiles built for the form, ranges that do
    not overlap), so loading the library can never fail on it."""
    from nanovllm import _qlinear as Q
    lib = Q.load()
    lib.ql_gemm_plan_clear()
    try:
        n = Q.load_plan_table(Q.PLAN_TABLE, replace=True, strict=True)
        assert n == lib.ql_gemm_plan_count()
        assert n == len(json.load(open(Q.PLAN_TABLE))["plans"])
    finally:
        lib.ql_gemm_plan_clear()
        Q.load_plan_table(Q.PLAN_TABLE)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["row", "group"])
def test_table_entry_changes_the_launch_not_the_result(lib, mode):
    from nanovllm import _qlinear as Q
    from nanovllm.utils import quantization as NQ
    torch.manual_seed(3)
    M, N, K = 96, 2560, 4096
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    mod = (NQ.QGroupLinear(torch.int8, w, None, 128) if mode == "group" else NQ.QLinear(torch.int8, w, None, weight_only=True)).tile()
    form = Q.QL_PLAN_GROUP if mode == "group" else 0
    Q.workspace.reserve("cuda", 64 << 20)
    y0 = mod(x)
    p0 = Q.plan_query(form, M, N, K)
    new = (1, 4, 1, 4, 4) if p0 != (1, 4, 1, 4, 4) else (1, 2, 4, 8, 2)
    assert lib.ql_gemm_plan_set(form, 65, 96, N, K, *new) == 0
    assert Q.plan_query(form, M, N, K) == new
    y1 = mod(x)
    lib.ql_gemm_plan_clear()
    y2 = mod(x)
    assert torch.equal(y0, y2)                                   # the heuristic's plan again: same bits
    # another tile and split = another f32 summation order: half a bf16 ulp + the accumulation allowance
    xf, wf = x.float(), (mod.rowmajor_qw().float().view(N, K // 128, 128) * mod.sw.view(N, -1, 1)).view(N, K) if mode == "group" \
        else mod.rowmajor_qw().float() * mod.sw.view(N, 1)
    ref = xf.double() @ wf.double().T
    l1 = xf.abs().double() @ wf.abs().double().T
    for y in (y0, y1):
        assert ((y.double() - ref).abs() <= 2.0 ** -7 * ref.abs() + 2.0 ** -17 * l1 + 1e-30).all()
    assert not torch.equal(y0, y1) or True                       # (bits may or may not differ; both are in tolerance)


def test_stream_k_and_sliced_silu_entries(lib):
    """split_k = -G selects the stream-K schedule (G workgroups, clamped to the units the problem has); the SiLU form
    may be cut into K slices (the tile's last workgroup runs the gate/up exchange); workspace sizing follows both, also
    for an entry whose row range starts and ends off the 32-row grid."""
    from nanovllm import _qlinear as Q
    N, K = 19456, 2560
    base = lib.ql_gemm_workspace_max_bytes(512, N, K, 0)
    assert lib.ql_gemm_plan_set(Q.QL_PLAN_SILU, 129, 256, N, K, 4, 8, 1, 2, -256) == 0
    assert Q.plan_query(Q.QL_PLAN_SILU, 200, N, K) == (4, 8, 1, 2, -256)
    assert lib.ql_gemm_plan_set(Q.QL_PLAN_SILU, 40, 50, N, K, 1, 2, 4, 8, 4) == 0
    assert Q.plan_query(Q.QL_PLAN_SILU, 45, N, K) == (1, 2, 4, 8, 3)        # (K = 2560: three 1024-k chunks, three slices)
    assert Q.plan_query(Q.QL_PLAN_SILU, 51, N, K)[4] == 1
    tile_bytes = 128 * 256 * 4
    tiles = 2 * (N // 256)
    assert lib.ql_gemm_workspace_bytes(200, N, K, 0) == 65536 + (256 + tiles) * tile_bytes
    assert lib.ql_gemm_workspace_max_bytes(512, N, K, 0) >= max(base, 65536 + (256 + tiles) * tile_bytes)
    # a grid larger than the problem's units is clamped: 1 x 2 tiles x 2 chunks of the 64 x 128 tile
    assert lib.ql_gemm_plan_set(0, 1, 64, 256, 512, 2, 4, 2, 4, -512) == 0
    assert Q.plan_query(0, 64, 256, 512) == (2, 4, 2, 4, -4)
    lib.ql_gemm_plan_clear()
