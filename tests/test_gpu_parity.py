"""GPU parity tests (run on the MI355X box: ``pytest -m gpu``).

Every test drives the HIP path through the C ABI (ctypes, bilinear_amd._native)
and compares with
  * the golden vectors captured from the reference itself (tests/golden/*.npz), and
  * the NumPy oracle (oracle/numpy_oracle.py, fp64 arbiter) on the same seeded inputs.

Tolerance: BASELINE.json north_star states "within 1e-3 rel fp32".  The tests
use RTOL = 1e-3 on |a-b| <= RTOL*(|b| + rms(b)); the kernels actually land
around 1e-5 (printed by -s).
"""
import ctypes

import numpy as np
import pytest
import torch

from golden_util import FIXTURES, Golden, is_prebn_bias, safe_masks
from oracle import numpy_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-3          # north_star tolerance
TIGHT = 1e-4         # what fp32 kernels should comfortably meet


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _close(got, ref, rtol, what):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    rms = np.sqrt((ref ** 2).mean()) if ref.size else 0.0
    bound = rtol * (np.abs(ref) + rms)
    err = np.abs(got - ref)
    worst = np.unravel_index(np.argmax(err - bound), err.shape) if ref.size else ()
    assert (err <= bound).all(), "%s: |err|=%.3e > %.3e at %s (ref %.3e rms %.3e)" % (
        what, err[worst], bound[worst], worst, ref[worst], rms)
    return float((err / np.maximum(bound, 1e-300)).max()) * rtol if ref.size else 0.0


# ----------------------------------------------------------------------------
# kernel level: the fp32 MFMA GEMM in its four operand layouts
# ----------------------------------------------------------------------------
GEMM_CASES = [
    # M, N, K, a_kmajor, b_kmajor, splits
    (256, 256, 64, 0, 0, 1),          # forward layout, full tiles
    (300, 1024, 32, 0, 0, 1),         # encode forward: K = 32, ragged M
    (4096, 48, 1024, 0, 0, 1),        # decode forward: N = 48
    (200, 1024, 1024, 0, 1, 1),       # dgrad
    (64, 1024, 48, 0, 1, 1),          # decode dgrad: K = 48 (ragged K tile)
    (1024, 1024, 512, 1, 1, 4),       # wgrad, split over the batch
    (1024, 32, 1000, 1, 1, 8),        # encode wgrad: N = 32, ragged reduction
    (48, 1024, 777, 1, 1, 1),         # decode wgrad: M = 48, odd batch
    (132, 200, 36, 1, 0, 1),
    # split grids that take the slab-major XCD mapping (xcd_remap_split: tiles % 8 == 0, S | 8,
    # tiles_m % (8 / S) == 0); every (tile, slab) must be produced exactly once (NaN-initialised C)
    (1024, 1024, 1024, 1, 1, 2),
    (1024, 1024, 2048, 1, 1, 8),
    (2048, 2048, 256, 1, 1, 2),
    (128, 1024, 1024, 0, 0, 8),       # small-batch forward: one row of tiles, split-K
    (512, 512, 1024, 1, 1, 4),        # 16 tiles, 4 slabs: two bands of output rows
    (384, 1024, 512, 1, 1, 4),        # 24 tiles: tiles_m = 3 does not divide -> plain mapping
]


@pytest.mark.parametrize("entry", ["blh_gemm_f32", "blh_gemm_bf16x3"])
@pytest.mark.parametrize("M,N,K,ak,bk,splits", GEMM_CASES)
def test_gemm_f32_layouts(native, M, N, K, ak, bk, splits, entry):
    """Both fp32-accurate GEMMs (exact fp32 MFMA; three-way bf16 split on the bf16 MFMA) against
    the fp64 product, same tolerance."""
    dev = _dev()
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    # asymmetric, non-trivial operands (a transposed output or a swapped fragment map shows)
    A = rng.standard_normal((K, M) if ak else (M, K)).astype(np.float32)
    B = rng.standard_normal((K, N) if bk else (N, K)).astype(np.float32)
    A[0] += 3.0
    ref = (A.T if ak else A).astype(np.float64) @ (B if bk else B.T).astype(np.float64)
    a, b = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    c = torch.full((splits, M, N), float("nan"), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = getattr(native, entry)(st, a.data_ptr(), A.shape[1], ak, b.data_ptr(), B.shape[1], bk,
                                c.data_ptr(), N, M, N, K, splits, None, None, 0)
    assert rc == 0, native.blh_status_string(rc)
    if splits > 1:
        out = torch.empty(M, N, device=dev)
        rc = native.blh_sum_slabs(st, c.data_ptr(), M * N, splits, out.data_ptr())
        assert rc == 0
    else:
        out = c[0]
    torch.cuda.synchronize()
    _close(out.cpu().numpy(), ref, 2e-5, "gemm")


@pytest.mark.parametrize("M,N,K,ak,bk,splits", [(512, 1024, 1024, 0, 0, 1), (384, 1024, 1024, 0, 1, 1),
                                                (1024, 1024, 4096, 1, 1, 4)])
def test_bf16x3_gemm_error_is_fp32_level(native, M, N, K, ak, bk, splits):
    """gemm_dtype = 2 claims fp32 accuracy: its error against the fp64 product, measured on the
    scale sum_k |a b| that bounds an fp32 dot product, must not exceed the exact-fp32 MFMA
    kernel's on operands with a wide dynamic range (1.5x slack for sampling noise)."""
    dev = _dev()
    rng = np.random.RandomState(K + M)
    shape_a, shape_b = ((K, M) if ak else (M, K)), ((K, N) if bk else (N, K))
    A = (rng.standard_normal(shape_a) * np.exp(2.0 * rng.standard_normal(shape_a))).astype(np.float32)
    B = (rng.standard_normal(shape_b) * 0.05).astype(np.float32)
    A64, B64 = (A.T if ak else A).astype(np.float64), (B if bk else B.T).astype(np.float64)
    ref, mag = A64 @ B64, np.abs(A64) @ np.abs(B64)
    a, b = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    errs = {}
    for entry in ("blh_gemm_f32", "blh_gemm_bf16x3"):
        c = torch.full((splits, M, N), float("nan"), device=dev)
        rc = getattr(native, entry)(st, a.data_ptr(), A.shape[1], ak, b.data_ptr(), B.shape[1], bk,
                                    c.data_ptr(), N, M, N, K, splits, None, None, 0)
        assert rc == 0, native.blh_status_string(rc)
        torch.cuda.synchronize()
        out = c.cpu().numpy().astype(np.float64).sum(axis=0)
        rel = np.abs(out - ref) / mag
        errs[entry] = (rel.max(), np.sqrt((rel ** 2).mean()))
    print("err / sum|ab|  fp32 MFMA: max %.2e rms %.2e | bf16x3: max %.2e rms %.2e" % (
        errs["blh_gemm_f32"] + errs["blh_gemm_bf16x3"]))
    assert errs["blh_gemm_bf16x3"][1] <= 1.5 * errs["blh_gemm_f32"][1]
    assert errs["blh_gemm_bf16x3"][0] <= 1.5 * errs["blh_gemm_f32"][0] + 1e-7
    assert errs["blh_gemm_bf16x3"][0] < 2e-6          # a dropped middle product would show as ~1e-5


def _gemm_fp16x2(native, st, a, A_ld, ak, b, B_ld, bk, c, N, M, K, splits, bias=None, addend=None, ldadd=0):
    dev = a.device
    ws = torch.empty(native.blh_gemm_fp16x2_workspace_bytes(), dtype=torch.uint8, device=dev)
    return native.blh_gemm_fp16x2(st, a.data_ptr(), A_ld, ak, b.data_ptr(), B_ld, bk, c.data_ptr(), N, M, N, K,
                                  splits, bias, addend, ldadd, ws.data_ptr(), 0)


@pytest.mark.parametrize("scale_a,scale_b", [(1.0, 1.0), (3e-7, 0.04), (2.0e4, 1e-3), (1e-12, 1e9),
                                             (7e4, 7e4)])
@pytest.mark.parametrize("M,N,K,ak,bk,splits", [(512, 1024, 1024, 0, 0, 1), (384, 1024, 1024, 0, 1, 1),
                                                (1024, 1024, 4096, 1, 1, 4), (300, 256, 96, 0, 0, 1)])
def test_fp16x2_gemm_error_is_fp32_level_at_any_scale(native, M, N, K, ak, bk, splits, scale_a, scale_b):
    """gemm_dtype = 3: two fp16 pieces per value and a per-operand power-of-two scale taken from
    the operand's largest magnitude.  Gradient-sized (1e-7), activation-sized and huge operands,
    each with four decades of spread inside the tensor, must come out with the error of the
    exact-fp32 MFMA kernel (measured on sum_k |a b|)."""
    dev = _dev()
    rng = np.random.RandomState(K + M + int(np.log10(scale_a) * 7))
    shape_a, shape_b = ((K, M) if ak else (M, K)), ((K, N) if bk else (N, K))
    A = (scale_a * rng.standard_normal(shape_a) * np.exp(2.0 * rng.standard_normal(shape_a))).astype(np.float32)
    B = (scale_b * rng.standard_normal(shape_b) * np.exp(1.0 * rng.standard_normal(shape_b))).astype(np.float32)
    A64, B64 = (A.T if ak else A).astype(np.float64), (B if bk else B.T).astype(np.float64)
    ref, mag = A64 @ B64, np.abs(A64) @ np.abs(B64)
    a, b = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    errs = {}
    for entry in ("blh_gemm_f32", "fp16x2"):
        c = torch.full((splits, M, N), float("nan"), device=dev)
        if entry == "fp16x2":
            rc = _gemm_fp16x2(native, st, a, A.shape[1], ak, b, B.shape[1], bk, c, N, M, K, splits)
        else:
            rc = native.blh_gemm_f32(st, a.data_ptr(), A.shape[1], ak, b.data_ptr(), B.shape[1], bk,
                                     c.data_ptr(), N, M, N, K, splits, None, None, 0)
        assert rc == 0, native.blh_status_string(rc)
        torch.cuda.synchronize()
        out = c.cpu().numpy().astype(np.float64).sum(axis=0)
        assert np.isfinite(out).all(), entry
        rel = np.abs(out - ref) / mag
        errs[entry] = (rel.max(), np.sqrt((rel ** 2).mean()))
    print("err / sum|ab|  fp32 MFMA: max %.2e rms %.2e | fp16x2: max %.2e rms %.2e" % (
        errs["blh_gemm_f32"] + errs["fp16x2"]))
    assert errs["fp16x2"][1] <= 1.5 * errs["blh_gemm_f32"][1]
    assert errs["fp16x2"][0] <= 1.5 * errs["blh_gemm_f32"][0] + 1e-7
    assert errs["fp16x2"][0] < 2e-6


def test_fp16x2_gemm_zero_operand_and_epilogues(native):
    dev = _dev()
    rng = np.random.RandomState(11)
    M, N, K = 260, 384, 96
    A = (1e-5 * rng.standard_normal((M, K))).astype(np.float32)
    Wt = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    add = rng.standard_normal((M, N)).astype(np.float32)
    a, w = torch.from_numpy(A).to(dev), torch.from_numpy(Wt).to(dev)
    bt, at = torch.from_numpy(bias).to(dev), torch.from_numpy(add).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    c = torch.empty(M, N, device=dev)
    assert _gemm_fp16x2(native, st, a, K, 0, w, K, 0, c, N, M, K, 1, bias=bt.data_ptr()) == 0
    torch.cuda.synchronize()
    _close(c.cpu().numpy(), A.astype(np.float64) @ Wt.T.astype(np.float64) + bias, 2e-5, "bias")
    Bk = rng.standard_normal((K, N)).astype(np.float32)
    bk = torch.from_numpy(Bk).to(dev)
    assert _gemm_fp16x2(native, st, a, K, 0, bk, N, 1, at, N, M, K, 1, addend=at.data_ptr(), ldadd=N) == 0
    torch.cuda.synchronize()
    _close(at.cpu().numpy(), A.astype(np.float64) @ Bk.astype(np.float64) + add, 2e-5, "addend")
    z = torch.zeros(M, K, device=dev)          # an all-zero operand: scale 1, exact zeros out
    assert _gemm_fp16x2(native, st, z, K, 0, w, K, 0, c, N, M, K, 1) == 0
    torch.cuda.synchronize()
    assert float(c.abs().max()) == 0.0


def _run_gemm(native, entry, A, B, ak, bk, splits=1):
    dev = _dev()
    M = A.shape[1] if ak else A.shape[0]
    K = A.shape[0] if ak else A.shape[1]
    N = B.shape[1] if bk else B.shape[0]
    a, b = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    c = torch.full((splits, M, N), float("nan"), device=dev)
    if entry == "fp16x2":
        rc = _gemm_fp16x2(native, st, a, A.shape[1], ak, b, B.shape[1], bk, c, N, M, K, splits)
    else:
        rc = getattr(native, entry)(st, a.data_ptr(), A.shape[1], ak, b.data_ptr(), B.shape[1], bk,
                                    c.data_ptr(), N, M, N, K, splits, None, None, 0)
    assert rc == 0, native.blh_status_string(rc)
    torch.cuda.synchronize()
    return c.cpu().numpy().astype(np.float64).sum(axis=0)


@pytest.mark.parametrize("entry", ["blh_gemm_bf16x3", "fp16x2"])
@pytest.mark.parametrize("ak,bk,splits", [(0, 0, 1), (0, 1, 1), (1, 1, 4)])
def test_split_gemm_non_finite_operands_poison_only_their_rows(native, entry, ak, bk, splits):
    """Inf / NaN in an operand: the exact fp32 kernel produces non-finite values in the affected
    output rows / columns only.  The split modes must do the same — in particular fp16x2's
    per-tensor scale (from max |finite value|) must not be dragged to 2^-inf by one overflowed
    element — and keep fp32 accuracy everywhere else."""
    rng = np.random.RandomState(3)
    M, N, K = 384, 256, 512
    A = rng.standard_normal((K, M) if ak else (M, K)).astype(np.float32)
    B = (0.05 * rng.standard_normal((K, N) if bk else (N, K))).astype(np.float32)
    bad_rows, bad_cols = [5, 200], [17]

    def put(arr, transposed, idx, k, val):      # element (row idx, reduction index k)
        if transposed:
            arr[k, idx] = val
        else:
            arr[idx, k] = val
    put(A, ak, 5, 100, np.inf)
    put(A, ak, 200, 7, np.nan)
    put(B, bk, 17, 300, -np.inf)
    exact = _run_gemm(native, "blh_gemm_f32", A, B, ak, bk, splits)
    got = _run_gemm(native, entry, A, B, ak, bk, splits)
    fin_exact = np.isfinite(exact)
    assert not fin_exact[bad_rows].any() and not fin_exact[:, bad_cols].any()
    clean = np.ones((M, N), bool)
    clean[bad_rows] = False
    clean[:, bad_cols] = False
    assert fin_exact[clean].all()
    # same finite / non-finite pattern, fp32-level values where finite
    assert np.array_equal(np.isfinite(got), fin_exact)
    A64 = np.where(np.isfinite(A), A, 0).astype(np.float64)
    B64 = np.where(np.isfinite(B), B, 0).astype(np.float64)
    ref = (A64.T if ak else A64) @ (B64 if bk else B64.T)
    mag = np.abs(A64.T if ak else A64) @ np.abs(B64 if bk else B64.T)
    assert (np.abs(got - ref)[clean] <= 2e-6 * mag[clean]).all()


@pytest.mark.parametrize("ak,bk,splits", [(0, 0, 1), (1, 1, 4)])
def test_fp16x2_gemm_range_edges(native, ak, bk, splits):
    """fp16x2 scale logic at the ends of the fp32 range: a tensor whose largest magnitude is close
    to FLT_MAX, a tensor of denormals (scale 1: they vanish, as far below 2^-126 * 2^13 as the
    exact result itself is), an all-zero operand (the dZ of a dead layer) and a huge dynamic range
    inside one tensor."""
    rng = np.random.RandomState(9)
    M, N, K = 256, 256, 256
    shape_a, shape_b = ((K, M) if ak else (M, K)), ((K, N) if bk else (N, K))
    B = (rng.standard_normal(shape_b) * 2.0 ** -20).astype(np.float32)
    # (1) values up to 1e37 (products 1e31, sums finite in fp32)
    A = (rng.standard_normal(shape_a) * 1e37 / 4).astype(np.float32)
    A64, B64 = (A.T if ak else A).astype(np.float64), (B if bk else B.T).astype(np.float64)
    got = _run_gemm(native, "fp16x2", A, B, ak, bk, splits)
    exact = _run_gemm(native, "blh_gemm_f32", A, B, ak, bk, splits)
    assert np.isfinite(exact).all() and np.isfinite(got).all()
    mag = np.abs(A64) @ np.abs(B64)
    assert (np.abs(got - A64 @ B64) <= 2e-6 * mag).all()
    # (2) all-denormal operand: exact result is at most K * 1e-38 * |b|; fp16x2 returns zeros
    Ad = (rng.standard_normal(shape_a) * 1e-39).astype(np.float32)
    got = _run_gemm(native, "fp16x2", Ad, B, ak, bk, splits)
    assert np.isfinite(got).all() and np.abs(got).max() <= 1e-36
    # (3) all-zero operand -> exact zeros
    got = _run_gemm(native, "fp16x2", np.zeros(shape_a, np.float32), B, ak, bk, splits)
    assert np.abs(got).max() == 0.0
    # (4) one element 2^40 times the rest: the small ones keep an absolute error of 2^-25 of the
    #     scaled maximum, i.e. the error bound of an fp32 dot product containing the big term
    A = rng.standard_normal(shape_a).astype(np.float32)
    if ak:
        A[3, 0] = 2.0 ** 40
    else:
        A[0, 3] = 2.0 ** 40
    A64 = (A.T if ak else A).astype(np.float64)
    got = _run_gemm(native, "fp16x2", A, B, ak, bk, splits)
    exact = _run_gemm(native, "blh_gemm_f32", A, B, ak, bk, splits)
    ref = A64 @ B64
    # row 0 carries the big term; every other row only sees the tensor-wide scale
    assert (np.abs(got[0] - ref[0]) <= 4e-7 * (np.abs(A64[0]) @ np.abs(B64))).all()
    tiny_bound = 2.0 ** 40 * 2.0 ** -24 * np.abs(B64).max() * K
    assert (np.abs(got[1:] - ref[1:]) <= tiny_bound).all()
    assert np.isfinite(exact).all()


@pytest.mark.parametrize("entry", ["blh_gemm_f32", "blh_gemm_bf16x3"])
def test_gemm_bias_and_addend(native, entry):
    dev = _dev()
    rng = np.random.RandomState(5)
    M, N, K = 260, 384, 96
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wt = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    add = rng.standard_normal((M, N)).astype(np.float32)
    a, w = torch.from_numpy(A).to(dev), torch.from_numpy(Wt).to(dev)
    bt, at = torch.from_numpy(bias).to(dev), torch.from_numpy(add).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    c = torch.empty(M, N, device=dev)
    assert getattr(native, entry)(st, a.data_ptr(), K, 0, w.data_ptr(), K, 0, c.data_ptr(), N, M, N, K,
                               1, bt.data_ptr(), None, 0) == 0
    torch.cuda.synchronize()
    _close(c.cpu().numpy(), A.astype(np.float64) @ Wt.T.astype(np.float64) + bias, 2e-5, "bias")
    # dgrad-with-residual form: C = A * B[K,N] + addend, in place on the addend buffer
    Bk = rng.standard_normal((K, N)).astype(np.float32)
    bk = torch.from_numpy(Bk).to(dev)
    assert getattr(native, entry)(st, a.data_ptr(), K, 0, bk.data_ptr(), N, 1, at.data_ptr(), N, M, N, K,
                               1, None, at.data_ptr(), N) == 0
    torch.cuda.synchronize()
    _close(at.cpu().numpy(), A.astype(np.float64) @ Bk.astype(np.float64) + add, 2e-5, "addend")


# ----------------------------------------------------------------------------
# module level against the reference's golden vectors
# ----------------------------------------------------------------------------
FP32_MODES = ["fp32", "bf16x3", "fp16x2"]     # the fp32-accurate GEMM paths: same tests, same tolerances


def _build(g, dev, num_blocks=2, width=1024, state=None, gemm_dtype="fp32"):
    import bilinear_amd
    net = bilinear_amd.BilinearUnit(num_blocks=num_blocks, width=width, gemm_dtype=gemm_dtype)
    st = state if state is not None else g.init_state()
    sd = net.state_dict()
    assert list(sd.keys()) == list(st.keys())
    net.load_state_dict({k: torch.from_numpy(np.array(st[k])).reshape(sd[k].shape) for k in sd})
    net = net.to(dev)
    opt = bilinear_amd.Adam(net.parameters(), lr=1e-3, module=net)
    net.train()
    return net, opt


def _compare_step(g, s, net, opt, raw, clipped, pred, loss, total_norm, rtol):
    worst = 0.0
    worst = max(worst, _close(pred.detach().cpu().numpy(), g.arr("step%d/pred" % s), rtol, "pred"))
    assert abs(float(loss) - g.scalar("step%d/loss" % s)) <= rtol * g.scalar("step%d/loss" % s)
    tn = g.scalar("step%d/total_norm" % s)
    assert abs(float(total_norm) - tn) <= rtol * tn
    for k, p in net.named_parameters():
        if is_prebn_bias(k):
            if raw is not None:
                assert np.abs(raw[k]).max() < 1e-5, k      # hazard H2: mathematically zero
            continue
        if raw is not None:
            worst = max(worst, g.compare("step%d/grad_raw" % s, k, raw[k], rtol) * rtol)
        worst = max(worst, g.compare("step%d/grad_clipped" % s, k, clipped[k], rtol) * rtol)
        stp = opt.state[p]
        g.compare("step%d/exp_avg" % s, k, stp["exp_avg"].cpu().numpy(), rtol)
        g.compare("step%d/exp_avg_sq" % s, k, stp["exp_avg_sq"].cpu().numpy(), 2 * rtol)
        g.compare("step%d/state" % s, k, p.detach().cpu().numpy(), rtol,
                  atol_abs=g.adam_atol(s, k, rtol, p.numel()))
    sd = net.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            g.compare("step%d/state" % s, k, sd[k].cpu().numpy(), rtol)
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(g.arr("step%d/state/%s/full" % (s, k)))
    return worst


@pytest.mark.parametrize("mode", FP32_MODES)
@pytest.mark.parametrize("fname", FIXTURES)
def test_dropin_steps_match_reference(fname, mode):
    """The reference's own step body (train_bilinear.py:66-83) on the drop-in surface:
    zero_grad, forward, nn.MSELoss, backward, clip_grad_norm_, Adam.step."""
    import bilinear_amd
    dev = _dev()
    g = Golden(fname)
    net, opt = _build(g, dev, gemm_dtype=mode)
    criterion = torch.nn.MSELoss()
    step = 1
    for s in range(g.steps):
        x, t = g.batch_xy(s)
        if bilinear_amd.config.bilinear.lr_decay.condition(step):      # the PRODUCT's hook (what train_bilinear.py runs)
            for pg in opt.param_groups:
                pg["lr"] = bilinear_amd.config.bilinear.lr_decay.function(step)
        net.engine.set_dropout_masks(g.masks(s))
        xt, tt = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
        opt.zero_grad()
        pred = net(xt)
        loss = criterion(pred, tt)
        loss.backward()
        raw = {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters()}
        total_norm = bilinear_amd.clip_grad_norm_(net.parameters(), max_norm=1, module=net)
        clipped = {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters()}
        opt.step()
        step += 1
        rtol = RTOL if s else TIGHT
        worst = _compare_step(g, s, net, opt, raw, clipped, pred, loss.item(), total_norm.item(), rtol)
        print("step %d worst normalised error %.2e" % (s, worst))
    net.eval()
    with torch.set_grad_enabled(False):
        pe = net(torch.from_numpy(g.arr("eval/x")).to(dev))
    ref = g.arr("eval/pred")
    assert np.abs(pe.cpu().numpy() - ref).max() <= 5e-3 * np.sqrt((ref ** 2).mean()) + 5e-3


@pytest.mark.parametrize("mode", FP32_MODES)
@pytest.mark.parametrize("fname", FIXTURES)
def test_fused_train_step_matches_reference(fname, mode):
    """The same steps through the one-enqueue fast path (blh_train_step)."""
    import bilinear_amd
    dev = _dev()
    g = Golden(fname)
    net, opt = _build(g, dev, gemm_dtype=mode)
    step = 1
    for s in range(g.steps):
        x, t = g.batch_xy(s)
        if bilinear_amd.config.bilinear.lr_decay.condition(step):
            for pg in opt.param_groups:
                pg["lr"] = bilinear_amd.config.bilinear.lr_decay.function(step)
        net.engine.set_dropout_masks(g.masks(s))
        pred, loss = net.train_step(opt, torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev),
                                    max_norm=1.0)
        step += 1
        clipped = {k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_parameters()}
        stats = opt.last_grad_norm_stats.cpu().numpy()
        _compare_step(g, s, net, opt, None, clipped, pred, loss.item(), stats[0], RTOL if s else TIGHT)


@pytest.mark.parametrize("mode", FP32_MODES)
@pytest.mark.parametrize("batch", [1, 37, 64, 100, 128, 333, 1500, 2048])
def test_eval_forward_matches_oracle(mode, batch):
    """valid_bilinear.py:31,52 — eval mode uses running statistics, no dropout.  (At the larger
    batch every heavy_linear is ONE kernel: bias + BatchNorm + ReLU + skip in the GEMM epilogue; at serving
    sizes, <= 64 rows, the small-batch stage kernels: small_eval_stage_kernel.)"""
    dev = _dev()
    g = Golden(FIXTURES[0])
    st = g.init_state()
    rng = np.random.RandomState(3)
    for k in st:        # non-trivial running statistics
        if k.endswith("running_mean"):
            st[k] = rng.standard_normal(st[k].shape).astype(np.float32) * 0.3
        if k.endswith("running_var"):
            st[k] = (0.5 + rng.random_sample(st[k].shape)).astype(np.float32)
    net, _ = _build(g, dev, state=st, gemm_dtype=mode)
    net.eval()
    x, _ = O.synthetic_batch(77, batch)
    with torch.no_grad():
        pred = net(torch.from_numpy(x).to(dev))
    ref, _ = O.forward(st, x, None, training=False, dtype=np.float64)
    _close(pred.cpu().numpy(), ref, TIGHT, "eval pred")


# ----------------------------------------------------------------------------
# other shapes against the oracle (the reference cannot express them)
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("mode", FP32_MODES)
@pytest.mark.parametrize("nb,width,batch", [(1, 256, 100), (3, 512, 257), (2, 1024, 30), (0, 128, 64),
                                            (2, 1024, 640)])
def test_general_shapes_against_oracle(nb, width, batch, mode):
    dev = _dev()
    st = O.init_state(100 + nb, nb, width)
    rng = np.random.RandomState(nb * 7 + 1)
    for k in st:        # make gamma/beta non-trivial so their gradients are exercised
        if k.endswith(".1.weight"):
            st[k] = (1.0 + 0.2 * rng.standard_normal(st[k].shape)).astype(np.float32)
        if k.endswith(".1.bias"):
            st[k] = (0.1 * rng.standard_normal(st[k].shape)).astype(np.float32)
    net, opt = _build(None, dev, nb, width, state={k: v.copy() for k, v in st.items()}, gemm_dtype=mode)
    x, t = O.synthetic_batch(5, batch)
    # (gates within 1e-4 of zero dropped: the comparison must not depend on how rounding opens them)
    masks = safe_masks(st, x, O.random_masks(9, batch, nb, width))
    net.engine.set_dropout_masks(masks)
    xt, tt = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
    pred = net(xt)
    loss = torch.nn.functional.mse_loss(pred, tt)
    loss.backward()
    ref_pred, cache = O.forward(st, x, masks, training=True, dtype=np.float64)
    ref_loss, dpred = O.mse_loss(ref_pred, t.astype(np.float64))
    ref_grads = O.backward(st, cache, dpred, dtype=np.float64)
    _close(pred.detach().cpu().numpy(), ref_pred, TIGHT, "pred")
    assert abs(loss.item() - ref_loss) <= TIGHT * ref_loss
    for k, p in net.named_parameters():
        if is_prebn_bias(k):
            continue
        _close(p.grad.cpu().numpy(), ref_grads[k], TIGHT * 3, "grad " + k)
    sd = net.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            _close(sd[k].cpu().numpy(), st[k], TIGHT, k)     # O.forward updated st in place


# ----------------------------------------------------------------------------
# Philox dropout: the mask backward regenerates is the one forward applied
# ----------------------------------------------------------------------------
def _philox_masks(native, net, step, batch):
    from bilinear_amd import _native as N
    eng = net.engine
    out = []
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for layer in range(eng.layout.num_heavy):
        m = torch.empty(batch, eng.width, dtype=torch.uint8, device=eng.device)
        d = N.Dropout(None, eng.seed, step, eng.row_offset)
        assert native.blh_dropout_mask(st, ctypes.byref(d), layer, batch, eng.width, m.data_ptr()) == 0
        out.append(m.cpu().numpy())
    return out


def test_philox_dropout_forward_backward_consistent(native):
    dev = _dev()
    nb, width, batch = 2, 1024, 96
    st = O.init_state(41, nb, width)
    net, opt = _build(None, dev, nb, width, state={k: v.copy() for k, v in st.items()})
    net.engine.seed = 1234567
    x, t = O.synthetic_batch(8, batch)
    xt, tt = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
    step0 = net.engine.rng_step
    pred = net(xt)
    torch.nn.functional.mse_loss(pred, tt).backward()
    masks = _philox_masks(native, net, step0, batch)
    rate = np.mean([m.mean() for m in masks])
    assert abs(rate - 0.5) < 0.01, rate
    assert not np.array_equal(masks[0], masks[1])
    ref_pred, cache = O.forward(st, x, masks, training=True, dtype=np.float64)
    _, dpred = O.mse_loss(ref_pred, t.astype(np.float64))
    ref_grads = O.backward(st, cache, dpred, dtype=np.float64)
    _close(pred.detach().cpu().numpy(), ref_pred, TIGHT, "pred (philox)")
    for k, p in net.named_parameters():
        if not is_prebn_bias(k):
            _close(p.grad.cpu().numpy(), ref_grads[k], TIGHT * 3, "grad " + k)
    # next step draws a different mask; same step + seed reproduces it
    m2 = _philox_masks(native, net, step0 + 1, batch)
    assert not np.array_equal(m2[0], masks[0])
    again = _philox_masks(native, net, step0, batch)
    assert all(np.array_equal(a, b) for a, b in zip(again, masks))


def test_philox_mask_independent_of_sharding(native):
    """Rows [64,128) generated with row_offset=64 equal rows 64.. of the unsharded mask
    (data-parallel ranks see the mask of the global batch)."""
    from bilinear_amd import _native as N
    dev = _dev()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    full = torch.empty(128, 256, dtype=torch.uint8, device=dev)
    part = torch.empty(64, 256, dtype=torch.uint8, device=dev)
    assert native.blh_dropout_mask(st, ctypes.byref(N.Dropout(None, 9, 3, 0)), 2, 128, 256, full.data_ptr()) == 0
    assert native.blh_dropout_mask(st, ctypes.byref(N.Dropout(None, 9, 3, 64)), 2, 64, 256, part.data_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(full[64:], part)


# ----------------------------------------------------------------------------
# full benchmark size (B = 4096): size-independent properties
# ----------------------------------------------------------------------------
def test_full_size_properties():
    dev = _dev()
    import bilinear_amd
    torch.manual_seed(0)
    net, opt, step, _ = bilinear_amd.load(dev)
    net.train()
    B = 4096
    x = torch.randn(B, 32, device=dev)
    t = torch.randn(B, 48, device=dev)
    # (1) backward is linear in d(pred): grads(2*dpred) == 2*grads(dpred)
    net.engine.seed = 5
    net.engine.rng_step = 0
    pred = net(x)
    dp = torch.randn_like(pred) / B
    pred.backward(dp)
    g1 = net.engine.grads.clone()
    opt.zero_grad()
    net.engine.rng_step = 0
    pred2 = net(x)
    assert torch.equal(pred, pred2)                  # same seed/step -> same dropout -> bit-equal
    pred2.backward(2 * dp)
    g2 = net.engine.grads.clone()
    rel = ((g2 - 2 * g1).norm() / g2.norm()).item()
    assert rel < 1e-5, rel
    # (2) batch-norm outputs: every pre-activation column has zero mean/unit variance, so the
    #     gradient of each pre-BN Linear bias is (numerically) zero
    lay = net.engine.layout
    for name, off, shape in lay.entries:
        if is_prebn_bias(name):
            assert net.engine.grads[off:off + shape[0]].abs().max().item() < 1e-6
    # (3) the fused step learns: loss on a fixed batch goes down
    opt.zero_grad()
    losses = []
    for _ in range(30):
        _, loss = net.train_step(opt, x, t, max_norm=1.0)
        losses.append(loss.item())
    assert losses[-1] < losses[0] * 0.9, losses
    assert all(np.isfinite(losses))
    # (4) eval after training is finite and deterministic
    net.eval()
    with torch.no_grad():
        a, b = net(x), net(x)
    assert torch.equal(a, b) and torch.isfinite(a).all()


def test_cpu_input_is_rejected():
    import bilinear_amd
    net = bilinear_amd.BilinearUnit()
    with pytest.raises(RuntimeError):
        net(torch.zeros(4, 32))


# ----------------------------------------------------------------------------
# hipGraph-captured step == eager fused step
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("batch,mode,nb", [(64, "fp32", 2), (512, "fp32", 2), (256, "fp16x2", 2), (256, "bf16x3", 2),
                                           (4096, "fp32", 2),         # BASELINE configs[1]: the non-split-K path
                                           (16384, "bf16s", 4),       # BASELINE configs[2]
                                           (2048, "bf16s", 1)])
def test_captured_step_matches_eager(batch, mode, nb):
    import bilinear_amd
    dev = _dev()

    def make():
        torch.manual_seed(5)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, gemm_dtype=mode)
        net.train()
        net.engine.seed = 99
        return net, opt

    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(batch, 32, generator=g).to(dev) for _ in range(4)]
    ts = [torch.randn(batch, 48, generator=g).to(dev) for _ in range(4)]
    net_a, opt_a = make()
    net_b, opt_b = make()
    net_a.engine.ensure(dev)
    step = bilinear_amd.CapturedTrainStep(net_b, opt_b, batch)
    assert torch.equal(net_a.engine.params, net_b.engine.params)   # capture did not move the model
    for i in range(4):
        if i == 2:                               # lr-decay hook between replays
            for o in (opt_a, opt_b):
                o.param_groups[0]["lr"] = 5e-4
        pa, la = net_a.train_step(opt_a, xs[i], ts[i])
        pb, lb = step(xs[i], ts[i])
        assert torch.equal(pa, pb), "prediction differs at step %d" % i
        assert la.item() == lb.item()
    torch.cuda.synchronize()
    assert torch.equal(net_a.engine.params, net_b.engine.params)
    assert torch.equal(opt_a._exp_avg_sq, opt_b._exp_avg_sq)
    assert torch.equal(net_a.engine.bn_running, net_b.engine.bn_running)
    assert int(net_b.encode[1].num_batches_tracked) == 4
    assert opt_b.state_dict()["state"][0]["step"] == 4


# ----------------------------------------------------------------------------
# "next" row: MPJPE of valid_bilinear.py:53-83 on the device
# ----------------------------------------------------------------------------
def test_mpjpe_matches_oracle():
    from bilinear_amd.metrics import MPJPE
    dev = _dev()
    rng = np.random.RandomState(12)
    B = 777
    pred = rng.standard_normal((B, 48)).astype(np.float32)
    tgt = rng.standard_normal((B, 48)).astype(np.float32)
    mean = (rng.standard_normal(48) * 100).astype(np.float32)
    std = (50 + 200 * rng.random_sample(48)).astype(np.float32)
    ids = rng.randint(0, 15, size=B).astype(np.int32)
    names = ["a%d" % i for i in range(15)]
    m = MPJPE(names, torch.from_numpy(mean), torch.from_numpy(std), dev)
    d1 = m.update(torch.from_numpy(pred[:400]).to(dev), torch.from_numpy(tgt[:400]).to(dev), torch.from_numpy(ids[:400]))
    d2 = m.update(torch.from_numpy(pred[400:]).to(dev), torch.from_numpy(tgt[400:]).to(dev), torch.from_numpy(ids[400:]))
    ref = O.mpjpe_sum(pred.astype(np.float64), tgt.astype(np.float64), mean.astype(np.float64), std.astype(np.float64))
    got = np.concatenate([d1.cpu().numpy(), d2.cpu().numpy()])
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
    per_action, avg = m.result()
    for i, n in enumerate(names):
        sel = ids == i
        assert abs(per_action[n] - ref[sel].sum() / (sel.sum() * 16)) <= 1e-5 * per_action[n]
    assert abs(avg - ref.sum() / (B * 16)) <= 1e-5 * avg


# ----------------------------------------------------------------------------
# (gemm_dtype = "bf16", round 1's mixed mode — fp32 tensors, operands rounded to bf16 on load — was
#  removed in round 4 together with its tests: gemm_dtype "bf16s" superseded it; tests/test_gpu_bf16s.py,
#  tests/test_gpu_timed_path.py.)  The model descriptor must refuse it:
def test_removed_mixed_mode_is_refused(native):
    from bilinear_amd import _native as N
    d = N.ModelDesc(2, 1024, 32, 48, 1)
    assert native.blh_param_arena_floats(ctypes.byref(d)) == -1          # BLH_ERR_INVALID_ARGUMENT
    import bilinear_amd
    with pytest.raises(ValueError):
        bilinear_amd.BilinearUnit(2, 1024, gemm_dtype="bf16")


def test_reset_statistics_cumulative_average():
    """model/bilinear.py:43-55: momentum=None makes BatchNorm keep the cumulative average
    of the batch statistics (factor 1/num_batches_tracked)."""
    dev = _dev()
    nb, width, batch = 1, 256, 160
    st = O.init_state(77, nb, width)
    net, _ = _build(None, dev, nb, width, state={k: v.copy() for k, v in st.items()})
    net.reset_statistics()
    assert net.encode[1].momentum is None
    for i in range(3):
        x, _ = O.synthetic_batch(300 + i, batch)
        masks = O.random_masks(400 + i, batch, nb, width)
        net.engine.set_dropout_masks(masks)
        with torch.no_grad():
            net(torch.from_numpy(x).to(dev))
        O.forward(st, x, masks, training=True, dtype=np.float64, momentum=None)
    sd = net.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            _close(sd[k].cpu().numpy(), st[k], TIGHT, k)
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == 3


def test_checkpoint_round_trip_and_reference_format(tmp_path):
    """train_bilinear.py:92-104 / model/bilinear.py:63-83: a checkpoint written in the
    reference's format resumes exactly; a checkpoint produced by plain PyTorch (the CPU port
    with torch.optim.Adam) loads into the HIP module."""
    import bilinear_amd
    from oracle import torch_port as TP
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    xs = [torch.randn(64, 32, generator=g).to(dev) for _ in range(3)]
    ts = [torch.randn(64, 48, generator=g).to(dev) for _ in range(3)]

    torch.manual_seed(3)
    net, opt, step, _ = bilinear_amd.load(dev)
    net.train()
    net.engine.seed = 5
    for i in range(2):
        net.train_step(opt, xs[i], ts[i])
    d = tmp_path / "parameter"
    d.mkdir()
    torch.save({"epoch": 1, "step": 3, "state": net.state_dict(), "optimizer": opt.state_dict()},
               str(d / "1.save"))
    net.train_step(opt, xs[2], ts[2])
    want = net.engine.params.clone()

    net2, opt2, step2, epoch2 = bilinear_amd.load(dev, parameter_dir=str(d))
    assert (step2, epoch2) == (3, 1)
    net2.train()
    net2.engine.seed = 5
    net2.engine.rng_step = 2                 # the reference does not checkpoint its RNG either
    net2.train_step(opt2, xs[2], ts[2])
    assert torch.equal(net2.engine.params, want)
    assert opt2.state_dict()["state"][0]["step"] == 3

    # a checkpoint made entirely by PyTorch on the CPU
    port = TP.LifterPort(2, 1024)
    popt = torch.optim.Adam(port.parameters(), lr=1e-3)
    port.train()
    TP.train_step(port, popt, xs[0].cpu(), ts[0].cpu())
    d2 = tmp_path / "ref"
    d2.mkdir()
    torch.save({"epoch": 7, "step": 2, "state": port.state_dict(), "optimizer": popt.state_dict()},
               str(d2 / "7.save"))
    net3, opt3, step3, epoch3 = bilinear_amd.load(dev, parameter_dir=str(d2))
    assert (step3, epoch3) == (2, 7)
    net3.eval(); port.eval()
    with torch.no_grad():
        a = net3(xs[1]).cpu().numpy()
        b = port(xs[1].cpu()).numpy()
    _close(a, b, TIGHT, "eval after loading a PyTorch checkpoint")
    net3.train()
    net3.train_step(opt3, xs[1], ts[1])      # Adam moments restored into the flat arenas
    assert opt3._t == 2 and float(opt3._exp_avg_sq.abs().sum()) > 0


def test_batchnorm_statistics_large_batch_with_offset():
    """SURVEY.md hazard H1: column statistics over a large batch whose mean dwarfs its
    spread (|mean|/std = 1000) — the tile-wise (mean, M2) + Chan merge must not cancel."""
    from bilinear_amd import _native as N
    dev = _dev()
    native = N.lib()
    M, K, Nn = 65536, 32, 128
    rng = np.random.RandomState(1)
    A = rng.standard_normal((M, K)).astype(np.float32)
    Wt = (rng.standard_normal((Nn, K)) * 0.01).astype(np.float32)
    bias = np.full(Nn, 10.0, np.float32)           # z = 10 +- 0.06
    a, w, b = (torch.from_numpy(v).to(dev) for v in (A, Wt, bias))
    Z = torch.empty(M, Nn, device=dev)
    part = torch.empty(M // 128, 2, Nn, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert native.blh_linear_fwd_stats(st, a.data_ptr(), w.data_ptr(), b.data_ptr(), Z.data_ptr(),
                                       part.data_ptr(), M, Nn, K) == 0
    torch.cuda.synchronize()
    z = Z.cpu().numpy().astype(np.float64)
    p = part.cpu().numpy().astype(np.float64)
    mean = p[:, 0].mean(axis=0)
    m2 = (p[:, 1] + 128 * (p[:, 0] - mean) ** 2).sum(axis=0)
    assert np.abs(mean - z.mean(axis=0)).max() <= 1e-6 * 10
    assert np.abs(m2 / M - z.var(axis=0)).max() <= 2e-3 * z.var(axis=0).max()


def test_standalone_heavy_linear_stage():
    """heavy_linear (model/bilinear.py:7-13) called on its own: forward/backward of one stage
    against plain PyTorch on the CPU with the same dropout mask (recovered from the output)."""
    import bilinear_amd
    dev = _dev()
    torch.manual_seed(21)
    stage = bilinear_amd.heavy_linear(64, 192).to(dev).train()
    ref = torch.nn.Sequential(torch.nn.Linear(64, 192), torch.nn.BatchNorm1d(192), torch.nn.ReLU())
    ref[0].load_state_dict({k: v.cpu() for k, v in stage[0].state_dict().items()})
    with torch.no_grad():
        stage[1].weight.uniform_(0.5, 1.5); stage[1].bias.normal_(0, 0.2)
    ref[1].load_state_dict({k: v.cpu() for k, v in stage[1].state_dict().items()})
    ref.train()
    x = torch.randn(300, 64)
    xg = x.to(dev).requires_grad_(True)
    out = stage(xg)
    xr = x.clone().requires_grad_(True)
    y = ref(xr)
    keep = (out.detach().cpu() != 0) | (y.detach() <= 0)       # mask irrelevant where relu is 0
    assert abs(keep[y.detach() > 0].float().mean().item() - 0.5) < 0.03
    yr = y * keep.float() * 2.0
    _close(out.detach().cpu().numpy(), yr.detach().numpy(), TIGHT, "stage output")
    g = torch.randn(300, 192)
    out.backward(g.to(dev))
    yr.backward(g)
    _close(xg.grad.cpu().numpy(), xr.grad.numpy(), TIGHT * 3, "d input")
    _close(stage[0].weight.grad.cpu().numpy(), ref[0].weight.grad.numpy(), TIGHT * 3, "d weight")
    _close(stage[1].weight.grad.cpu().numpy(), ref[1].weight.grad.numpy(), TIGHT * 3, "d gamma")
    _close(stage[1].bias.grad.cpu().numpy(), ref[1].bias.grad.numpy(), TIGHT * 3, "d beta")
    _close(stage[1].running_var.cpu().numpy(), ref[1].running_var.numpy(), TIGHT, "running_var")
    assert int(stage[1].num_batches_tracked) == 1
    stage.eval(); ref.eval()
    with torch.no_grad():
        _close(stage(x.to(dev)).cpu().numpy(), ref(x).numpy(), TIGHT, "eval stage")


# ----------------------------------------------------------------------------
# input pipeline on the device
# ----------------------------------------------------------------------------
def test_device_dataset_feeds_training_and_metric():
    """The device-resident split (bilinear_amd.data.DevicePoseDataset) against the oracle's
    restatement of H36M/data.py, then end to end: a few fused steps on shuffled batches reduce
    the loss, and the MPJPE of the validation split equals the oracle's per-action loop."""
    import bilinear_amd
    from bilinear_amd.data import DevicePoseDataset, synthetic_raw
    from bilinear_amd.metrics import MPJPE
    dev = _dev()
    raw_tr, raw_va = synthetic_raw(8192, seed=11), synthetic_raw(1000, seed=12)
    train = DevicePoseDataset(raw_tr, dev, seed=3)
    valid = DevicePoseDataset(raw_va, dev, stats_from=train)
    ptr, str_ = O.h36m_flatten(raw_tr["part"], raw_tr["S"])
    pva, sva = O.h36m_flatten(raw_va["part"], raw_va["S"])
    mx, sx = O.h36m_stats(ptr)
    mt, st = O.h36m_stats(str_)
    _close(valid.x.cpu().numpy(), O.h36m_normalise(pva, mx, sx), 2e-5, "valid x")
    _close(valid.t.cpu().numpy(), O.h36m_normalise(sva, mt, st), 2e-5, "valid t")
    torch.manual_seed(0)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=1, width=256)
    net.train()
    losses = []
    for epoch in range(3):
        for x, t in train.epoch(epoch, 1024, shuffle=True):
            _, loss = net.train_step(opt, x, t, max_norm=1.0)
            losses.append(float(loss.item()))
    assert losses[-1] < 0.7 * losses[0], losses
    net.eval()
    metric = MPJPE(valid.action_names, valid.norm_mean, valid.norm_stddev, dev)
    preds = []
    with torch.no_grad():
        for x, t, a in valid.epoch(0, 256, with_actions=True):
            p = net(x)
            metric.update(p, t, a)
            preds.append(p)
    per_action, avg = metric.result()
    pred = torch.cat(preds).cpu().numpy().astype(np.float64)
    gt = O.h36m_normalise(sva, mt, st).astype(np.float64)
    dist = O.mpjpe_sum(pred, gt, mt.astype(np.float64), st.astype(np.float64))
    ref_sum, ref_cnt = {}, {}
    for d, n in zip(dist, raw_va["image"]):          # the per-sample loop of valid_bilinear.py:61-70
        k = O.h36m_decode_action(n)
        ref_sum[k] = ref_sum.get(k, 0.0) + float(d)
        ref_cnt[k] = ref_cnt.get(k, 0) + 1
    for k in ref_sum:
        assert abs(per_action[k] - ref_sum[k] / (ref_cnt[k] * 16)) <= 1e-4 * per_action[k] + 1e-3
    tot = sum(ref_sum.values()) / (sum(ref_cnt.values()) * 16)
    assert abs(avg - tot) <= 1e-4 * tot + 1e-3


@pytest.mark.parametrize("protocol", ["SH", "SH+FT"])
def test_detected_joint_protocols_through_the_device_path(tmp_path, protocol):
    """SURVEY.md 8(f) rank 4 (/root/reference/H36M/protocol.py:1-4, H36M/data.py:31-34): the SH / SH+FT
    protocols only select which ``{task}_{protocol}.bin`` pickles feed the SAME path.  On the device:
    pickles of the reference's layout (python lists) -> DevicePoseDataset.from_pickles(protocol=...) ->
    one fused step of the HIP path on the first batch == the oracle's step on the oracle's own
    restatement of the loader (H36M/data.py:36-59,108-110), and the detected-joint inputs really are
    the ones that were fed (they differ from the ground-truth protocol's)."""
    import pickle

    import bilinear_amd
    from bilinear_amd.data import DevicePoseDataset, synthetic_raw
    dev = _dev()
    base_tr, base_va = synthetic_raw(1024, seed=21), synthetic_raw(128, seed=22)
    rng = np.random.RandomState(23)
    for proto, noise in (("GT", 0.0), ("SH", 6.0), ("SH+FT", 3.0)):
        for task, raw in (("train", base_tr), ("valid", base_va)):
            r = dict(raw)
            r["part"] = (raw["part"] + noise * rng.standard_normal(raw["part"].shape)).astype(np.float32).tolist()
            r["S"] = raw["S"].tolist()
            with open(tmp_path / ("%s_%s.bin" % (task, proto)), "wb") as f:
                pickle.dump(r, f)
    train, valid = DevicePoseDataset.from_pickles(str(tmp_path), dev, protocol=protocol)
    gt_train, _ = DevicePoseDataset.from_pickles(str(tmp_path), dev, protocol="GT")
    assert train.x.is_cuda and not torch.allclose(train.x, gt_train.x) and torch.allclose(train.t, gt_train.t)
    with open(tmp_path / ("train_%s.bin" % protocol), "rb") as f:
        raw_tr = pickle.load(f)
    ptr, str_ = O.h36m_flatten(np.asarray(raw_tr["part"], np.float32), np.asarray(raw_tr["S"], np.float32))
    mx, sx = O.h36m_stats(ptr)
    mt, st_ = O.h36m_stats(str_)
    xo, to = O.h36m_normalise(ptr, mx, sx), O.h36m_normalise(str_, mt, st_)
    batch, nb, width = 512, 1, 256
    x, t = next(iter(train.epoch(0, batch, shuffle=False)))
    _close(x.cpu().numpy(), xo[:batch], 2e-5, "SH batch x")
    _close(t.cpu().numpy(), to[:batch], 2e-5, "SH batch t")
    st = O.init_state(77, nb, width)
    net = bilinear_amd.BilinearUnit(nb, width)
    sd = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(np.array(st[k])).reshape(sd[k].shape) for k in sd})
    net = net.to(dev).train()
    opt = bilinear_amd.Adam(net.parameters(), lr=1e-3, module=net)
    masks = O.random_masks(5, batch, nb, width)
    net.engine.set_dropout_masks(masks)
    pred, loss = net.train_step(opt, x, t)
    torch.cuda.synchronize()
    ost = {k: v.copy() for k, v in st.items()}
    oopt = O.adam_init(ost, O.param_keys(nb))
    r = O.train_step(ost, oopt, xo[:batch], to[:batch], masks, 1e-3)
    _close(pred.cpu().numpy(), r["pred"], 1e-4, "%s step prediction" % protocol)
    assert abs(loss.item() - r["loss"]) <= 1e-5 * r["loss"]
    w = dict(net.named_parameters())["decode.weight"].detach().cpu().numpy()
    assert np.abs(w - ost["decode.weight"]).max() <= 2e-5 + 1e-3 * 1e-3     # post-Adam: |update| <= lr


@pytest.mark.parametrize("loss_scale,gamma_scale", [(1e-7, 1.0), (1e4, 1.0), (1.0, 300.0), (1e-5, 0.01)])
def test_fp16x2_mode_keeps_fp32_accuracy_when_magnitudes_move(loss_scale, gamma_scale):
    """Network level: tiny / huge gradients (loss scaled) and huge / tiny activations (BatchNorm
    gamma, beta scaled) through gemm_dtype = "fp16x2" against the fp64 oracle, with the tolerance
    of the exact path: the per-tensor scales must follow the magnitudes."""
    dev = _dev()
    nb, width, batch = 2, 1024, 640
    st = O.init_state(321, nb, width)
    rng = np.random.RandomState(5)
    for k in st:
        if k.endswith(".1.weight"):
            st[k] = (gamma_scale * (1.0 + 0.2 * rng.standard_normal(st[k].shape))).astype(np.float32)
        if k.endswith(".1.bias"):
            st[k] = (gamma_scale * 0.1 * rng.standard_normal(st[k].shape)).astype(np.float32)
    net, opt = _build(None, dev, nb, width, state={k: v.copy() for k, v in st.items()}, gemm_dtype="fp16x2")
    x, t = O.synthetic_batch(5, batch)
    masks = safe_masks(st, x, O.random_masks(9, batch, nb, width), thr=1e-4 * gamma_scale)
    net.engine.set_dropout_masks(masks)
    xt, tt = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
    pred = net(xt)
    loss = torch.nn.functional.mse_loss(pred, tt) * loss_scale
    loss.backward()
    ref_pred, cache = O.forward(st, x, masks, training=True, dtype=np.float64)
    ref_loss, dpred = O.mse_loss(ref_pred, t.astype(np.float64))
    ref_grads = O.backward(st, cache, dpred * loss_scale, dtype=np.float64)
    _close(pred.detach().cpu().numpy(), ref_pred, TIGHT, "pred")
    for k, p in net.named_parameters():
        if is_prebn_bias(k):
            continue
        g = p.grad.cpu().numpy()
        assert np.isfinite(g).all(), k
        _close(g, ref_grads[k], TIGHT * 3, "grad " + k)


# ----------------------------------------------------------------------------
# torch.compile: the fused step as a custom operator inside a compiled function
# ----------------------------------------------------------------------------
def test_train_step_operator_under_torch_compile_fullgraph():
    """``torch.ops.bilinear_hip.train_step`` inside ``torch.compile(fullgraph=True)``: dynamo traces
    through the operator's fake implementation (no graph break) and the compiled function runs the
    same native step as the eager call, bit for bit."""
    import bilinear_amd
    dev = _dev()
    x = torch.randn(256, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    t = torch.randn(256, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(2))

    def make():
        torch.manual_seed(3)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=1, width=256)
        net.train()
        eng = net.engine
        eng.ensure(dev)
        eng.seed = 11
        opt._ensure_moments(eng)
        return net, opt, eng

    def step(x, t, params, grads, m, v, running, nbt, ws, stats, args):
        return torch.ops.bilinear_hip.train_step(x, t, params, grads, m, v, running, nbt, ws, stats, None, *args)

    outs = []
    for compiled in (False, True):
        net, opt, eng = make()
        ws = eng.workspace(256)
        args = eng._op_args() + (eng.seed, 0, 0, 0.1, 1e-3, 0.9, 0.999, 1e-8, 1.0, 1)
        fn = torch.compile(step, fullgraph=True, backend="aot_eager") if compiled else step
        with torch.no_grad():
            pred, loss = fn(x, t, eng.params, eng.grads, opt._exp_avg, opt._exp_avg_sq, eng.bn_running,
                            eng.bn_nbt, ws, opt._stats, args)
        torch.cuda.synchronize()
        outs.append((pred.clone(), loss.clone(), eng.params.clone(), opt._exp_avg_sq.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


def test_drop_in_step_through_the_differentiable_operator(monkeypatch):
    """The five-call step of /root/reference/train_bilinear.py:75-83 (zero_grad, forward, MSELoss, backward,
    clip_grad_norm_, Adam.step) with ``loss.backward()`` going through ``torch.ops.bilinear_hip.lifter_train``'s
    registered autograd formula (torch.library.register_autograd): bit-identical to the autograd.Function bridge;
    the operator is FUNCTIONAL (round 5: it returns the saved activations and the updated BatchNorm statistics, the
    module's buffers advance through ordinary copy_ calls) — so two forwards may be outstanding at once, each
    backward reading its own saved activations; the same forward + loss under torch.compile(fullgraph=True)
    (AOTAutograd traces the formula down to ``lifter_backward``) gives the same gradients."""
    import bilinear_amd
    from bilinear_amd.model.bilinear import _LifterFunction
    dev = _dev()
    x = torch.randn(512, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    t = torch.randn(512, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(2))

    def make():
        torch.manual_seed(3)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=1, width=256)
        net.train()
        net.engine.ensure(dev)
        net.engine.seed = 11
        return net, opt

    import bilinear_amd.model.bilinear as MB
    assert MB.EAGER_AUTOGRAD == "function"         # eager default: the lean bridge (the operator costs host time);
    monkeypatch.setattr(MB, "EAGER_AUTOGRAD", "op")   # this test drives the operator in eager mode
    request_default = lambda: monkeypatch.setattr(MB, "EAGER_AUTOGRAD", "function")
    outs = {}
    for path in ("operator", "function"):
        net, opt = make()
        eng = net.engine
        for _ in range(2):
            opt.zero_grad()
            if path == "operator":
                pred = net(x)                      # BilinearUnit.forward -> lifter_train
                assert type(pred.grad_fn).__name__ != "_LifterFunctionBackward"
            else:
                pred = _LifterFunction.apply(x, eng, *[p for _, p, _, _ in eng._named_params()])
            loss = torch.nn.functional.mse_loss(pred, t)
            loss.backward()
            grads = torch.cat([p.grad.reshape(-1) for _, p, _, _ in eng._named_params()])
            bilinear_amd.clip_grad_norm_(net.parameters(), 1.0, module=net)
            opt.step()
        torch.cuda.synchronize()
        outs[path] = (pred.detach().clone(), grads.clone(), eng.params.clone(), eng.bn_running.clone(), eng.bn_nbt.clone())
    for a, b in zip(outs["operator"], outs["function"]):
        assert torch.equal(a, b)
    assert int(outs["operator"][4][0]) == 2
    # two outstanding forwards: each backward reads the activations ITS forward returned
    net, opt = make()
    p1 = net(x)
    p2 = net(x)                                    # (another dropout step: other masks)
    g1 = torch.autograd.grad(p1.sum(), [p for _, p, _, _ in net.engine._named_params()])
    net_b, _ = make()
    q1 = net_b(x)
    h1 = torch.autograd.grad(q1.sum(), [p for _, p, _, _ in net_b.engine._named_params()])
    assert all(torch.equal(a, b) for a, b in zip(g1, h1))
    del p2
    # torch.compile(fullgraph=True): forward + loss traced through the operator, backward through its formula
    net, opt = make()
    eng = net.engine
    named = eng._named_params()
    views = [p for _, p, _, _ in named]
    offs = [int(off) for _, _, off, _ in named]
    args = eng._op_args() + (eng.seed, 0, 0, 0.1)
    wsb = eng.layout.workspace_bytes(512)

    def fwd_loss(x, t, views, arena, running, nbt):
        pred, _, new_running, new_nbt = torch.ops.bilinear_hip.lifter_train(x, views, arena, running, nbt, None, *args,
                                                                            offs, wsb)
        return torch.nn.functional.mse_loss(pred, t), new_running, new_nbt

    compiled = torch.compile(fwd_loss, fullgraph=True, backend="aot_eager")
    run0, nbt0 = eng.bn_running.clone(), eng.bn_nbt.clone()
    loss_c, new_running, new_nbt = compiled(x, t, views, eng.params, eng.bn_running, eng.bn_nbt)
    assert torch.equal(eng.bn_running, run0) and torch.equal(eng.bn_nbt, nbt0)      # inputs untouched
    loss_c.backward()
    torch.cuda.synchronize()
    g_compiled = torch.cat([p.grad.reshape(-1) for p in views])
    net2, _ = make()
    opt2_pred = net2(x)
    torch.nn.functional.mse_loss(opt2_pred, t).backward()
    torch.cuda.synchronize()
    g_eager = torch.cat([p.grad.reshape(-1) for _, p, _, _ in net2.engine._named_params()])
    assert torch.equal(g_compiled, g_eager)
    assert torch.equal(new_running, net2.engine.bn_running) and torch.equal(new_nbt, net2.engine.bn_nbt)
    assert abs(float(loss_c) - float(torch.nn.functional.mse_loss(opt2_pred, t))) == 0.0
    request_default()
    pred = net2(x)
    assert type(pred.grad_fn).__name__ == "_LifterFunctionBackward"


def test_compiled_module_advances_batchnorm_statistics_under_cudagraph_mode(monkeypatch):
    """with a functional schema that hid its writes, torch.compile was free to drop or re-order
    the operator, and under CUDA-graph modes (mode="reduce-overhead": static-input copies) BatchNorm's running
    statistics could stop advancing.  The module compiled with mode="reduce-overhead" (explicit dropout masks, so
    that no integer argument changes between calls): running statistics and num_batches_tracked advance on every
    call exactly as in eager mode, and the gradients equal the eager ones."""
    import bilinear_amd
    dev = _dev()
    batch, width, nb = 512, 256, 1
    x = torch.randn(batch, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    t = torch.randn(batch, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    g = torch.Generator(device=dev).manual_seed(9)
    masks = [(torch.rand(batch, width, device=dev, generator=g) < 0.5).to(torch.uint8) for _ in range(1 + 2 * nb)]

    def make():
        torch.manual_seed(3)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width)
        net.train()
        net.engine.ensure(dev)
        net.engine.set_dropout_masks(masks)
        return net

    def run(net, fn, steps):
        out = []
        for _ in range(steps):
            for p in net.parameters():
                p.grad = None
            loss = torch.nn.functional.mse_loss(fn(x), t)
            loss.backward()
            torch.cuda.synchronize()
            out.append((loss.detach().clone(), torch.cat([p.grad.reshape(-1) for _, p, _, _ in net.engine._named_params()]),
                        net.engine.bn_running.clone(), net.engine.bn_nbt.clone()))
        return out

    eager_net = make()
    eager = run(eager_net, eager_net, 3)
    comp_net = make()
    compiled = torch.compile(comp_net, mode="reduce-overhead")
    comp = run(comp_net, compiled, 3)
    for i, (e, c) in enumerate(zip(eager, comp)):
        assert int(c[3][0]) == i + 1, ("num_batches_tracked", i, c[3])
        for a, b, what in zip(e, c, ("loss", "gradients", "running statistics", "num_batches_tracked")):
            assert torch.equal(a, b), (what, i)
    assert not torch.equal(comp[0][2], comp[2][2])           # the statistics really moved


def test_contexts_of_one_device_share_the_side_stream_and_outlive_each_other():
    """blh_context_create: the side stream is one lowest-priority stream per device and process,
    reference-counted (include/bilinear_hip.h, "Context"); destroying one context must leave the
    others usable."""
    import bilinear_amd
    from bilinear_amd import _native as N
    dev = _dev()
    a, b = N.Context(dev), N.Context(dev)
    sa, sb = a.side_stream(), b.side_stream()
    assert sa and sa == sb
    del a
    import gc
    gc.collect()
    torch.manual_seed(0)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=1, width=1024)
    net.train()
    x = torch.randn(2048, 32, device=dev)
    t = torch.randn(2048, 48, device=dev)
    net.engine.ensure(dev)
    assert net.engine.ctx.side_stream() == sb
    pred, loss = net.train_step(opt, x, t, max_norm=1.0)     # two-stream backward on the shared stream
    torch.cuda.synchronize()
    assert torch.isfinite(loss).item() and torch.isfinite(net.engine.params).all().item()
    assert b.side_stream() == sb


# ------------------------------------------------------------------------------------------------------------
# the operator path with explicit masks, and many outstanding forwards of mixed formats
def _op_net(dev, nb=1, width=256, seed=3):
    import bilinear_amd
    torch.manual_seed(seed)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width)
    net.train()
    net.engine.ensure(dev)
    net.engine.seed = 11
    return net, opt


def test_operator_path_moves_or_refuses_explicit_masks(monkeypatch):
    """EAGER_AUTOGRAD = "op": explicit dropout masks handed over as NumPy arrays (host memory) are moved to the device
    before the operator sees them, and masks built for another batch raise — neither reaches the kernels as a bad
    pointer (forward_train_autograd had lost the _drop_struct call)."""
    import bilinear_amd.model.bilinear as MB
    dev = _dev()
    monkeypatch.setattr(MB, "EAGER_AUTOGRAD", "op")
    nb, width, batch = 1, 256, 512
    rng = np.random.RandomState(5)
    masks = [(rng.rand(batch, width) < 0.5).astype(np.uint8) for _ in range(1 + 2 * nb)]
    x = torch.randn(batch, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    outs = []
    for as_numpy in (True, False):
        net, _ = _op_net(dev, nb, width)
        net.engine.set_dropout_masks(masks if as_numpy else [torch.from_numpy(m).to(dev) for m in masks])
        if as_numpy:
            assert net.engine.masks.device.type == "cpu"          # (the hazard: still host memory here)
        pred = net(x)
        assert type(pred.grad_fn).__name__ != "_LifterFunctionBackward"
        assert net.engine.masks.device == x.device
        g = torch.autograd.grad(pred.square().mean(), [p for _, p, _, _ in net.engine._named_params()])
        torch.cuda.synchronize()
        outs.append((pred.detach().clone(), torch.cat([t.reshape(-1) for t in g])))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    net, _ = _op_net(dev, nb, width)
    net.engine.set_dropout_masks(masks)
    with pytest.raises(RuntimeError, match="dropout masks have shape"):
        net(x[:256])                                              # masks of 512 rows, batch of 256


def test_many_outstanding_forwards_of_mixed_batch_sizes_each_get_their_own_backward(monkeypatch):
    """Operator path, one context: five forwards outstanding at once — 4096, 4096, 64, 4096, 1000 rows, i.e. the
    encode-without-Z0 format, the small-batch format and again — then their backwards in forward order.  Every
    backward must read its activations in the format its forward wrote (the context keeps a record per workspace);
    compared with the same forward + backward run alone.  ADVICE r05: the old table of four fell back to "the last
    forward" and silently produced wrong gradients."""
    import bilinear_amd.model.bilinear as MB
    dev = _dev()
    monkeypatch.setattr(MB, "EAGER_AUTOGRAD", "op")
    sizes = [4096, 4096, 64, 4096, 1000, 640, 4096]
    gen = torch.Generator(device=dev).manual_seed(9)
    xs = [torch.randn(b, 32, device=dev, generator=gen) for b in sizes]

    def alone(i):
        net, _ = _op_net(dev, 1, 256)
        net.engine.rng_step = i                    # the same Philox step the shared model uses for forward i
        pred = net(xs[i])
        g = torch.autograd.grad(pred.square().mean(), [p for _, p, _, _ in net.engine._named_params()])
        torch.cuda.synchronize()
        return pred.detach().clone(), torch.cat([t.reshape(-1) for t in g])

    want = [alone(i) for i in range(len(sizes))]
    net, _ = _op_net(dev, 1, 256)
    running0 = net.engine.bn_running.clone()
    preds = []
    for i, x in enumerate(xs):
        net.engine.bn_running.copy_(running0)      # (the running statistics do not enter a train-mode forward)
        preds.append(net(x))
    params = [p for _, p, _, _ in net.engine._named_params()]
    for i, pred in enumerate(preds):
        g = torch.autograd.grad(pred.square().mean(), params)
        torch.cuda.synchronize()
        assert torch.equal(pred.detach(), want[i][0]), i
        assert torch.equal(torch.cat([t.reshape(-1) for t in g]), want[i][1]), (i, sizes[i])


def test_second_backward_with_retain_graph_matches_the_first_or_is_refused(monkeypatch):
    """loss.backward(retain_graph=True) twice on the operator path.  The multi-launch backward (with and without Z0)
    writes gradient buffers only — the activations, keep bits and statistics it reads stay — so the second pass gives
    the same gradients.  The small-batch backward (at most 384 rows) works IN the buffers it reads: its second pass
    is refused (RuntimeError from BLH_ERR_INVALID_ARGUMENT), never silently wrong."""
    import bilinear_amd.model.bilinear as MB
    dev = _dev()
    monkeypatch.setattr(MB, "EAGER_AUTOGRAD", "op")
    gen = torch.Generator(device=dev).manual_seed(4)
    for batch in (64, 300, 1000, 4096):
        net, _ = _op_net(dev, 1, 256)
        x = torch.randn(batch, 32, device=dev, generator=gen)
        params = [p for _, p, _, _ in net.engine._named_params()]
        loss = net(x).square().mean()
        g1 = torch.autograd.grad(loss, params, retain_graph=True)
        g1 = torch.cat([t.reshape(-1) for t in g1]).clone()
        if batch <= 384:
            with pytest.raises(RuntimeError, match="blh_backward failed"):
                torch.autograd.grad(loss, params)
            continue
        g2 = torch.autograd.grad(loss, params)
        torch.cuda.synchronize()
        assert torch.equal(g1, torch.cat([t.reshape(-1) for t in g2])), batch


def test_backward_refuses_a_workspace_whose_record_says_another_batch(native):
    """C ABI: blh_backward on a workspace whose last forward saved another batch size (or nothing usable: an eval
    forward overwrote it) returns BLH_ERR_INVALID_ARGUMENT instead of reading the buffers in a wrong layout."""
    from bilinear_amd import _native as N
    dev = _dev()
    net, _ = _op_net(dev, 1, 256)
    eng = net.engine
    ws = eng.workspace(4096)
    x = torch.randn(4096, 32, device=dev)
    grads = torch.empty_like(eng.params)
    dpred = torch.zeros(1024, 48, device=dev)
    st = N.current_stream()

    def backward(batch):
        drop = N.Dropout(None, 1, 0, 0, 0, 0)
        return N.lib().blh_backward(eng.ctx.handle, ctypes.byref(eng.layout.desc), st, N.ptr(eng.params), N.ptr(x),
                                    ctypes.byref(drop), N.ptr(ws), ws.numel(), N.ptr(dpred), N.ptr(grads), batch,
                                    ctypes.cast(None, N.GradReadyFn), None)
    eng.forward_train(x)                                           # saves 4096 rows (encode stage without Z0)
    assert backward(1024) == -1                                    # BLH_ERR_INVALID_ARGUMENT
    with torch.no_grad():
        net.eval()
        eng.forward_eval(x[:1024])                                 # overwrites the buffers: nothing left to back-propagate
        net.train()
    assert backward(1024) == -1
    eng.forward_train(x[:1024])
    assert backward(1024) == 0
    torch.cuda.synchronize()
