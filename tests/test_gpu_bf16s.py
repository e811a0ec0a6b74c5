"""gemm_dtype = "bf16s" (bf16 storage, BASELINE configs 3-5) on the MI355X: kernel level.

The bf16-storage GEMM reads bf16 operands straight from memory (LDS-DMA, transposing LDS reads
for the operands whose reduction index is the memory row) and accumulates in fp32, so its result
must equal the exact product of the bf16 operand VALUES to fp32 accumulation accuracy (2e-5 of
sum |a b|), in every operand layout, with ragged M / N / K, split reductions, both output types
and every epilogue."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _to_bf16_bits(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)


def _bits_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


CASES = [
    # M, N, K, a_kmajor, b_kmajor, splits
    (256, 256, 128, 0, 0, 1),         # forward layout, full tiles
    (4096, 1024, 1024, 0, 0, 1),      # hidden Linear forward at B = 4096
    (300, 1024, 64, 0, 0, 1),         # ragged M, one K tile
    (200, 1024, 1024, 0, 1, 1),       # dgrad
    (264, 264, 200, 0, 1, 1),         # dgrad, ragged everything (K % 64 != 0)
    (1024, 1024, 4096, 1, 1, 4),      # wgrad, split over the batch
    (1024, 1024, 1000, 1, 1, 2),      # wgrad, ragged reduction
    (136, 200, 328, 1, 1, 1),         # wgrad, ragged M / N
    (64, 1024, 48, 0, 1, 1),          # K = 48 (one ragged tile)
    # width 2048 (BASELINE configs[4]) and the large-batch tile shapes (M >= 8192 selects the
    # 256 x 256 kernel where one exists)
    (512, 2048, 2048, 0, 0, 1),       # hidden Linear forward, W = 2048
    (384, 2048, 2048, 0, 1, 1),       # dgrad, W = 2048, ragged M
    (2048, 2048, 1536, 1, 1, 3),      # wgrad, W = 2048, split over the batch
    (8192, 1024, 1024, 0, 0, 1),      # forward at M = 8192
    (8192, 1024, 1024, 0, 1, 1),      # dgrad at M = 8192
    (8200, 2048, 2048, 0, 0, 1),      # forward, W = 2048, ragged last row tile
    (8320, 1024, 1024, 0, 1, 1),      # dgrad, ragged last row tile (M % 256 = 128)
    (1024, 1024, 16384, 1, 1, 4),     # wgrad at B = 16384
    (2048, 2048, 16384, 1, 1, 1),     # wgrad, W = 2048, B = 16384, one slab
    # the 128 x 256 kernel (gemm_bf16s_128x256.h) picks up M = 8192 at N = 1024 above; its ragged edges:
    (8300, 1024, 320, 0, 0, 1),       # ragged last row tile (108 rows), 5 K tiles (K % 128 = 64)
    (8250, 1024, 128, 0, 1, 1),       # dgrad, two K tiles: the loop body never runs
    (7300, 1024, 192, 0, 0, 1),       # three K tiles: one trip through the loop
    (3600, 2048, 448, 0, 1, 1),       # N = 2048: 29 x 8 tiles, ragged last row tile (16 rows)
]


@pytest.mark.parametrize("out_bf16", [0, 1])
@pytest.mark.parametrize("M,N,K,ak,bk,splits", CASES)
def test_gemm_bf16s_layouts(native, M, N, K, ak, bk, splits, out_bf16, force_tile=0):
    if out_bf16 and (splits > 1 or (ak and bk)):
        pytest.skip("weight-gradient outputs (slabs) are fp32")
    if force_tile:
        assert native.blh_gemm_bf16s_force_tile(force_tile) == 0
    dev = _dev()
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    A = rng.standard_normal((K, M) if ak else (M, K)).astype(np.float32)
    B = rng.standard_normal((K, N) if bk else (N, K)).astype(np.float32)
    A[0] += 3.0                                        # asymmetric operands
    Ab, Bb = _to_bf16_bits(A), _to_bf16_bits(B)
    A64, B64 = _bits_to_f32(Ab).astype(np.float64), _bits_to_f32(Bb).astype(np.float64)
    ref = (A64.T if ak else A64) @ (B64 if bk else B64.T)
    mag = np.abs(A64.T if ak else A64) @ np.abs(B64 if bk else B64.T)
    a = torch.from_numpy(Ab.view(np.int16)).to(dev)
    b = torch.from_numpy(Bb.view(np.int16)).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if out_bf16:
        c = torch.full((M, N), -1, dtype=torch.int16, device=dev)
    else:
        c = torch.full((splits, M, N), float("nan"), device=dev)
    try:
        rc = native.blh_gemm_bf16s(st, a.data_ptr(), A.shape[1], ak, b.data_ptr(), B.shape[1], bk,
                                   c.data_ptr(), N, out_bf16, M, N, K, splits, None, None, 0, None)
        torch.cuda.synchronize()
    finally:
        if force_tile:
            native.blh_gemm_bf16s_force_tile(-1)
    assert rc == 0, native.blh_status_string(rc)
    if out_bf16:
        got = _bits_to_f32(c.cpu().numpy().view(np.uint16)).astype(np.float64)
        # one rounding to bf16 on top of the fp32 accumulation
        assert (np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 2e-5 * mag).all()
    else:
        got = c.cpu().numpy().astype(np.float64).sum(axis=0)
        err = np.abs(got - ref) / mag
        assert err.max() <= 2e-5, err.max()


@pytest.mark.parametrize("M,N,K,ak,bk,splits,out_bf16", [
    (4096, 1024, 1024, 0, 0, 1, 1),   # shapes the dispatcher gives to other kernels, forced onto 128 x 256 tiles
    (300, 512, 192, 0, 1, 1, 0),      # three row tiles (the last one 44 rows), fp32 out
    (1024, 1024, 4096, 1, 1, 4, 0),   # weight gradient: both operands through the transposing read, 4 slabs
    (384, 768, 1088, 1, 1, 1, 0),     # weight gradient, 17 K tiles, one slab
    (1024, 1024, 8192, 1, 1, 8, 0),   # weight gradient, 8 slabs (the XCD-per-slab map)
])
def test_gemm_bf16s_128x256_forced(native, M, N, K, ak, bk, splits, out_bf16):
    """Every operand layout of gemm_bf16s_128x256_kernel, including those the dispatcher does not route to it."""
    assert native.blh_gemm_bf16s_force_tile(384) == 0
    try:
        assert native.blh_gemm_bf16s_tile(M, N, K, ak, bk, out_bf16, splits) == 128
        assert native.blh_gemm_bf16s_tile_cols(M, N, K, ak, bk, out_bf16, splits) == 256
    finally:
        native.blh_gemm_bf16s_force_tile(-1)
    test_gemm_bf16s_layouts(native, M, N, K, ak, bk, splits, out_bf16, force_tile=384)


@pytest.mark.parametrize("M,N,K,items,splits,pad", [
    (1024, 1024, 8192, 4, 4, 0),     # a group of four W = 1024 stages (the data-parallel hook path)
    (1024, 1024, 4096, 8, 2, 512),   # eight stages x two slabs, item strides larger than the tensors
    (512, 768, 2048, 10, 4, 0),      # non-square output, items x slabs x tiles = 240 workgroups
    (2048, 2048, 4096, 4, 1, 0),     # W = 2048: one slab, results written in place (no slabs)
])
def test_gemm_bf16s_batched_weight_gradients(native, M, N, K, items, splits, pad):
    """blh_gemm_bf16s_batched: `items` weight gradients dW_i = dZ_i^T A_i (both operands with the
    reduction index as the memory row) in ONE launch of the 256 x 256 kernel, item strides on every
    operand, slabs over the reduction; each item against the exact product of its bf16 operands."""
    dev = _dev()
    rng = np.random.RandomState(items + splits)
    sa, sb = K * M + pad, K * N + pad          # elements between items (the step: one [B, W] tensor apart)
    A = rng.standard_normal((items, sa)).astype(np.float32)
    B = rng.standard_normal((items, sb)).astype(np.float32)
    Ab, Bb = _to_bf16_bits(A), _to_bf16_bits(B)
    a = torch.from_numpy(Ab.view(np.int16)).to(dev)
    b = torch.from_numpy(Bb.view(np.int16)).to(dev)
    sc = splits * M * N + (64 if splits > 1 else 192)
    c = torch.full((items, sc), float("nan"), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = native.blh_gemm_bf16s_batched(st, a.data_ptr(), M, 1, sa, b.data_ptr(), N, 1, sb, c.data_ptr(), N, sc,
                                       M, N, K, items, splits)
    assert rc == 0, native.blh_status_string(rc)
    torch.cuda.synchronize()
    got_all = c.cpu().numpy().astype(np.float64)
    for i in range(items):
        A64 = _bits_to_f32(Ab[i, :K * M]).astype(np.float64).reshape(K, M)
        B64 = _bits_to_f32(Bb[i, :K * N]).astype(np.float64).reshape(K, N)
        ref = A64.T @ B64
        mag = np.abs(A64).T @ np.abs(B64)
        got = got_all[i, :splits * M * N].reshape(splits, M, N).sum(axis=0)
        err = np.abs(got - ref) / mag
        assert err.max() <= 2e-5, (i, err.max())
        assert np.isnan(got_all[i, splits * M * N:]).all(), "wrote past the item's slabs"
    # too few workgroups for the 256 x 256 kernel: refused, not silently run on another kernel
    rc = native.blh_gemm_bf16s_batched(st, a.data_ptr(), M, 1, sa, b.data_ptr(), N, 1, sb, c.data_ptr(), N, sc,
                                       256, 256, K, 2, 1)
    assert rc != 0


@pytest.fixture
def forced_tile(native, request):
    tile = request.param
    assert native.blh_gemm_bf16s_force_tile(tile) == 0
    yield tile
    native.blh_gemm_bf16s_force_tile(-1)


@pytest.mark.parametrize("M,N,K,forced_tile,rows", [
    (392, 384, 192, 0, 128),      # the 128 x 128 kernel
    (8392, 768, 256, 256, 256),   # the 256 x 256 kernel: 33 x 3 tiles, a ragged last row tile of 200 rows
    (8392, 768, 256, 384, 128),   # the 128 x 256 kernel: 66 x 3 tiles, the last one 72 rows
    (9800, 768, 256, 0, 128),     # the shape the dispatcher itself gives to the 128 x 256 kernel (77 x 3 tiles)
], indirect=["forced_tile"])
def test_gemm_bf16s_epilogues(native, M, N, K, forced_tile, rows):
    """Bias + BatchNorm tile partials and the skip-gradient addend on each of the three kernels."""
    dev = _dev()
    rng = np.random.RandomState(4)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (0.1 * rng.standard_normal((N, K))).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    add = rng.standard_normal((M, N)).astype(np.float32)
    Ab, Wb, addb = _to_bf16_bits(A), _to_bf16_bits(W), _to_bf16_bits(add)
    A64, W64 = _bits_to_f32(Ab).astype(np.float64), _bits_to_f32(Wb).astype(np.float64)
    a = torch.from_numpy(Ab.view(np.int16)).to(dev)
    w = torch.from_numpy(Wb.view(np.int16)).to(dev)
    bt = torch.from_numpy(bias).to(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # forward: + bias, bf16 out, BatchNorm tile partials of the un-rounded values
    z = torch.empty(M, N, dtype=torch.int16, device=dev)
    T = native.blh_gemm_bf16s_tile(M, N, K, 0, 0, 1, 1)          # rows per statistics tile
    assert T == rows
    assert native.blh_gemm_bf16s_tile_cols(M, N, K, 0, 0, 1, 1) == (128 if (forced_tile == 0 and M < 1000) else 256)
    part = torch.empty((M + T - 1) // T, 2, N, device=dev)
    assert native.blh_gemm_bf16s(st, a.data_ptr(), K, 0, w.data_ptr(), K, 0, z.data_ptr(), N, 1, M, N, K,
                                 1, bt.data_ptr(), None, 0, part.data_ptr()) == 0
    torch.cuda.synchronize()
    ref = A64 @ W64.T + bias
    got = _bits_to_f32(z.cpu().numpy().view(np.uint16)).astype(np.float64)
    assert (np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 1e-4).all()
    p = part.cpu().numpy().astype(np.float64)
    for t in range(p.shape[0]):
        rows = ref[t * T:(t + 1) * T]
        assert np.abs(p[t, 0] - rows.mean(0)).max() <= 1e-5 * (1 + np.abs(rows).max())
        m2 = ((rows - rows.mean(0)) ** 2).sum(0)
        assert np.abs(p[t, 1] - m2).max() <= 1e-4 * m2.max()
    # dgrad + skip gradient: C = A * Wk + addend (bf16), may alias the addend buffer
    Wk = (0.1 * rng.standard_normal((K, N))).astype(np.float32)
    Wkb = _to_bf16_bits(Wk)
    wk = torch.from_numpy(Wkb.view(np.int16)).to(dev)
    g = torch.from_numpy(addb.view(np.int16)).to(dev)
    assert native.blh_gemm_bf16s(st, a.data_ptr(), K, 0, wk.data_ptr(), N, 1, g.data_ptr(), N, 1, M, N, K,
                                 1, None, g.data_ptr(), N, None) == 0
    torch.cuda.synchronize()
    ref = A64 @ _bits_to_f32(Wkb).astype(np.float64) + _bits_to_f32(addb).astype(np.float64)
    got = _bits_to_f32(g.cpu().numpy().view(np.uint16)).astype(np.float64)
    assert (np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 1e-4).all()


def test_cast_round_trip(native):
    dev = _dev()
    rng = np.random.RandomState(1)
    x = (rng.standard_normal(4096 * 33) * np.exp(3 * rng.standard_normal(4096 * 33))).astype(np.float32)
    x[:4] = [np.inf, -np.inf, np.nan, 0.0]
    xt = torch.from_numpy(x).to(dev)
    b = torch.empty(x.size, dtype=torch.int16, device=dev)
    y = torch.empty(x.size, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert native.blh_cast_f32_to_bf16(st, xt.data_ptr(), b.data_ptr(), x.size) == 0
    assert native.blh_cast_bf16_to_f32(st, b.data_ptr(), y.data_ptr(), x.size) == 0
    torch.cuda.synchronize()
    want = _bits_to_f32(_to_bf16_bits(x[4:]))
    assert np.array_equal(y.cpu().numpy()[4:], want)
    head = y.cpu().numpy()[:4]
    assert head[0] == np.inf and head[1] == -np.inf and np.isnan(head[2]) and head[3] == 0.0


def test_gemm_bf16s_speed(native):
    """Not a pass/fail on speed: prints the achieved TFLOP/s of the three contractions at the
    config-3 shape (B = 16384, W = 1024) for the record (-s)."""
    dev = _dev()
    M, W = 16384, 1024
    a = torch.randn(M, W, device=dev).to(torch.bfloat16)
    w = (torch.randn(W, W, device=dev) * 0.03).to(torch.bfloat16)
    bias = torch.randn(W, device=dev)
    z = torch.empty(M, W, dtype=torch.bfloat16, device=dev)
    part = torch.empty(M // 128, 2, W, device=dev)
    slabs = torch.empty(4, W, W, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fwd():
        return native.blh_gemm_bf16s(st, a.data_ptr(), W, 0, w.data_ptr(), W, 0, z.data_ptr(), W, 1, M, W, W, 1,
                                     bias.data_ptr(), None, 0, part.data_ptr())

    def dgrad():
        return native.blh_gemm_bf16s(st, a.data_ptr(), W, 0, w.data_ptr(), W, 1, z.data_ptr(), W, 1, M, W, W, 1,
                                     None, None, 0, None)

    def wgrad():
        return native.blh_gemm_bf16s(st, a.data_ptr(), W, 1, z.data_ptr(), W, 1, slabs.data_ptr(), W, 0, W, W, M, 4,
                                     None, None, 0, None)

    for name, fn in (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)):
        for _ in range(20):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 100
        print("bf16s %s M=%d W=%d: %.1f us, %.0f TFLOP/s" % (name, M, W, ms * 1e3, 2.0 * M * W * W / ms / 1e9))


# ----------------------------------------------------------------------------
# network level: gemm_dtype = "bf16s" against the oracle run with the same storage rounding
# ----------------------------------------------------------------------------
def _bf16s_net(st, nb, width, dev):
    import bilinear_amd
    net = bilinear_amd.BilinearUnit(nb, width, gemm_dtype="bf16s")
    sd = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(np.array(st[k])).reshape(sd[k].shape) for k in sd})
    net = net.to(dev).train()
    opt = bilinear_amd.Adam(net.parameters(), lr=1e-3, module=net)
    return net, opt


# (2, 1024, 4100) and (1, 1024, 37): ragged batches — the last batch of an epoch of the reference's DataLoader
# (/root/reference/train_bilinear.py:33-43, drop_last unset); not multiples of 8: the weight-gradient GEMMs take the
# 8-row groups, wgrad_tail_h_kernel the rows that are left (r06; such batches used to be refused)
@pytest.mark.parametrize("nb,width,batch", [(1, 256, 512), (2, 1024, 640), (4, 1024, 2048), (2, 1024, 4100), (1, 1024, 37)])
def test_bf16s_network_against_same_rounding_oracle(nb, width, batch):
    """Forward, loss and every gradient of the bf16-storage path against the NumPy oracle rounding
    at the same places (operands and stored tensors), fp64 accumulation.  Two correct bf16 runs
    drift apart (an fp32-accumulation-order difference of 1e-6 moves a stored value across a bf16
    rounding boundary, 2^-8 relative, for a fraction of the elements), so the tolerances are
    bf16's, as for the round-1 mixed mode; the bit-level check of the kernel is
    test_gemm_bf16s_layouts.  ReLU gates within 2e-2 of zero are dropped from the masks (the gate
    decision must not depend on that noise)."""
    from golden_util import is_prebn_bias, safe_masks
    from oracle import numpy_oracle as O
    dev = _dev()
    st = O.init_state(300 + nb, nb, width)
    rng = np.random.RandomState(nb)
    for k in st:
        if k.endswith(".1.weight"):
            st[k] = (1.0 + 0.2 * rng.standard_normal(st[k].shape)).astype(np.float32)
        if k.endswith(".1.bias"):
            st[k] = (0.1 * rng.standard_normal(st[k].shape)).astype(np.float32)
    x, t = O.synthetic_batch(5, batch)
    masks = safe_masks(st, x, O.random_masks(9, batch, nb, width), rounding="bf16s", thr=2e-2)
    net, opt = _bf16s_net(st, nb, width, dev)
    net.engine.set_dropout_masks(masks)
    xt, tt = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
    pred = net(xt)
    loss = torch.nn.functional.mse_loss(pred, tt)
    loss.backward()
    O.set_gemm_rounding("bf16s")
    try:
        s2 = {k: v.copy() for k, v in st.items()}
        rp, cache = O.forward(s2, x, masks, training=True, dtype=np.float64)
        rl, dp = O.mse_loss(rp, t.astype(np.float64))
        rg = O.backward(s2, cache, dp, dtype=np.float64)
    finally:
        O.set_gemm_rounding(None)
    got = pred.detach().cpu().numpy().astype(np.float64)
    prel = np.linalg.norm(got - rp) / np.linalg.norm(rp)
    worst = 0.0
    for k, p in net.named_parameters():
        if is_prebn_bias(k):
            continue
        g = p.grad.cpu().numpy().astype(np.float64)
        assert np.isfinite(g).all(), k
        rel = np.linalg.norm(g - rg[k]) / np.linalg.norm(rg[k])
        worst = max(worst, rel)
        assert rel <= 0.1, (k, rel)
    print("bf16s %dx%d B=%d: pred rel L2 %.2e, loss %.5f (oracle %.5f), worst grad rel L2 %.2e" % (
        nb, width, batch, prel, loss.item(), rl, worst))
    assert prel <= 1e-2
    assert abs(loss.item() - rl) <= 5e-3 * rl
    sd = net.state_dict()
    for k in sd:
        if k.endswith("running_mean") or k.endswith("running_var"):
            ref = s2[k]
            assert np.abs(sd[k].cpu().numpy() - ref).max() <= 2e-3 * (1 + np.abs(ref).max()), k


def test_bf16s_fused_step_learns_and_is_deterministic():
    import bilinear_amd
    dev = _dev()
    torch.manual_seed(0)
    net, opt, _, _ = bilinear_amd.load(dev, num_blocks=2, width=1024, gemm_dtype="bf16s")
    net.train()
    net.engine.seed = 7
    x, t = torch.randn(4096, 32, device=dev), torch.randn(4096, 48, device=dev)
    losses = []
    for _ in range(30):
        _, loss = net.train_step(opt, x, t, max_norm=1.0)
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < 0.9 * losses[0], losses
    # same seed / step / state -> bit-identical step
    torch.manual_seed(0)
    net2, opt2, _, _ = bilinear_amd.load(dev, num_blocks=2, width=1024, gemm_dtype="bf16s")
    net2.train()
    net2.engine.seed = 7
    l2 = [net2.train_step(opt2, x, t, max_norm=1.0)[1].item() for _ in range(3)]
    assert l2 == losses[:3]
    net.eval()
    with torch.no_grad():
        a, b = net(x), net(x)
    assert torch.equal(a, b) and torch.isfinite(a).all()


def test_bf16s_backward_schedules_are_bit_identical():
    """Two-stream backward of the bf16-storage path (weight gradients on the side stream, forked
    behind the data-gradient GEMM or behind bn_bwd_apply) only re-orders independent kernels:
    every gradient, moment and parameter is bit-equal to the single-stream order, through the
    fused step and through autograd (blh_backward)."""
    import bilinear_amd
    from bilinear_amd import _native as N
    dev = _dev()
    x, t = (torch.randn(8192, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(3)),
            torch.randn(8192, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(4)))
    out = {}
    for sched in ((True, 1), (True, 0), (True, 2), (False, 1)):
        torch.manual_seed(0)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=2, width=1024, gemm_dtype="bf16s")
        net.train()
        net.engine.ensure(dev)
        net.engine.seed = 11
        net.engine.set_two_stream(sched[0])
        net.engine.ctx.set_option(N.OPT_LATE_FORK, sched[1])
        for _ in range(2):
            pred, loss = net.train_step(opt, x, t, max_norm=1.0)
        # autograd path (blh_backward) on the third step
        opt.zero_grad()
        p3 = net(x)
        torch.nn.functional.mse_loss(p3, t).backward()
        torch.cuda.synchronize()
        out[sched] = (pred.clone(), net.engine.params.clone(), opt._exp_avg_sq.clone(),
                      net.engine.grads.clone(), p3.detach().clone())
    for other in ((True, 0), (True, 2), (False, 1)):
        for a, b, what in zip(out[(True, 1)], out[other], ("pred", "params", "exp_avg_sq", "grads", "pred3")):
            assert torch.equal(a, b), (other, what)
    assert torch.isfinite(out[(True, 1)][3]).all()


@pytest.mark.parametrize("nb,batch", [(2, 8192), (1, 16384)])
def test_bn_backward_reductions_in_the_dgrad_epilogue_match_the_streaming_kernel(nb, batch):
    """SURVEY K9 (round 4): the data-gradient GEMMs whose output only feeds the BatchNorm backward of the stage
    below form that stage's gated gradient dY' and the (dY' z, dY') column sums in their epilogue
    (EPI_BN_BWD, the 256 x 256 and 128 x 256 kernels), and the stage below skips bn_bwd_reduce_h2.  The
    gated values are the same bf16 numbers the streaming kernel forms; only the order of the column sums
    differs, so every gradient agrees with the BLH_NO_K9=1 run to summation rounding (plus the few bf16
    roundings of dZ that a 1e-7 change of the column sums moves)."""
    import os

    import bilinear_amd
    dev = _dev()
    x, t = (torch.randn(batch, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(3)),
            torch.randn(batch, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(4)))
    out = {}
    for k9 in (True, False):
        if not k9:
            os.environ["BLH_NO_K9"] = "1"
        try:
            torch.manual_seed(0)
            net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=1024, gemm_dtype="bf16s")
            net.train()
            net.engine.ensure(dev)
            net.engine.seed = 11
            opt.zero_grad()
            pred = net(x)
            torch.nn.functional.mse_loss(pred, t).backward()
            torch.cuda.synchronize()
            out[k9] = (pred.detach().clone(), net.engine.grads.clone(),
                       {k: p.grad.detach().clone() for k, p in net.named_parameters()})
        finally:
            os.environ.pop("BLH_NO_K9", None)
    assert torch.equal(out[True][0], out[False][0])              # the forward is untouched
    assert not torch.equal(out[True][1], out[False][1])          # ... and the backward really took another path
    for k in out[True][2]:
        a, b = out[True][2][k].double(), out[False][2][k].double()
        if k.endswith(".0.bias") and not k.startswith("decode"):
            continue                                             # pre-BatchNorm biases: rounding noise
        rel = float((a - b).norm() / b.norm())
        assert rel <= 2e-3, (k, rel)


@pytest.mark.parametrize("nb,batch", [(2, 4096), (4, 16384), (4, 2048)])
def test_gradient_norm_partials_from_the_batched_slab_sum(nb, batch):
    """r06: when the hidden weight gradients of the bf16-storage fused step leave through ONE batched slab sum, that
    kernel also takes the partials of the gradient norm (clip_grad_norm_, /root/reference/train_bilinear.py:81) — of
    its own output and of every other range of the arena — and the step runs no pass of its own over the arena.  The
    gradients are the same bits as with BLH_NO_SUMSQ_FOLD=1; the norm is the same sum in another order (fp64 partials):
    equal to 1e-6, and so is everything Adam makes of it."""
    import os

    import bilinear_amd
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(5)
    x, t = torch.randn(batch, 32, device=dev, generator=g), torch.randn(batch, 48, device=dev, generator=g)
    out = {}
    for fold in (True, False):
        if not fold:
            os.environ["BLH_NO_SUMSQ_FOLD"] = "1"
        try:
            torch.manual_seed(0)
            net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=1024, gemm_dtype="bf16s")
            net.train()
            net.engine.seed = 3
            stats = []
            for _ in range(2):
                net.train_step(opt, x, t, max_norm=1.0)
                stats.append(opt.last_grad_norm_stats.clone())
            torch.cuda.synchronize()
            out[fold] = (torch.stack(stats).cpu().double(), net.engine.grads.clone(), net.engine.params.clone())
        finally:
            os.environ.pop("BLH_NO_SUMSQ_FOLD", None)
    a, b = out[True][0], out[False][0]
    assert (a[:, 0] > 0).all() and ((a - b).abs() <= 1e-6 * b.abs()).all(), (a, b)
    # the first step's clipped gradient = the same raw gradient times coefficients that agree to 1e-6; the second
    # step starts from parameters that agree to Adam's rounding
    rel = float((out[True][2] - out[False][2]).double().norm() / out[False][2].double().norm())
    assert rel <= 1e-6, rel


def test_persistent_shadow_is_bit_identical_and_invalidated_by_parameter_writes():
    """BLH_OPT_PERSISTENT_SHADOW (fused Adam -> bf16 weight image + the decode weight's K-major image of the
    one-pass decode; the host layer's default for bf16 storage since round 6): the same steps with and
    without it are bit-identical — through plain fused steps, a load_state_dict between two steps (the module drops the
    image), an in-place write to a Parameter (caught through its version counter), an unfused optimizer.step, and a
    captured graph that holds no arena re-cast."""
    import bilinear_amd
    from bilinear_amd import _native as N
    dev = _dev()
    x, t = (torch.randn(2048, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(3)),
            torch.randn(2048, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(4)))

    def run(persistent, captured=False):
        torch.manual_seed(0)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=1, width=1024, gemm_dtype="bf16s")
        net.train()
        net.engine.ensure(dev)
        net.engine.seed = 5
        step = None
        assert net.engine.ctx.get_option(N.OPT_PERSISTENT_SHADOW) == 1          # the default for bf16 storage
        if captured:
            step = bilinear_amd.CapturedTrainStep(net, opt, 2048, persistent_shadow=persistent)
        else:
            net.engine.set_persistent_shadow(persistent)
        losses = []
        for i in range(7):
            if i == 5:          # an in-place write the engine is not told about: seen through the version counter
                with torch.no_grad():
                    net.decode.weight.mul_(0.75)
                    net.bilinear[0][0][0].weight.add_(0.01)
            if i == 2:          # a checkpoint load between two steps: scaled weights
                sd = {k: (v * 0.5 if v.dtype.is_floating_point and k.endswith("0.weight") else v)
                      for k, v in net.state_dict().items()}
                net.load_state_dict(sd)
            if i == 4 and not captured:      # the drop-in path in between: forward, backward, optimizer.step
                opt.zero_grad()
                torch.nn.functional.mse_loss(net(x), t).backward()
                opt.step()
            pred, loss = step(x, t) if captured else net.train_step(opt, x, t, max_norm=1.0)
            losses.append(loss.item())
        torch.cuda.synchronize()
        return losses, net.engine.params.clone(), pred.clone()

    for captured in (False, True):
        a = run(False, captured)
        b = run(True, captured)
        assert a[0] == b[0], (captured, a[0], b[0])
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), captured


@pytest.mark.parametrize("nb,width,batch", [(2, 1024, 8192), (1, 1024, 4104), (1, 2048, 2048)])
def test_bf16s_encode_stage_without_z0_matches_the_materialised_path(nb, width, batch):
    """Round 5, bf16 storage: the encode stage runs without its pre-BatchNorm tensor (encode_f32.hip, bf16 form: x and
    the W0 shadow are bf16, the batch statistics come from the sums of x, z is rounded to bf16 before BatchNorm
    normalises it, A0 and the keep-and-gate bits go out in the bf16 layouts; the backward needs dA0, the bits and x).
    Against BLH_NO_ENCODE_FUSE=1 (bf16 GEMM -> bn_fwd_finalize -> bn_apply_h2, streaming BatchNorm backward,
    weight-gradient GEMM) on the same explicit masks: the two contract the same bf16 operands in another order, so a
    few z round to the neighbouring bf16 value — predictions agree to 2e-3, gradients to the bf16 path's 2e-2, running
    statistics to 1e-5 (they come from the unrounded z in both)."""
    import os

    import bilinear_amd
    dev = _dev()
    x = torch.randn(batch, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    t = torch.randn(batch, 48, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    g = torch.Generator(device=dev).manual_seed(9)
    masks = [(torch.rand(batch, width, device=dev, generator=g) < 0.5).to(torch.uint8) for _ in range(1 + 2 * nb)]
    out = {}
    for fused in (True, False):
        if not fused:
            os.environ["BLH_NO_ENCODE_FUSE"] = "1"
        try:
            torch.manual_seed(0)
            net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width, gemm_dtype="bf16s")
            net.train()
            net.engine.ensure(dev)
            net.engine.set_dropout_masks(masks)
            opt.zero_grad()
            pred = net(x)
            torch.nn.functional.mse_loss(pred, t).backward()
            torch.cuda.synchronize()
            first = (pred.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()},
                     net.engine.bn_running.clone(), net.engine.bn_nbt.clone())
            for _ in range(2):
                p2, loss = net.train_step(opt, x, t, max_norm=1.0)
            torch.cuda.synchronize()
            out[fused] = first + (p2.clone(), loss.clone(), net.engine.bn_running.clone())
        finally:
            os.environ.pop("BLH_NO_ENCODE_FUSE", None)

    def rel(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))

    a, b = out[True], out[False]
    assert rel(a[0], b[0]) <= 2e-3, ("pred", rel(a[0], b[0]))
    for k in a[1]:
        if k.endswith(".0.bias") and not k.startswith("decode"):
            continue
        assert rel(a[1][k], b[1][k]) <= 2e-2, (k, rel(a[1][k], b[1][k]))
    assert rel(a[2][0], b[2][0]) <= 1e-5 and torch.equal(a[3], b[3])        # stage 0's running statistics
    assert abs(float(a[5]) - float(b[5])) <= 2e-3 * abs(float(b[5]))
    assert torch.isfinite(a[4]).all() and rel(a[6][0], b[6][0]) <= 1e-4


def test_weight_image_kept_by_adam_survives_a_change_of_the_batch_size():
    """bf16 storage keeps the bf16 weight image (and the one-pass decode's K-major decode weight) that the fused step's
    Adam kernel writes, and the next step skips the re-cast (default since round 6).  The reference's loader ends an epoch
    with a smaller batch (/root/reference/train_bilinear.py:33-43) and the engine runs it in the SAME workspace: the
    images must sit at offsets that do not depend on the batch.  Round 6 found the decode image behind the batch-sized
    buffers — after a batch-size change the decode read a stale or never-written image and the parameters drifted by
    3e-2 in seven steps.  With the image and without it: bit-identical, through 4096 / 2048 / 1236-row steps."""
    import bilinear_amd
    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(1)
    xs = {b: (torch.randn(b, 32, device=dev, generator=g), torch.randn(b, 48, device=dev, generator=g))
          for b in (4096, 2048, 1236)}
    out = {}
    for keep in (True, False):
        torch.manual_seed(0)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=2, width=1024, gemm_dtype="bf16s")
        net.train()
        net.engine.seed = 5
        net.engine.ensure(dev)
        net.engine.set_persistent_shadow(keep)
        losses = []
        for b in (4096, 4096, 2048, 2048, 4096, 1236, 4096):
            x, t = xs[b]
            losses.append(float(net.train_step(opt, x, t, max_norm=1.0)[1].item()))
        torch.cuda.synchronize()
        out[keep] = (net.engine.params.clone(), net.engine.bn_running.clone(), losses)
    assert out[True][2] == out[False][2], (out[True][2], out[False][2])
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
