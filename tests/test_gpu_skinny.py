"""The skinny (HBM-bound) projections of the lifter, exactly as the training step launches them
(blh_skinny_* entry points), against fp64 NumPy: encode forward with BatchNorm partials, decode
forward fused with the MSE loss, decode backward (weight + data gradient), encode weight gradient.
Reference call-sites: /root/reference/model/bilinear.py:22,29,39; train_bilinear.py:78-79."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _close(got, ref, rtol, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    rms = np.sqrt((ref ** 2).mean())
    err = np.abs(got - ref)
    bound = rtol * (np.abs(ref) + rms)
    assert (err <= bound).all(), "%s: max excess %.3e" % (what, (err - bound).max())


@pytest.mark.parametrize("batch,width", [(4096, 1024), (4100, 1024), (300, 256), (16384, 2048), (40000, 512)])
def test_decode_forward_fused_mse(native, batch, width):
    dev = _dev()
    rng = np.random.RandomState(batch + width)
    OF = 48
    A = rng.standard_normal((batch, width)).astype(np.float32)
    A[:, 0] += 2.0
    Wd = (rng.standard_normal((OF, width)) * 0.05).astype(np.float32)
    bd = rng.standard_normal(OF).astype(np.float32)
    t = rng.standard_normal((batch, OF)).astype(np.float32)
    a, w, b, tt = (torch.from_numpy(v).to(dev) for v in (A, Wd, bd, t))
    pred = torch.full((batch, OF), float("nan"), device=dev)
    dpred = torch.full((batch, OF), float("nan"), device=dev)
    loss = torch.zeros((), device=dev)
    wsb = native.blh_skinny_workspace_bytes(batch, width, 32, OF)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = native.blh_skinny_decode_fwd_mse(st, a.data_ptr(), w.data_ptr(), b.data_ptr(), tt.data_ptr(),
                                          pred.data_ptr(), dpred.data_ptr(), loss.data_ptr(), ws.data_ptr(),
                                          wsb, batch, width, OF)
    assert rc == 0, native.blh_status_string(rc)
    torch.cuda.synchronize()
    ref = A.astype(np.float64) @ Wd.T.astype(np.float64) + bd
    _close(pred.cpu().numpy(), ref, 2e-5, "pred")
    d = ref - t
    _close(dpred.cpu().numpy(), 2 * d / d.size, 1e-4, "dpred")
    assert abs(loss.item() - (d ** 2).mean()) <= 1e-5 * (d ** 2).mean()


@pytest.mark.parametrize("batch", [4096, 4100, 640])
def test_decode_backward_and_encode_kernels(native, batch):
    dev = _dev()
    rng = np.random.RandomState(batch)
    W, OF, IF = 1024, 48, 32
    A = rng.standard_normal((batch, W)).astype(np.float32)
    Wd = (rng.standard_normal((OF, W)) * 0.05).astype(np.float32)
    dP = (rng.standard_normal((batch, OF)) * 1e-3).astype(np.float32)
    x = rng.standard_normal((batch, IF)).astype(np.float32)
    W0 = (rng.standard_normal((W, IF)) * 0.25).astype(np.float32)
    b0 = rng.standard_normal(W).astype(np.float32)
    dZ = (rng.standard_normal((batch, W)) * 1e-3).astype(np.float32)
    a, wd, dp, xt, w0, bt, dz = (torch.from_numpy(v).to(dev) for v in (A, Wd, dP, x, W0, b0, dZ))
    wsb = native.blh_skinny_workspace_bytes(batch, W, IF, OF)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    # decode backward
    dWd = torch.full((OF, W), float("nan"), device=dev)
    dA = torch.full((batch, W), float("nan"), device=dev)
    assert native.blh_skinny_decode_bwd(st, dp.data_ptr(), a.data_ptr(), wd.data_ptr(), dWd.data_ptr(),
                                        dA.data_ptr(), ws.data_ptr(), wsb, batch, W, OF) == 0
    torch.cuda.synchronize()
    _close(dWd.cpu().numpy(), dP.T.astype(np.float64) @ A.astype(np.float64), 3e-5, "dWd")
    _close(dA.cpu().numpy(), dP.astype(np.float64) @ Wd.astype(np.float64), 3e-5, "dA")
    # encode forward + BatchNorm tile partials
    Z = torch.full((batch, W), float("nan"), device=dev)
    part = torch.zeros((batch + 63) // 64, 2, W, device=dev)
    rows = ctypes.c_int32(0)
    assert native.blh_skinny_encode_fwd(st, xt.data_ptr(), w0.data_ptr(), bt.data_ptr(), Z.data_ptr(),
                                        part.data_ptr(), ctypes.byref(rows), batch, W, IF) == 0
    torch.cuda.synchronize()
    zr = x.astype(np.float64) @ W0.T.astype(np.float64) + b0
    _close(Z.cpu().numpy(), zr, 2e-5, "Z")
    tr = rows.value
    assert tr in (64, 128)
    p = part.cpu().numpy().astype(np.float64)
    for ti in range((batch + tr - 1) // tr):
        blk = zr[ti * tr:(ti + 1) * tr]
        assert np.abs(p[ti, 0] - blk.mean(0)).max() <= 1e-5 * (1 + np.abs(blk).max())
        m2 = ((blk - blk.mean(0)) ** 2).sum(0)
        assert np.abs(p[ti, 1] - m2).max() <= 1e-4 * m2.max() + 1e-6
    # encode weight gradient
    dW0 = torch.full((W, IF), float("nan"), device=dev)
    assert native.blh_skinny_encode_wgrad(st, dz.data_ptr(), xt.data_ptr(), dW0.data_ptr(), ws.data_ptr(), wsb,
                                          batch, W, IF) == 0
    torch.cuda.synchronize()
    _close(dW0.cpu().numpy(), dZ.T.astype(np.float64) @ x.astype(np.float64), 3e-5, "dW0")


@pytest.mark.parametrize("batch,width", [(4096, 1024), (4100, 1024), (16384, 1024), (37, 512), (2048, 512)])
def test_decode_one_pass_forward_mse_and_data_gradient(native, batch, width):
    """decode_fused_kernel (round 5): prediction, MSE, dpred, the loss AND the decode data gradient dA = dP Wd from
    one read of the last activation — what the fused fp32 step launches (/root/reference/model/bilinear.py:29,39;
    train_bilinear.py:78-79).  Against fp64 NumPy; ragged last row block (4100, 37); rows beyond the batch untouched."""
    dev = _dev()
    rng = np.random.RandomState(batch + width + 5)
    OF = 48
    A = rng.standard_normal((batch, width)).astype(np.float32)
    A[:, 0] += 2.0
    Wd = (rng.standard_normal((OF, width)) * 0.05).astype(np.float32)
    bd = rng.standard_normal(OF).astype(np.float32)
    t = rng.standard_normal((batch, OF)).astype(np.float32)
    a, w, b, tt = (torch.from_numpy(v).to(dev) for v in (A, Wd, bd, t))
    pred = torch.full((batch + 3, OF), float("nan"), device=dev)
    dpred = torch.full((batch + 3, OF), float("nan"), device=dev)
    dA = torch.full((batch + 3, width), float("nan"), device=dev)
    loss = torch.zeros((), device=dev)
    wsb = native.blh_skinny_workspace_bytes(batch, width, 32, OF)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = native.blh_skinny_decode_fused(st, a.data_ptr(), w.data_ptr(), b.data_ptr(), tt.data_ptr(), pred.data_ptr(),
                                        dpred.data_ptr(), dA.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, batch,
                                        width, OF)
    assert rc == 0, native.blh_status_string(rc)
    torch.cuda.synchronize()
    ref = A.astype(np.float64) @ Wd.T.astype(np.float64) + bd
    _close(pred[:batch].cpu().numpy(), ref, 2e-5, "pred")
    d = ref - t
    dp = 2 * d / d.size
    _close(dpred[:batch].cpu().numpy(), dp, 1e-4, "dpred")
    assert abs(loss.item() - (d ** 2).mean()) <= 1e-5 * (d ** 2).mean()
    _close(dA[:batch].cpu().numpy(), dp @ Wd.astype(np.float64), 1e-4, "dA")
    assert torch.isnan(pred[batch:]).all() and torch.isnan(dpred[batch:]).all() and torch.isnan(dA[batch:]).all()
    # shapes the one-pass kernel does not serve are refused (the step then takes the two-kernel path)
    assert native.blh_skinny_decode_fused(st, a.data_ptr(), w.data_ptr(), b.data_ptr(), tt.data_ptr(), pred.data_ptr(),
                                          dpred.data_ptr(), dA.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, batch,
                                          width, 32) != 0


def _bf16_round(a):
    """fp32 -> nearest-even bf16, as fp32 values and as raw uint16 bit patterns."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    return (r << 16).astype(np.uint32).view(np.float32), r.astype(np.uint16)


@pytest.mark.parametrize("batch,width", [(4096, 1024), (4104, 1024), (16384, 2048), (40, 256)])
def test_decode_one_pass_bf16_storage(native, batch, width):
    """decode_fwd_mse_h_kernel<RT, FUSE> (round 5, what the bf16-storage step launches): prediction, MSE, dpred, loss and
    the decode data gradient dA = dP Wd (bf16) from one read of the bf16 activation, through the C entry point against
    fp64 NumPy on the same bf16-rounded operands (A, Wd; dP is rounded to bf16 before the second contraction, as the
    backward GEMM this replaces did).  (16384, 2048) is configs[4]'s per-GPU decode; 4104 and 40 leave a ragged last
    row block.  /root/reference/model/bilinear.py:29,39; train_bilinear.py:78-79."""
    dev = _dev()
    rng = np.random.RandomState(batch + width + 11)
    OF = 48
    A32, Abits = _bf16_round(rng.standard_normal((batch, width)).astype(np.float32))
    Wd = (rng.standard_normal((OF, width)) * 0.05).astype(np.float32)
    Wd32, _ = _bf16_round(Wd)
    bd = rng.standard_normal(OF).astype(np.float32)
    t = rng.standard_normal((batch, OF)).astype(np.float32)
    a = torch.from_numpy(Abits.view(np.int16)).to(dev)
    w, b, tt = (torch.from_numpy(v).to(dev) for v in (Wd, bd, t))
    pred = torch.full((batch + 3, OF), float("nan"), device=dev)
    dpred = torch.full((batch + 3, OF), float("nan"), device=dev)
    dA = torch.full((batch + 3, width), -1, dtype=torch.int16, device=dev)
    loss = torch.zeros((), device=dev)
    wsb = native.blh_skinny_decode_fused_bf16_workspace_bytes(batch, width, OF)
    assert wsb > 0
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = native.blh_skinny_decode_fused_bf16(st, a.data_ptr(), w.data_ptr(), b.data_ptr(), tt.data_ptr(), pred.data_ptr(),
                                             dpred.data_ptr(), dA.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, batch,
                                             width, OF)
    assert rc == 0, native.blh_status_string(rc)
    torch.cuda.synchronize()
    ref = A32.astype(np.float64) @ Wd32.T.astype(np.float64) + bd
    _close(pred[:batch].cpu().numpy(), ref, 2e-5, "pred")
    d = ref - t
    dp = 2 * d / d.size
    _close(dpred[:batch].cpu().numpy(), dp, 1e-4, "dpred")
    assert abs(loss.item() - (d ** 2).mean()) <= 1e-5 * (d ** 2).mean()
    # dA: the kernel's own fp32 dpred rounded to bf16, times the bf16 weight, fp32 accumulation, one bf16 rounding
    dph, _ = _bf16_round(dpred[:batch].cpu().numpy())
    want = dph.astype(np.float64) @ Wd32.astype(np.float64)
    got = (dA[:batch].cpu().numpy().view(np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2.0 ** -8 * scale * 1.01 + 1e-12, np.abs(got - want).max() / scale   # half a bf16 ulp
    rel = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert rel <= 3e-3, rel
    assert torch.isnan(pred[batch:]).all() and torch.isnan(dpred[batch:]).all() and (dA[batch:] == -1).all()
    assert native.blh_skinny_decode_fused_bf16(st, a.data_ptr(), w.data_ptr(), b.data_ptr(), tt.data_ptr(), pred.data_ptr(),
                                               dpred.data_ptr(), dA.data_ptr(), loss.data_ptr(), ws.data_ptr(), wsb, batch,
                                               width, 32) != 0


@pytest.mark.parametrize("batch,width", [(4096, 1024), (4100, 1024), (900, 256), (16384, 1024)])
def test_encode_stage_without_its_pre_batchnorm_tensor(native, batch, width):
    """encode_f32.hip (round 5) through its C entry points against fp64 NumPy: forward x -> A0 = 2 keep relu(BN(x W0^T +
    b0)) with the batch statistics derived from the sums of x (saved mean / invstd / scale / shift, running statistics
    with momentum 0.1 and the unbiased variance, counter), backward dA0 -> dW0, db0, dgamma, dbeta from dA0, the
    keep-AND-gate bits and x alone.  Explicit masks; elements whose pre-activation is within 1e-4 of the ReLU kink are
    masked out (the gate of such an element is decided by rounding).  /root/reference/model/bilinear.py:7-13,22,34."""
    from bilinear_amd import _native as N
    dev = _dev()
    rng = np.random.RandomState(batch + width)
    IF = 32
    x = rng.standard_normal((batch, IF)).astype(np.float32)
    x[:, 3] += 0.5
    W0 = (rng.standard_normal((width, IF)) * 0.25).astype(np.float32)
    b0 = (rng.standard_normal(width) * 0.1).astype(np.float32)
    gamma = (1.0 + 0.1 * rng.standard_normal(width)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(width)).astype(np.float32)
    dA = (rng.standard_normal((batch, width)) * 1e-3).astype(np.float32)
    z = x.astype(np.float64) @ W0.T.astype(np.float64) + b0
    mean, var = z.mean(0), z.var(0)
    invstd = 1.0 / np.sqrt(var + 1e-5)
    y = (z - mean) * invstd * gamma + beta
    keep = (rng.random_sample((batch, width)) < 0.5)
    keep &= np.abs(y) > 1e-4
    A_ref = 2.0 * keep * np.maximum(y, 0.0)
    xt, w0, bt, gt, bet, dat = (torch.from_numpy(v).to(dev) for v in (x, W0, b0, gamma, beta, dA))
    km = torch.from_numpy(keep.astype(np.uint8)).to(dev)
    rm, rv = torch.zeros(width, device=dev), torch.ones(width, device=dev)
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    saved = torch.full((4, width), float("nan"), device=dev)
    scratch = torch.empty(batch * width, device=dev)
    A = torch.full((batch + 1, width), float("nan"), device=dev)
    bits = torch.zeros(((batch + 7) // 8) * (width // 4), dtype=torch.int32, device=dev)
    drop = N.Dropout(km.data_ptr(), 0, 0, 0, 0, 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = native.blh_skinny_encode_fused_fwd(st, xt.data_ptr(), w0.data_ptr(), bt.data_ptr(), gt.data_ptr(), bet.data_ptr(),
                                            rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1, saved.data_ptr(),
                                            scratch.data_ptr(), A.data_ptr(), bits.data_ptr(), ctypes.byref(drop), batch,
                                            width, IF)
    assert rc == 0, native.blh_status_string(rc)
    torch.cuda.synchronize()
    _close(A[:batch].cpu().numpy(), A_ref, 2e-5, "A0")
    assert torch.isnan(A[batch:]).all() and int(nbt) == 1
    sv = saved.cpu().numpy().astype(np.float64)
    _close(sv[0], mean, 1e-6, "saved mean")
    _close(sv[1], invstd, 1e-6, "saved invstd")
    _close(sv[2], gamma * invstd, 1e-6, "scale")
    _close(rm.cpu().numpy(), 0.1 * mean, 1e-6, "running mean")
    _close(rv.cpu().numpy(), 0.9 + 0.1 * var * batch / (batch - 1), 1e-6, "running var")
    # backward (BatchNorm1d backward with batch statistics, /root/reference/model/bilinear.py:10)
    dY = 2.0 * keep * (y > 0) * dA.astype(np.float64)
    zhat = (z - mean) * invstd
    dgamma_ref, dbeta_ref = (dY * zhat).sum(0), dY.sum(0)
    dZ = gamma * invstd * (dY - dbeta_ref / batch - zhat * dgamma_ref / batch)
    dW0 = torch.full((width, IF), float("nan"), device=dev)
    db0, dg, db = (torch.full((width,), float("nan"), device=dev) for _ in range(3))
    rc = native.blh_skinny_encode_fused_bwd(st, dat.data_ptr(), xt.data_ptr(), w0.data_ptr(), bt.data_ptr(),
                                            saved.data_ptr(), bits.data_ptr(), scratch.data_ptr(), dW0.data_ptr(),
                                            db0.data_ptr(), dg.data_ptr(), db.data_ptr(), batch, width, IF)
    assert rc == 0, native.blh_status_string(rc)
    torch.cuda.synchronize()
    _close(dg.cpu().numpy(), dgamma_ref, 1e-4, "dgamma")
    _close(db.cpu().numpy(), dbeta_ref, 1e-4, "dbeta")
    _close(dW0.cpu().numpy(), dZ.T @ x.astype(np.float64), 1e-4, "dW0")
    assert np.abs(db0.cpu().numpy()).max() <= 1e-5 * max(1.0, np.abs(dZ).sum(0).max())      # sum of dZ: zero up to rounding
    # shapes the kernels do not serve are refused
    assert native.blh_skinny_encode_fused_fwd(st, xt.data_ptr(), w0.data_ptr(), bt.data_ptr(), gt.data_ptr(),
                                              bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1,
                                              saved.data_ptr(), scratch.data_ptr(), A.data_ptr(), bits.data_ptr(),
                                              ctypes.byref(drop), batch, width, 48) != 0


def _bf16_round(a):
    """fp32 array -> (values rounded to bf16 as fp32, their uint16 bit patterns)."""
    t = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16)
    return t.float().numpy(), t.view(torch.int16).numpy().view(np.uint16)


@pytest.mark.parametrize("batch,width", [(16384, 1024), (8192, 1024), (4100, 512), (2048, 2048), (1000, 512)])
def test_encode_stage_without_its_pre_batchnorm_tensor_bf16_storage(native, batch, width):
    """The bf16-storage form of encode_f32.hip (r06: z from ONE v_mfma_f32_16x16x32_bf16 per tile, the backward's
    dY'^T X on bf16 MFMAs over x^T left by the forward, keep-and-gate bits in bn_bf16.hip's [B/4][W/8] layout) through
    its C entry points against fp64 NumPy on the SAME stored values: x, W0 and dA0 are bf16, z is rounded to bf16 before
    BatchNorm normalises it (statistics from the un-rounded z), A0 is stored in bf16.  Elements within 2e-2 of the ReLU
    kink are masked out (one bf16 step of z decides their gate).  /root/reference/model/bilinear.py:7-13,22,34."""
    from bilinear_amd import _native as N
    dev = _dev()
    rng = np.random.RandomState(batch + width)
    IF = 32
    x, xb = _bf16_round(rng.standard_normal((batch, IF)))
    x[:, 3] += 0.5
    x, xb = _bf16_round(x)
    W0, W0b = _bf16_round(rng.standard_normal((width, IF)) * 0.25)
    b0 = (rng.standard_normal(width) * 0.1).astype(np.float32)
    gamma = (1.0 + 0.1 * rng.standard_normal(width)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(width)).astype(np.float32)
    dA, dAb = _bf16_round(rng.standard_normal((batch, width)) * 1e-3)
    z = x.astype(np.float64) @ W0.T.astype(np.float64) + b0
    mean, var = z.mean(0), z.var(0)
    invstd = 1.0 / np.sqrt(var + 1e-5)
    zr = _bf16_round(z)[0].astype(np.float64)                      # what BatchNorm normalises
    y = (zr - mean) * invstd * gamma + beta
    keep = (rng.random_sample((batch, width)) < 0.5)
    keep &= np.abs(y) > 2e-2
    A_ref = 2.0 * keep * np.maximum(y, 0.0)
    dt = lambda v: torch.from_numpy(v).to(dev)
    xt, w0 = dt(xb.view(np.int16)), dt(W0b.view(np.int16))
    bt, gt, bet, dat = dt(b0), dt(gamma), dt(beta), dt(dAb.view(np.int16))
    km = dt(keep.astype(np.uint8))
    rm, rv = torch.zeros(width, device=dev), torch.ones(width, device=dev)
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    saved = torch.full((4, width), float("nan"), device=dev)
    scratch = torch.empty(batch * width, dtype=torch.int16, device=dev)
    A = torch.full((batch + 1, width), -1, dtype=torch.int16, device=dev)
    bits = torch.zeros(((batch + 3) // 4) * (width // 8), dtype=torch.int32, device=dev)
    drop = N.Dropout(km.data_ptr(), 0, 0, 0, 0, 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = native.blh_skinny_encode_fused_fwd_bf16(st, xt.data_ptr(), w0.data_ptr(), bt.data_ptr(), gt.data_ptr(),
                                                 bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1,
                                                 saved.data_ptr(), scratch.data_ptr(), A.data_ptr(), bits.data_ptr(),
                                                 ctypes.byref(drop), batch, width, IF)
    assert rc == 0, native.blh_status_string(rc)
    torch.cuda.synchronize()
    got = A[:batch].view(torch.bfloat16).float().cpu().numpy()
    assert (A[batch:] == -1).all() and int(nbt) == 1               # nothing written past the batch
    # bf16 storage of A0: half an ulp (2^-9 relative) + the fp32 statistics
    # (a z whose fp32 sum lands on the other side of a bf16 rounding boundary moves by 2^-8 |z|)
    assert (np.abs(got - A_ref) <= 2.0 ** -7 * (np.abs(A_ref) + 1.0)).all(), np.abs(got - A_ref).max()
    sv = saved.cpu().numpy().astype(np.float64)
    _close(sv[0], mean, 1e-6, "saved mean")
    _close(sv[1], invstd, 1e-6, "saved invstd")
    _close(rv.cpu().numpy(), 0.9 + 0.1 * var * batch / (batch - 1), 1e-6, "running var")
    # the bits: keep AND gate, byte j of word [row / 4][col / 8] = row 4 g + j
    bw = bits.cpu().numpy().view(np.uint32).reshape(-1, width // 8)
    want = keep & (y > 0)
    rows = np.arange(batch)
    for c in (0, 3, 7):                                              # three of the eight bit positions, all rows
        cols = np.arange(c, width, 8)
        gotb = (bw[(rows // 4)[:, None], (cols // 8)[None, :]] >> (8 * (rows % 4)[:, None] + c)) & 1
        assert np.array_equal(gotb.astype(bool), want[:, cols]), c
    # backward (BatchNorm1d backward with batch statistics on the stored, rounded z)
    # (the kernels form S1 = sum dY' z and z^T X analytically from the UN-rounded z = x W0^T + b0 — the gates come from
    #  the stored, rounded z — so the tight reference uses z; with the rounded z in its place every element of z is off
    #  by up to 2^-9 relative, which averages out over the batch: checked at 1e-2)
    dY = 2.0 * want * dA.astype(np.float64)
    zhat = (z - mean) * invstd
    dgamma_ref, dbeta_ref = (dY * zhat).sum(0), dY.sum(0)
    dZ = gamma * invstd * (dY - dbeta_ref / batch - zhat * dgamma_ref / batch)
    dgamma_rounded = (dY * ((zr - mean) * invstd)).sum(0)
    dW0 = torch.full((width, IF), float("nan"), device=dev)
    db0, dg, db = (torch.full((width,), float("nan"), device=dev) for _ in range(3))
    rc = native.blh_skinny_encode_fused_bwd_bf16(st, dat.data_ptr(), xt.data_ptr(), w0.data_ptr(), bt.data_ptr(),
                                                 saved.data_ptr(), bits.data_ptr(), scratch.data_ptr(), dW0.data_ptr(),
                                                 db0.data_ptr(), dg.data_ptr(), db.data_ptr(), batch, width, IF)
    assert rc == 0, native.blh_status_string(rc)
    torch.cuda.synchronize()
    _close(db.cpu().numpy(), dbeta_ref, 1e-4, "dbeta")
    _close(dg.cpu().numpy(), dgamma_ref, 1e-4, "dgamma")
    _close(dg.cpu().numpy(), dgamma_rounded, 1e-2, "dgamma (stored z)")
    _close(dW0.cpu().numpy(), dZ.T @ x.astype(np.float64), 1e-4, "dW0")
    assert native.blh_skinny_encode_fused_fwd_bf16(st, xt.data_ptr(), w0.data_ptr(), bt.data_ptr(), gt.data_ptr(),
                                                   bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), nbt.data_ptr(), 0.1,
                                                   saved.data_ptr(), scratch.data_ptr(), A.data_ptr(), bits.data_ptr(),
                                                   ctypes.byref(drop), batch, 768, IF) != 0      # width % 512
