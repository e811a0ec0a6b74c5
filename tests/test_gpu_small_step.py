"""The one-launch training step for batches of at most 64 rows (csrc/small_step.hip, BLH_OPT_SMALL_STEP): the
reference's own batch size (/root/reference/util/config.py:15).  The golden-vector tests of test_gpu_parity.py
(reference fixtures at B = 8 and 64) run through it by default; here it is compared with the multi-launch path
on the same inputs — every tensor the step writes — over ragged batches, widths and depths, with Philox and with
explicit dropout masks, and against the fp64 oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda", 0)


def _rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def _pair(dev, nb, width, seed=11, mode=1):
    import bilinear_amd
    nets = []
    for small in (mode, 0):
        torch.manual_seed(seed)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=nb, width=width, gemm_dtype="fp32")
        net.train()
        net.engine.ensure(dev)
        net.engine.seed = 4242
        net.engine.set_small_step(small)
        nets.append((net, opt))
    return nets


@pytest.mark.parametrize("mode", [1])           # one launch per stage (the persistent form left the library in round 6)
@pytest.mark.parametrize("nb,width,batch", [(2, 1024, 64), (2, 1024, 8), (2, 1024, 33), (1, 256, 2), (0, 64, 17),
                                            (4, 512, 64), (3, 1024, 50),
                                            (2, 1024, 128), (2, 1024, 96), (1, 512, 65), (3, 768, 101),
                                            (2, 1024, 256), (2, 1024, 200), (1, 512, 300), (2, 1024, 384), (1, 256, 257)])
def test_small_step_equals_the_multi_launch_step(nb, width, batch, mode):
    dev = _dev()
    (na, oa), (nb_, ob) = _pair(dev, nb, width, mode=mode)
    assert na.engine.ctx.get_option(4) == mode and nb_.engine.ctx.get_option(4) == 0
    g = torch.Generator().manual_seed(batch)
    # (at B = 2 BatchNorm leaves gradients that are rounding noise, which Adam's sign-like update amplifies into
    #  the parameters: one step there, three elsewhere)
    steps = 1 if batch < 8 else 3
    for s in range(steps):
        x = torch.randn(batch, 32, generator=g).to(dev)
        t = torch.randn(batch, 48, generator=g).to(dev)
        pa, la = na.train_step(oa, x, t, max_norm=1.0)
        pb, lb = nb_.train_step(ob, x, t, max_norm=1.0)
        torch.cuda.synchronize()
        # a ReLU gate within rounding of zero may open on one path and not on the other: whole-tensor norms
        tol = 2e-5 * (s + 1)
        # (from the second step on the two parameter sets differ by Adam's rounding-level noise, and with it a few
        #  ReLU gates: one flipped gate in 10^5 elements moves a gradient tensor by ~3e-3; the first step starts from
        #  identical parameters and is held to the tight bound, the oracle test below is the rigorous one)
        gtol = 10 * tol if s == 0 else 5e-3
        ptol = tol if s == 0 else 1e-3
        assert _rel(pa, pb) <= ptol, ("pred", s, _rel(pa, pb))
        assert abs(la.item() - lb.item()) <= ptol * abs(lb.item())
        assert _rel(na.engine.grads, nb_.engine.grads) <= gtol, ("grads", s, _rel(na.engine.grads, nb_.engine.grads))
        assert _rel(oa._exp_avg, ob._exp_avg) <= gtol
        assert _rel(oa._exp_avg_sq, ob._exp_avg_sq) <= 2 * gtol
        # (Adam's update is sign-like where the gradient is rounding noise — at B = 2 BatchNorm leaves nothing
        #  else — so the parameters get the bound of one update per step next to the norm)
        assert float((na.engine.params - nb_.engine.params).abs().max()) <= 2.1e-3 * (s + 1)
        assert _rel(na.engine.params, nb_.engine.params) <= (tol if (s == 0 and batch * width >= 64 * 512) else 1e-3)
        assert _rel(na.engine.bn_running, nb_.engine.bn_running) <= (tol if s == 0 else 1e-4)
        sa, sb = oa.last_grad_norm_stats.cpu(), ob.last_grad_norm_stats.cpu()
        assert abs(sa[0] - sb[0]) <= gtol * abs(sb[0]) and abs(sa[1] - sb[1]) <= gtol
    assert int(na.encode[1].num_batches_tracked) == steps == int(nb_.encode[1].num_batches_tracked)


@pytest.mark.parametrize("mode", [1])
def test_small_step_is_deterministic(mode):
    """Two runs of the same steps are bit-identical (every sum has a fixed order)."""
    dev = _dev()
    (na, oa), _ = _pair(dev, 2, 1024, mode=mode)
    (nc, oc), _ = _pair(dev, 2, 1024, mode=mode)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 32, generator=g).to(dev); t = torch.randn(64, 48, generator=g).to(dev)
    for _ in range(3):
        pa, la = na.train_step(oa, x, t)
        pc, lc = nc.train_step(oc, x, t)
        assert torch.equal(pa, pc) and la.item() == lc.item()
    assert torch.equal(na.engine.params, nc.engine.params) and torch.equal(oa._exp_avg_sq, oc._exp_avg_sq)
    assert torch.equal(na.engine.grads, nc.engine.grads) and torch.equal(na.engine.bn_running, nc.engine.bn_running)


@pytest.mark.parametrize("mode", [1])
@pytest.mark.parametrize("batch", [64, 24, 128, 100, 256, 330, 384])
def test_small_step_against_the_fp64_oracle_and_philox_replay(batch, mode):
    """(a) explicit gate-safe masks: every observable of the step against oracle/numpy_oracle.py at the tight fp32
    tolerance of the timed-path tests; (b) the Philox step is bit-identical to the explicit-mask step fed the
    materialised Philox masks (same keep bits as the multi-launch kernels, philox.h)."""
    import test_gpu_timed_path as T
    dev = _dev()
    nb, width = 2, 1024
    entry = T._entry_with_masks(nb, width, batch, dev)
    xt, tt = torch.from_numpy(entry["x"]).to(dev), torch.from_numpy(entry["t"]).to(dev)
    r = T._run_oracle(entry, entry["safe"])
    net, opt = T._build(entry["st0"], dev, nb, width, "fp32")
    assert net.engine.ctx.get_option(4) == 1                   # default: one launch per stage
    net.engine.set_small_step(mode)
    net.engine.set_dropout_masks(entry["safe"])
    pred, loss = net.train_step(opt, xt, tt, max_norm=1.0)
    torch.cuda.synchronize()
    T._compare_fused_step(net, opt, pred, loss.item(), r, T.TIGHT)
    out = []
    for explicit in (False, True):
        net, opt = T._build(entry["st0"], dev, nb, width, "fp32")
        net.engine.set_small_step(mode)
        if explicit:
            net.engine.set_dropout_masks(entry["philox"])
        pred, loss = net.train_step(opt, xt, tt, max_norm=1.0)
        torch.cuda.synchronize()
        out.append((pred.clone(), net.engine.grads.clone(), net.engine.params.clone(), opt._exp_avg.clone(),
                    net.engine.bn_running.clone(), loss.item()))
    for a, b, what in zip(out[0][:5], out[1][:5], ("pred", "grads", "params", "exp_avg", "running")):
        assert torch.equal(a, b), what
    assert out[0][5] == out[1][5]


@pytest.mark.parametrize("mode", [1])
@pytest.mark.parametrize("nb,width,batch", [(2, 1024, 64), (1, 512, 40), (2, 1024, 128), (1, 256, 77), (2, 1024, 256),
                                            (1, 512, 380)])
def test_drop_in_forward_and_backward_take_the_one_launch_kernels(nb, width, batch, mode):
    """The reference's five-call step body (/root/reference/train_bilinear.py:75-83) at <= 64 rows: forward and
    backward are one launch each (SS_FWD / SS_BWD, the saved activations cross in the workspace).  Raw gradients,
    the clipped step and the BatchNorm buffers against the multi-launch path; switching the option between forward
    and backward does not mix the two saved-activation formats; gradient accumulation over two backwards works."""
    import bilinear_amd
    dev = _dev()
    (na, oa), (nm, om) = _pair(dev, nb, width, mode=mode)
    g = torch.Generator().manual_seed(9)
    crit = torch.nn.MSELoss()
    for s in range(3):
        x = torch.randn(batch, 32, generator=g).to(dev)
        t = torch.randn(batch, 48, generator=g).to(dev)
        raws = []
        for net, opt in ((na, oa), (nm, om)):
            opt.zero_grad()
            pred = net(x)
            loss = crit(pred, t)
            if net is na and s == 1:
                net.engine.set_small_step(False)      # the backward must still be the small-batch one
            loss.backward()
            if net is na and s == 1:
                net.engine.set_small_step(mode)
            raws.append((pred.detach().clone(), loss.item(), net.engine.grads.clone()))
            bilinear_amd.clip_grad_norm_(net.parameters(), max_norm=1, module=net)
            opt.step()
        torch.cuda.synchronize()
        tol = 2e-5 * (s + 1)
        assert _rel(raws[0][0], raws[1][0]) <= tol
        assert abs(raws[0][1] - raws[1][1]) <= tol * abs(raws[1][1])
        assert _rel(raws[0][2], raws[1][2]) <= 10 * tol, ("raw grads", s, _rel(raws[0][2], raws[1][2]))
        assert _rel(na.engine.params, nm.engine.params) <= tol
        assert _rel(na.engine.bn_running, nm.engine.bn_running) <= tol
    assert int(na.encode[1].num_batches_tracked) == 3
    # accumulation: two forward/backward pairs without zero_grad add up
    x = torch.randn(batch, 32, generator=g).to(dev); t = torch.randn(batch, 48, generator=g).to(dev)
    oa.zero_grad()
    crit(na(x), t).backward()
    g1 = na.engine.grads.clone()
    crit(na(x), t).backward()
    torch.cuda.synchronize()
    # (dropout draws a new mask for the second forward, so only the structure is checked: finite, changed, and the
    #  parameters' .grad are still the arena views)
    assert torch.isfinite(na.engine.grads).all() and not torch.equal(g1, na.engine.grads)
    for (_, p, off, shape), ptr in zip(na.engine._named_params(), na.engine.grad_ptrs()):
        assert p.grad.data_ptr() == ptr


def test_small_step_shape_sweep():
    """Seeded sweep over depths, widths (every multiple of 64 up to 1024 is a legal width: the XCD-aware column-group
    map, the batched weight-gradient GEMM and the ragged tiles all depend on it) and batches 2 .. 384 (above 64 rows the stage kernels run 8 waves per workgroup; batches that are
    not multiples of 32 take the in-kernel weight gradients): one fused step and one five-call step against the
    multi-launch path."""
    import bilinear_amd
    dev = _dev()
    rng = np.random.default_rng(20260)
    crit = torch.nn.MSELoss()
    combos = [(int(rng.integers(0, 6)), int(64 * rng.integers(1, 17)), int(rng.integers(2, 385))) for _ in range(14)]
    combos += [(2, 960, 64), (1, 192, 32), (5, 448, 63), (2, 320, 127), (1, 1024, 66)]
    for nb, width, batch in combos:
        (na, oa), (nm, om) = _pair(dev, nb, width, seed=nb * 1000 + width + batch)
        g = torch.Generator().manual_seed(width + batch)
        x = torch.randn(batch, 32, generator=g).to(dev)
        t = torch.randn(batch, 48, generator=g).to(dev)
        pa, la = na.train_step(oa, x, t, max_norm=1.0)
        pm, lm = nm.train_step(om, x, t, max_norm=1.0)
        torch.cuda.synchronize()
        what = (nb, width, batch)
        assert _rel(pa, pm) <= 2e-5, (what, "pred", _rel(pa, pm))
        assert abs(la.item() - lm.item()) <= 2e-5 * abs(lm.item()), what
        if batch >= 8:           # (below that BatchNorm leaves gradients that are rounding noise)
            assert _rel(na.engine.grads, nm.engine.grads) <= 2e-4, (what, "grads", _rel(na.engine.grads, nm.engine.grads))
            assert _rel(oa._exp_avg_sq, om._exp_avg_sq) <= 4e-4, what
        assert _rel(na.engine.bn_running, nm.engine.bn_running) <= 2e-5, what
        raws = []
        for net, opt in ((na, oa), (nm, om)):
            opt.zero_grad()
            loss = crit(net(x), t)
            loss.backward()
            raws.append(net.engine.grads.clone())
        torch.cuda.synchronize()
        if batch >= 8:
            # (second step: the parameters differ by Adam noise already, a few ReLU gates with them)
            assert _rel(raws[0], raws[1]) <= 5e-3, (what, "drop-in raw grads", _rel(raws[0], raws[1]))
        assert torch.isfinite(raws[0]).all(), what


@pytest.mark.parametrize("mode", [1])
def test_captured_small_step_replays_like_eager(mode):
    """hipGraph replay of the small-batch step (blh_train_step_captured: Adam scalars and the dropout step from device
    memory; persistent form: the barrier's base survives from replay to replay) is bit-identical to the eager step."""
    import bilinear_amd
    dev = _dev()

    def make():
        torch.manual_seed(5)
        net, opt, _, _ = bilinear_amd.load(dev, num_blocks=2, gemm_dtype="fp32")
        net.train()
        net.engine.ensure(dev)
        net.engine.seed = 99
        net.engine.set_small_step(mode)
        return net, opt

    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(64, 32, generator=g).to(dev) for _ in range(4)]
    ts = [torch.randn(64, 48, generator=g).to(dev) for _ in range(4)]
    net_a, opt_a = make()
    net_b, opt_b = make()
    step = bilinear_amd.CapturedTrainStep(net_b, opt_b, 64)
    for i in range(4):
        pa, la = net_a.train_step(opt_a, xs[i], ts[i])
        pb, lb = step(xs[i], ts[i])
        assert torch.equal(pa, pb) and la.item() == lb.item(), i
    torch.cuda.synchronize()
    assert torch.equal(net_a.engine.params, net_b.engine.params)
    assert torch.equal(opt_a._exp_avg_sq, opt_b._exp_avg_sq)


def test_two_models_interleaved_on_one_context_keep_their_saved_formats():
    """A C-ABI user may drive several models through ONE context (include/bilinear_hip.h: one in-flight call per
    context).  The context remembers, per workspace, which forward saved the activations there: a small-batch
    forward of model A followed by a multi-launch forward of model B (and the other way round) must send each
    backward to the kernels that can read its workspace.  Against the same two models on contexts of their own."""
    import bilinear_amd
    dev = _dev()
    crit = torch.nn.MSELoss()
    g = torch.Generator().manual_seed(21)
    xs = torch.randn(64, 32, generator=g).to(dev); ts = torch.randn(64, 48, generator=g).to(dev)
    xl = torch.randn(640, 32, generator=g).to(dev); tl = torch.randn(640, 48, generator=g).to(dev)

    def build():
        nets = []
        for seed in (3, 4):
            torch.manual_seed(seed)
            net, _, _, _ = bilinear_amd.load(dev, num_blocks=1, width=256, gemm_dtype="fp32")
            net.train(); net.engine.ensure(dev); net.engine.seed = 77
            nets.append(net)
        return nets

    ref_a, ref_b = build()
    crit(ref_a(xs), ts).backward()
    crit(ref_b(xl), tl).backward()
    torch.cuda.synchronize()
    for order in ("small_first", "large_first"):
        a, b = build()
        b.engine.ctx = a.engine.ctx                  # both models on A's context
        if order == "small_first":
            la = crit(a(xs), ts); lb = crit(b(xl), tl)
            la.backward(); lb.backward()
        else:
            lb = crit(b(xl), tl); la = crit(a(xs), ts)
            lb.backward(); la.backward()
        torch.cuda.synchronize()
        assert torch.equal(a.engine.grads, ref_a.engine.grads), order
        assert torch.equal(b.engine.grads, ref_b.engine.grads), order


def test_raw_current_stream_accessor_follows_torch_stream_contexts():
    """bilinear_amd._native.current_stream() (the raw accessor the host-bound drop-in step uses) returns the handle of
    torch's current stream — the default stream, and a side stream inside ``torch.cuda.stream(...)``."""
    from bilinear_amd import _native as N
    dev = _dev()
    with torch.cuda.device(dev):
        assert (N.current_stream().value or 0) == torch.cuda.current_stream().cuda_stream
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            assert (N.current_stream().value or 0) == side.cuda_stream
        assert (N.current_stream().value or 0) == torch.cuda.current_stream().cuda_stream
