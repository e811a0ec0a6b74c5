"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports
every symbol include/bilinear_hip.h declares; the host-side mirror of the reference
interface (model.bilinear.{heavy_linear, BilinearUnit, load}) has the reference's
names, state_dict keys and checkpoint discovery; the product refuses to run without
a HIP device instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import numpy_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "bilinear_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(blh_[a-z0-9_]+)\s*\(", text))
    names -= {"blh_grad_ready_fn"}
    return sorted(names)


def test_library_exports_every_declared_symbol(native):
    decl = _declared_symbols()
    assert len(decl) >= 18
    for name in decl:
        assert hasattr(native, name), "libbilinear_hip.so does not export %s" % name
    from bilinear_amd import _native
    assert sorted(_native.exported_names()) == decl, "ctypes table and header disagree"
    assert native.blh_abi_version() == 6
    assert native.blh_status_string(0) == b"ok"
    assert native.blh_status_string(-4) == b"workspace too small"
    assert b"RCCL" in native.blh_status_string(-5)


def test_every_abi_version_check_names_the_header_version(native):
    """The header's BLH_ABI_VERSION, the library's answer, build()'s assertion (__graft_entry__.py: the driver's "does it
    build" check) and the stub in INTEGRATION.md must agree — a bump that misses one of them fails HERE, not on the driver."""
    hdr = open(os.path.join(REPO, "include", "bilinear_hip.h")).read()
    ver = int(re.search(r"#define\s+BLH_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert native.blh_abi_version() == ver
    entry = open(os.path.join(REPO, "__graft_entry__.py")).read()
    assert re.findall(r"blh_abi_version\(\)\s*==\s*(\d+)", entry) == [str(ver)]
    integ = open(os.path.join(REPO, "INTEGRATION.md")).read()
    assert re.findall(r"assert _lib\.blh_abi_version\(\)\s*==\s*(\d+)", integ) == [str(ver)]      # (the stub)


def test_comm_entry_points_validate_without_gpu(native):
    """csrc/comm.hip: librccl is resolved with dlopen (torch's copy is in the process), nothing is called on a device
    here — argument checks only."""
    from bilinear_amd import _native as N
    assert native.blh_rccl_version() >= 20000          # an NCCL-style version code, e.g. 22606
    buf = ctypes.create_string_buffer(128)
    assert native.blh_rccl_unique_id(buf, 64) == -1    # an id has 128 bytes
    assert native.blh_rccl_unique_id(None, 128) == -1
    h = ctypes.c_void_p()
    assert native.blh_comm_create(ctypes.byref(h), buf, 128, 0, 0) == -1       # world < 1
    assert native.blh_comm_create(ctypes.byref(h), buf, 128, 2, 2) == -1       # rank out of range
    assert native.blh_comm_create(ctypes.byref(h), buf, 100, 1, 0) == -1
    assert not h.value
    assert native.blh_comm_destroy(None) == 0
    assert native.blh_comm_info(None, None, None, None, None) == -1
    assert native.blh_comm_all_reduce(None, None, None, 4, 0, 0) == -1
    assert native.blh_comm_stream(None) is None
    with pytest.raises(ValueError):
        N.Comm("cuda:0", b"x" * 5, 1, 0)


def test_arena_layout_matches_reference_parameter_order(native):
    from bilinear_amd import _native as N
    from bilinear_amd.engine import ArenaLayout
    for nb, width, nparams in [(2, 1024, 4291632), (4, 1024, 8498224), (8, 2048, 67377200)]:
        lay = ArenaLayout(nb, width)
        names = [n for n, _, _ in lay.entries]
        assert names == O.param_keys(nb)                       # module.parameters() order
        spec = {k: s for k, s, kind in O.state_spec(nb, width) if kind == "param"}
        assert all(tuple(s) == tuple(spec[n]) for n, _, s in lay.entries)
        assert sum(int(np.prod(s)) for _, _, s in lay.entries) == nparams
        offs = [o for _, o, _ in lay.entries]
        assert offs == sorted(offs) and all(o % 64 == 0 for o in offs)
        assert lay.total >= nparams and lay.total % 64 == 0
        assert lay.bn_floats == (1 + 2 * nb) * 2 * width
        assert lay.workspace_bytes(4096) > 0
    # invalid descriptions are refused with a status, not a crash
    bad = N.ModelDesc(2, 1000, 32, 48)
    assert native.blh_param_arena_floats(ctypes.byref(bad)) == -2


def test_argument_validation_without_gpu(native):
    from bilinear_amd import _native as N
    d = N.ModelDesc(2, 1024, 32, 48)
    drop = N.Dropout(None, 1, 0, 0)
    # NULL pointers / bad sizes are rejected before any HIP call
    # (a NULL context is refused first: every network-level call needs a caller-owned context)
    assert native.blh_forward_train(None, ctypes.byref(d), None, None, None, None, None, ctypes.byref(drop),
                                    0.1, None, 0, None, 64) == -1
    assert native.blh_context_create(None) == -1
    assert native.blh_context_destroy(None) == 0
    assert native.blh_context_set_option(None, 0, 1) == -1
    assert native.blh_context_side_stream(None) is None
    assert native.blh_gemm_f32(None, None, 0, 0, None, 0, 0, None, 0, 0, 0, 0, 1, None, None, 0) == -1
    drop_bad = N.Dropout(None, 1, 0, 5)          # Philox needs row_offset % 32 == 0
    ws = ctypes.create_string_buffer(512)
    assert native.blh_dropout_mask(None, ctypes.byref(drop_bad), 0, 64, 64, ws) == -2


def test_module_surface_matches_reference():
    import bilinear_amd
    import model                                   # top-level alias used by reference scripts
    assert model.bilinear.BilinearUnit is bilinear_amd.BilinearUnit
    assert {"heavy_linear", "BilinearUnit", "load"} <= set(dir(model.bilinear))
    net = bilinear_amd.BilinearUnit()
    sd = net.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in O.state_spec(2, 1024)]
    assert len(sd) == 37 and sum(p.numel() for p in net.parameters()) == 4291632
    h = bilinear_amd.heavy_linear(32, 64)
    assert [type(m).__name__ for m in h] == ["Linear", "BatchNorm1d", "ReLU", "Dropout"]
    assert h[3].p == 0.5 and h[1].eps == 1e-5 and h[1].momentum == 0.1
    # every stand-alone stage owns a Philox stream (stacked equal stages never share a mask),
    # all above the stage indices a BilinearUnit uses (<= 32)
    h2 = bilinear_amd.heavy_linear(32, 64)
    assert h._stage_id != h2._stage_id and min(h._stage_id, h2._stage_id) > 32
    # reset_statistics (model/bilinear.py:43-55): cumulative average mode
    net.encode[1].running_mean.fill_(3.0)
    net.reset_statistics()
    assert net.encode[1].momentum is None and float(net.encode[1].running_mean.abs().max()) == 0.0


def test_no_cpu_fallback():
    import bilinear_amd
    net = bilinear_amd.BilinearUnit(1, 64)
    with pytest.raises(RuntimeError, match="HIP device"):
        net(torch.zeros(8, 32))
    net.eval()
    with pytest.raises(RuntimeError, match="HIP device"):
        net(torch.zeros(8, 32))
    with pytest.raises(RuntimeError, match="HIP device"):
        bilinear_amd.heavy_linear(32, 64)(torch.zeros(8, 32))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from bilinear_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        _native.lib()


def test_load_initialises_and_discovers_checkpoints(tmp_path):
    import bilinear_amd
    torch.manual_seed(0)
    net, opt, step, epoch = bilinear_amd.load(torch.device("cpu"))
    assert (step, epoch) == (1, 0)
    assert isinstance(opt, torch.optim.Optimizer) and opt.param_groups[0]["lr"] == 1e-3
    # kaiming_normal (fan_in, gain sqrt 2): std 0.25 for encode, 0.0442 for hidden
    assert abs(net.encode[0].weight.std().item() - 0.25) < 0.01
    assert abs(net.bilinear[0][0][0].weight.std().item() - (2 / 1024) ** 0.5) < 0.001
    assert float(net.encode[1].weight.min()) == 1.0 and float(net.encode[1].bias.abs().max()) == 0.0
    # checkpoint in the reference's format and naming (train_bilinear.py:92-104)
    d = tmp_path / "parameter"
    d.mkdir()
    for e in (3, 12):
        torch.save({"epoch": e, "step": 100 * e, "state": net.state_dict(),
                    "optimizer": torch.optim.Adam(net.parameters(), lr=5e-4).state_dict()},
                   str(d / ("%d.save" % e)))
    net2, opt2, step2, epoch2 = bilinear_amd.load(torch.device("cpu"), parameter_dir=str(d))
    assert (step2, epoch2) == (1200, 12)
    assert opt2.param_groups[0]["lr"] == 5e-4
    assert torch.equal(net2.decode.weight, net.decode.weight)
    for pg in opt2.param_groups:                   # lr-decay hook of train_bilinear.py:66-70
        pg["lr"] = O.lr_decay_function(step2)
    assert opt2.param_groups[0]["lr"] == O.lr_decay_function(1200)
    (d / "notes.txt.bak").write_text("x")         # the reference raises on names with != 1 dot
    with pytest.raises(ValueError):
        bilinear_amd.load(torch.device("cpu"), parameter_dir=str(d))


def test_host_layout_logic_under_sanitizers():
    """The arena layout and workspace-carving arithmetic of the C ABI (bilinear_amd/csrc/
    api_layout.h, pure C++) compiled with AddressSanitizer + UBSan and walked over every GEMM mode,
    0-15 blocks, widths 64-4096 and batches 2-131072: carved regions must be aligned, disjoint, in
    order and inside the reported size (tools/host_sanitize.cpp)."""
    import subprocess
    csrc = os.path.join(REPO, "bilinear_amd", "csrc")
    out = subprocess.run(["make", "-C", csrc, "sanitize"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "host sanitize ok" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr


def test_custom_ops_trace_through_their_fake_implementations():
    """torch.ops.bilinear_hip.* carry shape functions (register_fake): under FakeTensorMode — what
    torch.compile / export trace with — every operator returns tensors of the right shape and
    dtype without touching a device (no kernel runs: there is no CPU implementation at all)."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode

    import bilinear_amd.ops  # noqa: F401  (registers the operators)
    with FakeTensorMode():
        B, W, total = 64, 1024, 4291840
        x = torch.empty(B, 32)
        t = torch.empty(B, 48)
        params, grads, m, v = (torch.empty(total) for _ in range(4))
        running = torch.empty(5, 2, W)
        nbt = torch.empty(5, dtype=torch.int64)
        ws = torch.empty(1 << 20, dtype=torch.uint8)
        stats = torch.empty(2)
        pred = torch.ops.bilinear_hip.eval_fwd(x, params, running, ws, 0, 2, W, 0)
        assert tuple(pred.shape) == (B, 48) and pred.dtype == torch.float32
        pred = torch.ops.bilinear_hip.forward_train(x, params, running, nbt, ws, None, 0, 2, W, 0, 1, 0, 0, 0.1)
        assert tuple(pred.shape) == (B, 48)
        assert torch.ops.bilinear_hip.backward(x, pred, params, ws, grads, None, 0, 2, W, 0, 1, 0, 0) is None
        # the differentiable operator: the parameter views carry requires_grad, and a traced backward (what
        # AOTAutograd records) reaches the gradient arena through the registered formula
        views = [params[:W * 32].view(W, 32).detach().requires_grad_(), params[W * 32:W * 33].detach().requires_grad_()]
        pred, saved, new_running, new_nbt = torch.ops.bilinear_hip.lifter_train(
            x, views, params, running, nbt, None, 0, 2, W, 0, 1, 0, 0, 0.1, [0, W * 32], 1 << 20)
        assert tuple(pred.shape) == (B, 48) and pred.requires_grad
        # functional: what the native call writes comes back as outputs, no input is mutated
        assert saved.dtype == torch.uint8 and saved.numel() == 1 << 20
        assert new_running.shape == running.shape and new_nbt.dtype == torch.int64
        gv = torch.autograd.grad(pred.sum(), views)
        assert [tuple(g.shape) for g in gv] == [(W, 32), (W,)]
        g = torch.ops.bilinear_hip.lifter_backward(x, pred.detach(), params, saved, None, 0, 2, W, 0, 1, 0, 0)
        assert g.shape == params.shape
        schema = str(torch.ops.bilinear_hip.lifter_train.default._schema)
        assert "!" not in schema, schema                     # no mutable annotation: and none is needed
        assert "(a!) saved" in str(torch.ops.bilinear_hip.lifter_backward.default._schema)
        pred, loss = torch.ops.bilinear_hip.train_step(x, t, params, grads, m, v, running, nbt, ws, stats, None,
                                                       0, 2, W, 0, 1, 0, 0, 0.1, 1e-3, 0.9, 0.999, 1e-8, 1.0, 1)
        assert tuple(pred.shape) == (B, 48) and tuple(loss.shape) == ()


def test_cached_parameter_list_follows_module_surgery():
    """Engine._named_params() is cached (the drop-in step asks for it seven times per step and walking
    named_parameters() was half of the host time at batch 64); the cache is re-validated through the module-tree
    links it was resolved through, so replacing a submodule or a Parameter object is seen at the next call."""
    import torch
    import bilinear_amd
    net = bilinear_amd.BilinearUnit(num_blocks=1, width=64)
    eng = net.engine
    a = eng._named_params()
    assert eng._named_params() is a                                   # cached
    assert [n for n, _, _, _ in a] == [n for n, _ in net.named_parameters()]
    assert all(p is q for (_, p, _, _), (_, q) in zip(a, net.named_parameters()))
    bns = eng._bn_modules()
    assert bns[0] is net.encode[1] and bns[2] is net.bilinear[0][1][1]
    # a new Parameter object in place of an old one
    net.decode.bias = torch.nn.Parameter(torch.zeros(48))
    b = eng._named_params()
    assert b is not a and b[-1][1] is net.decode.bias
    # a replaced submodule (its parameters are new objects)
    net.encode[0] = torch.nn.Linear(32, 64)
    c = eng._named_params()
    assert c is not b and c[0][1] is net.encode[0].weight and c[1][1] is net.encode[0].bias
    net.bilinear[0][0][1] = torch.nn.BatchNorm1d(64)
    assert eng._bn_modules()[1] is net.bilinear[0][0][1]
    # in-place data changes keep the cache (same objects)
    d = eng._named_params()
    with torch.no_grad():
        net.decode.weight.mul_(2.0)
    assert eng._named_params() is d


def test_product_lr_decay_hook_is_the_reference_one():
    """bilinear_amd.config (what train_bilinear.py imports) restates /root/reference/util/config.py:19-23: the hook fires
    on the pre-increment step at step 1 and every 100000 steps, lr = 1e-3 * 0.96 ** (step / 100000); checked against the
    oracle's restatement (pinned by the reference's own fixtures: the golden lr values) and by value."""
    from bilinear_amd import config
    from oracle import numpy_oracle as O
    hook = config.bilinear.lr_decay
    assert hook.activate is True and config.bilinear.batch_size == 64 and config.bilinear.comment == "Bilinear GT"
    for step in (1, 2, 99999, 100000, 100001, 200000, 250000, 300000, 12345678):
        assert hook.condition(step) == O.lr_decay_condition(step), step
        assert hook.function(step) == O.lr_decay_function(step), step
    assert hook.function(1) == 1.0e-3 * 0.96 ** (1 / 100000) and hook.function(100000) == 1.0e-3 * 0.96
    assert [s for s in range(1, 200002) if hook.condition(s)] == [1, 100000, 200000]


def test_loss_ring_reports_every_step_with_one_readback_per_window():
    """bilinear_amd.LossRing on the CPU (host logic only): every step's loss reaches the sink in order, read back in
    windows of `every` steps and at flush(); push() copies a scalar, slot() hands out the tensor to write into."""
    import torch
    from bilinear_amd import LossRing
    got = []
    ring = LossRing(torch.device("cpu"), every=4, sink=lambda s, v: got.append((s, v)))
    for step in range(1, 11):
        if step % 2:
            ring.slot().fill_(float(step))           # (the fused step writes its loss here)
        else:
            ring.push(torch.tensor(float(step)))     # (the five-call loop copies its loss tensor)
        ring.advance(step)
        assert len(got) == 4 * (step // 4)           # nothing is read back between two windows
    ring.flush()
    assert got == [(s, float(s)) for s in range(1, 11)] and ring.last == 10.0


def test_optimizer_step_hooks_and_zero_grad_survive_the_lean_wrappers():
    """bilinear_amd.Adam.step skips torch's per-call profiler / hook wrapper when no step hook is registered (host
    time of the five-call loop) and takes it when one is: pre and post hooks registered the torch way still fire, in
    order, around the step; zero_grad() keeps torch's semantics (set_to_none default)."""
    import torch
    import bilinear_amd
    net = bilinear_amd.BilinearUnit(num_blocks=1, width=64)
    opt = bilinear_amd.Adam(net.parameters(), lr=1e-3, module=net)
    assert getattr(type(opt).step, "hooked", False) is True            # torch did not wrap the class's step
    calls = []
    assert opt.step() is None                                           # (no device arenas yet: nothing to do, no hook)
    h1 = opt.register_step_pre_hook(lambda o, a, k: calls.append("pre"))
    h2 = opt.register_step_post_hook(lambda o, a, k: calls.append("post"))
    opt.step()
    assert calls == ["pre", "post"]
    h1.remove(); h2.remove()
    opt.step()
    assert calls == ["pre", "post"]
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
    opt.zero_grad()
    assert all(p.grad is None for p in net.parameters())
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    opt.zero_grad(set_to_none=False)
    assert all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in net.parameters())
