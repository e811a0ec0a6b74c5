"""Data-parallel host logic on CPU: world_size 2, gloo.

The bucketed, overlapped gradient all-reduce of bilinear_amd/dp.py is
backend-agnostic (RCCL on the GPUs, gloo here).  Each rank computes the gradients
of ITS shard with the NumPy oracle, feeds them through GradBucketReducer in the
order blh_backward reports them (decode -> encode), and must end up with the mean
of the two shards' gradients, bit-identical on both ranks; the replicated
clip + Adam then keeps the replicas identical without any parameter broadcast."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import numpy_oracle as O

NB, WIDTH, LOCAL_B = 1, 64, 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_grads(rank, st):
    x, t = O.synthetic_batch(50, 2 * LOCAL_B)
    masks = O.random_masks(51, 2 * LOCAL_B, NB, WIDTH)
    sl = slice(rank * LOCAL_B, (rank + 1) * LOCAL_B)
    pred, cache = O.forward({k: v.copy() for k, v in st.items()}, x[sl], [m[sl] for m in masks],
                            training=True)
    _, dpred = O.mse_loss(pred, t[sl])
    return O.backward(st, cache, dpred)


def _worker(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bilinear_amd.dp import GradBucketReducer
        from bilinear_amd.engine import ArenaLayout
        lay = ArenaLayout(NB, WIDTH)
        st = O.init_state(3, NB, WIDTH)
        grads = _shard_grads(rank, st)
        flat = torch.zeros(lay.total)
        for name, off, shape in lay.entries:
            flat[off:off + int(np.prod(shape))] = torch.from_numpy(grads[name].reshape(-1).copy())
        red = GradBucketReducer(flat, bucket_floats=3000)
        red.begin()
        # ranges in the order blh_backward reports them: decode first, then stages top-down
        starts = {n: o for n, o, _ in lay.entries}
        bounds = [starts["encode.0.weight"]]
        bounds += [starts["bilinear.%d.%d.0.weight" % (b, l)] for b in range(NB) for l in range(2)]
        bounds += [starts["decode.weight"], lay.total]
        for i in range(len(bounds) - 2, -1, -1):
            red.on_ready(bounds[i], bounds[i + 1] - bounds[i])
        red.finish()
        launched = list(red.launched)
        # every element reduced exactly once, buckets walk the arena downwards
        assert launched[0][1] == lay.total and launched[-1][0] == 0
        assert all(a[0] == b[1] for a, b in zip(launched[:-1], launched[1:]))
        assert len(launched) >= 2
        # expected: mean of both shards' oracle gradients
        other = _shard_grads(1 - rank, st)
        for name, off, shape in lay.entries:
            n = int(np.prod(shape))
            want = 0.5 * (grads[name].astype(np.float64) + other[name].astype(np.float64))
            got = flat[off:off + n].numpy().astype(np.float64).reshape(shape)
            assert np.abs(got - want).max() <= 1e-6 * (np.abs(want).max() + 1e-12) + 1e-12, name
        # replicas stay identical after the replicated clip + Adam
        keys = O.param_keys(NB)
        g = {name: flat[off:off + int(np.prod(shape))].numpy().reshape(shape).copy()
             for name, off, shape in lay.entries}
        O.clip_grad_norm(g, keys)
        opt = O.adam_init(st, keys)
        O.adam_step(st, g, opt, keys, 1e-3)
        mine = torch.from_numpy(np.concatenate([st[k].reshape(-1) for k in keys]))
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert torch.equal(both[0], both[1])
        open(os.path.join(result_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")


def test_bucket_merging_single_process():
    """Without a process group the reducer only records its bucket plan."""
    from bilinear_amd.dp import GradBucketReducer
    flat = torch.zeros(1000)
    red = GradBucketReducer(flat, bucket_floats=300)
    red.begin()
    for lo, hi in [(900, 1000), (700, 900), (650, 700), (200, 650), (0, 200)]:
        red.on_ready(lo, hi - lo)
    red.finish()
    assert red.launched == [(700, 1000), (200, 700), (0, 200)]


def test_default_bucket_size_is_a_quarter_of_the_arena():
    """Default plan: about four collectives per step, none under 1 Mi elements (dp.py:
    GradBucketReducer) — the arena of BASELINE configs[2] (4 blocks x 1024) reported stage by stage
    as blh_backward does it (decode first) gives four buckets, not nine."""
    from bilinear_amd.dp import GradBucketReducer
    W, stages = 1024, 9
    sizes = [W * 32 + 3 * W] + [W * W + 3 * W] * (stages - 1) + [48 * W + 64]
    total = sum(sizes)
    flat = torch.zeros(total)
    red = GradBucketReducer(flat)
    assert red.bucket_floats == max(1 << 20, total // 4)
    assert GradBucketReducer(torch.zeros(1000)).bucket_floats == 1 << 20
    red.begin()
    hi = total
    for sz in reversed(sizes):
        red.on_ready(hi - sz, sz)
        hi -= sz
    red.finish()
    assert len(red.launched) == 4
    assert red.launched[0][1] == total and red.launched[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(red.launched, red.launched[1:]))     # contiguous, descending


def _worker_bf16(rank, world, port, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bilinear_amd.dp import GradBucketReducer
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(4096, generator=g) * 1e-3
        mine = flat.clone()
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        red = GradBucketReducer(flat, bucket_floats=1024, compress="bf16")
        red.begin()
        for lo in (3072, 2048, 1024, 0):
            red.on_ready(lo, 1024)
        red.finish()
        want = 0.5 * (both[0].to(torch.bfloat16).float() + both[1].to(torch.bfloat16).float())
        # sum of two bf16 values, rounded once to bf16 by the reduction, halved exactly
        assert torch.allclose(flat, want, rtol=2 ** -7, atol=0)
        assert len(red.launched) == 4
        open(os.path.join(result_dir, "bf16_ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_bf16_compressed_buckets_world2_gloo(tmp_path):
    """compress="bf16": buckets travel as bf16 (half the bytes), average = mean of the rounded
    shards to one bf16 rounding."""
    port = _free_port()
    mp.spawn(_worker_bf16, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "bf16_ok0") and os.path.exists(tmp_path / "bf16_ok1")


def test_collectives_argument_of_the_data_parallel_driver():
    """``collectives="native"`` (the library issues the RCCL calls itself, csrc/comm.hip) is an opt-in of HIP devices with
    fp32 buckets: the constructor refuses what it cannot do, and the communicator is not created on a CPU module."""
    from bilinear_amd.dp import DataParallel

    class _Eng:
        device = torch.device("cpu")

    class _Mod:
        engine = _Eng()

        def parameters(self):
            return iter([torch.zeros(1)])

    with pytest.raises(ValueError):
        DataParallel(_Mod(), None, collectives="rccl")
    assert DataParallel(_Mod(), None, collectives="native", compress="bf16").compress == "bf16"   # (bf16 buckets: both drivers)
    with pytest.raises(ValueError):
        DataParallel(_Mod(), None, collectives="native", compress="fp8")
    with pytest.raises(ValueError):
        DataParallel(_Mod(), None, collectives="native", native_tail="main")
    dp = DataParallel(_Mod(), None, collectives="native")
    assert dp.collectives == "native" and dp.native_tail == "producer" and dp._comm is None
    with pytest.raises(RuntimeError, match="HIP device"):
        dp.native_comm()
    assert DataParallel(_Mod(), None).collectives == "torch"        # the default stays torch.distributed
