"""Device-resident input pipeline against the oracle's restatement of
/root/reference/H36M/data.py: joint selection, root-centring, train-split z-scoring, action
decoding, DataLoader batch semantics, pickle round trip.  Runs on the CPU device (torch ops);
the GPU variant lives in test_gpu_parity.py."""
import os
import pickle

import numpy as np
import pytest
import torch

from bilinear_amd.data import ACTIONS, DevicePoseDataset, decode_action, synthetic_raw
from oracle import numpy_oracle as O


def _oracle_split(raw_train, raw):
    ptr, str_ = O.h36m_flatten(raw_train["part"], raw_train["S"])
    p, s = O.h36m_flatten(raw["part"], raw["S"])
    mx, sx = O.h36m_stats(ptr)
    mt, st = O.h36m_stats(str_)
    return O.h36m_normalise(p, mx, sx), O.h36m_normalise(s, mt, st), mt, st


def test_preprocessing_matches_oracle():
    raw_tr, raw_va = synthetic_raw(5000, seed=1), synthetic_raw(777, seed=2)
    train = DevicePoseDataset(raw_tr, "cpu")
    valid = DevicePoseDataset(raw_va, "cpu", stats_from=train)
    for ds, raw in ((train, raw_tr), (valid, raw_va)):
        x, t, mt, st = _oracle_split(raw_tr, raw)
        assert ds.x.shape == (len(raw["part"]), 32) and ds.t.shape == (len(raw["part"]), 48)
        np.testing.assert_allclose(ds.x.numpy(), x, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(ds.t.numpy(), t, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(ds.norm_mean.numpy(), mt, rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(ds.norm_stddev.numpy(), st, rtol=1e-5)
    # the train split is z-scored with its own statistics
    assert abs(float(train.x.mean())) < 1e-4 and abs(float(train.t.std(unbiased=False)) - 1) < 1e-2
    # the nose (joint 9) is gone, the pelvis is gone and every 3D joint is root-relative
    p = raw_tr["part"][3]
    assert np.allclose(train.x[3].numpy() * train.std_x.numpy() + train.mean_x.numpy(),
                       np.delete(p, 9, axis=0).reshape(-1), atol=1e-2)
    s = raw_tr["S"][3]
    assert np.allclose(train.t[3].numpy() * train.std_t.numpy() + train.mean_t.numpy(),
                       (s - s[0])[1:].reshape(-1), atol=1e-1)


def test_actions_decode_like_the_reference():
    assert decode_action("S1_Directions_1.54138969_000001.jpg") == "Directions"
    assert decode_action("S11_WalkDog.60457274_000123.jpg") == "WalkDog"
    for n in ("S1_Directions_1.54138969_000001.jpg", "S9_SittingDown_2.58860488_000017.jpg",
              "S5_Photo.55011271_000999.jpg"):
        assert decode_action(n) == O.h36m_decode_action(n)
    raw = synthetic_raw(100, seed=3)
    ds = DevicePoseDataset(raw, "cpu")
    assert [ds.action_names[i] for i in ds.action_ids.tolist()] == [O.h36m_decode_action(n) for n in raw["image"]]
    assert set(ds.action_names) == set(ACTIONS)


def test_epoch_has_dataloader_semantics():
    ds = DevicePoseDataset(synthetic_raw(1000, seed=4), "cpu", seed=7)
    assert ds.num_batches(64) == 16 and ds.num_batches(64, drop_last=True) == 15
    plain = list(ds.epoch(0, 64))
    assert len(plain) == 16 and plain[-1][0].shape[0] == 1000 - 15 * 64       # last partial batch kept
    assert torch.equal(torch.cat([b[0] for b in plain]), ds.x)
    sh1 = torch.cat([b[0] for b in ds.epoch(1, 64, shuffle=True)])
    sh1b = torch.cat([b[0] for b in ds.epoch(1, 64, shuffle=True)])
    sh2 = torch.cat([b[0] for b in ds.epoch(2, 64, shuffle=True)])
    assert torch.equal(sh1, sh1b) and not torch.equal(sh1, sh2)              # reproducible, varies with epoch
    assert torch.equal(sh1.sort(0).values, ds.x.sort(0).values)               # a permutation of the split
    x, t, a = next(iter(ds.epoch(3, 32, shuffle=True, with_actions=True)))
    # rows stay aligned across x / t / action
    rows = [int((ds.x == x[i]).all(1).nonzero()[0]) for i in range(4)]
    assert torch.equal(ds.t[rows], t[:4]) and torch.equal(ds.action_ids[rows], a[:4])


def test_reference_pickles_round_trip(tmp_path):
    d = tmp_path / "Human3.6M"
    d.mkdir()
    raws = {"train": synthetic_raw(300, seed=5), "valid": synthetic_raw(90, seed=6)}
    for task, raw in raws.items():      # the reference stores lists of arrays (H36M/data.py:38)
        payload = {"part": [p for p in raw["part"]], "S": [s for s in raw["S"]], "image": raw["image"],
                   "center": raw["center"], "scale": raw["scale"]}
        with open(d / ("%s_GT.bin" % task), "wb") as f:
            pickle.dump(payload, f)
    train, valid = DevicePoseDataset.from_pickles(str(d), "cpu", protocol="GT")
    x, t, _, _ = _oracle_split(raws["train"], raws["valid"])
    np.testing.assert_allclose(valid.x.numpy(), x, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(valid.t.numpy(), t, rtol=2e-5, atol=2e-5)
    assert len(train) == 300 and len(valid) == 90
    with pytest.raises(ValueError):
        DevicePoseDataset({"part": np.zeros((4, 16, 2)), "S": np.zeros((4, 17, 3))}, "cpu")


@pytest.mark.parametrize("protocol", ["GT", "SH", "SH+FT"])
def test_from_pickles_selects_the_protocol_file(tmp_path, protocol):
    """/root/reference/H36M/protocol.py:1-4 + H36M/data.py:31-34: the protocol only selects which
    ``{task}_{protocol}.bin`` pair feeds the same path (GT: ground-truth 2D joints; SH / SH+FT:
    joints detected by the (fine-tuned) hourglass).  Synthetic pickles of the reference's layout,
    one pair per protocol with different 2D inputs and identical 3D targets."""
    base_tr, base_va = synthetic_raw(600, seed=5), synthetic_raw(200, seed=6)
    rng = np.random.RandomState(7)
    for proto, noise in (("GT", 0.0), ("SH", 6.0), ("SH+FT", 3.0)):
        for task, raw in (("train", base_tr), ("valid", base_va)):
            r = dict(raw)
            r["part"] = (raw["part"] + noise * rng.standard_normal(raw["part"].shape)).astype(np.float32).tolist()
            r["S"] = raw["S"].tolist()                 # the reference stores python lists
            with open(tmp_path / ("%s_%s.bin" % (task, proto)), "wb") as f:
                pickle.dump(r, f)
    train, valid = DevicePoseDataset.from_pickles(str(tmp_path), "cpu", protocol=protocol)
    with open(tmp_path / ("train_%s.bin" % protocol), "rb") as f:
        raw_tr = pickle.load(f)
    with open(tmp_path / ("valid_%s.bin" % protocol), "rb") as f:
        raw_va = pickle.load(f)
    x, t, mt, st = _oracle_split(raw_tr, raw_va)
    np.testing.assert_allclose(valid.x.numpy(), x, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(valid.t.numpy(), t, rtol=2e-5, atol=2e-5)
    assert len(train) == 600 and len(valid) == 200
    if protocol != "GT":     # detected joints differ from the ground truth, the 3D targets do not
        gt_train, _ = DevicePoseDataset.from_pickles(str(tmp_path), "cpu", protocol="GT")
        assert not torch.allclose(gt_train.x, train.x)
        assert torch.allclose(gt_train.t, train.t)
    with pytest.raises(ValueError):       # H36M/data.py:22 asserts protocol in {GT, SH, SH+FT}
        DevicePoseDataset.from_pickles(str(tmp_path), "cpu", protocol="nope")
    os.remove(tmp_path / ("valid_%s.bin" % protocol))
    with pytest.raises(FileNotFoundError):
        DevicePoseDataset.from_pickles(str(tmp_path), "cpu", protocol=protocol)
