"""The values of the reference's ``util/config.py`` that the hot path reads (/root/reference/util/config.py:13-25;
``dotmap`` is not importable here, so a tiny attribute namespace stands in for ``DotMap``): ONE place for the batch
size and the learning-rate decay hook, imported by ``train_bilinear.py`` and by the tests — the same attribute
paths as the reference, ``config.bilinear.lr_decay.condition(step)`` / ``.function(step)``
(/root/reference/train_bilinear.py:66-70)."""
from types import SimpleNamespace


def _lr_decay_condition(step):
    """util/config.py:21 — the hook fires on the PRE-increment step: at step 1 and every 100000 steps."""
    return step % 100000 == 0 or step == 1


def _lr_decay_function(step):
    """util/config.py:22 — 1e-3 * 0.96 ** (step / 100000)."""
    return 1.0e-3 * 0.96 ** (step / 100000)


bilinear = SimpleNamespace(
    comment="Bilinear GT",                 # util/config.py:14
    batch_size=64,                         # :15
    num_workers=8,                         # :16 (the device-resident pipeline has no workers)
    data_dir="data/Human3.6M",             # :18
    lr_decay=SimpleNamespace(activate=True, condition=_lr_decay_condition, function=_lr_decay_function),   # :19-23
    protocol="GT",                         # :24 (H36M/protocol.py:1-4)
)
