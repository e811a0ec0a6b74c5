"""bilinear_amd — MI355X (gfx950) native implementation of the one hot path of
nulledge/bilinear: the forward / backward / optimiser step of its 2D->3D
pose-lifting MLP (reference: model/bilinear.py + train_bilinear.py:75-83).

    from bilinear_amd.model import bilinear          # == reference `model.bilinear`
    net, opt, step, epoch = bilinear.load(device)    # same signature and returns

Everything numerical runs in libbilinear_hip.so (include/bilinear_hip.h); there
is no CPU or ATen fallback.
"""
from . import _native  # noqa: F401
from .model.bilinear import Bilinear, BilinearUnit, heavy_linear, load  # noqa: F401
from .graph import CapturedTrainStep  # noqa: F401
from .optim import Adam, clip_grad_norm_  # noqa: F401
from .loss_log import LossRing  # noqa: F401
from . import config  # noqa: F401



def configure_for_small_batches(enabled=True):
    """The reference's own loop (five calls per step, /root/reference/train_bilinear.py:75-83) at its own batch of 64
    is HOST-bound on this hardware: 0.15 ms of kernels per step against ~0.3 ms of Python and autograd-engine time.
    The largest single item is not ours: ``loss.backward()`` hands the graph to the autograd engine's device thread
    and waits for it — one operator deep, the hand-off (two thread wake-ups) costs more than the work.  This runs
    backward on the calling thread instead (``torch.autograd.set_multithreading_enabled(False)``: a process-wide
    PyTorch setting, which is why it is a call of the user's and not a default of this package): five-call step
    0.33 -> 0.26 ms at batch 64 (bench.py: batch_64.five_call_drop_in).  Returns the previous setting; results are
    bit-identical.  For batches where the step is GPU-bound (>= 1024 rows) it changes nothing measurable."""
    import torch
    previous = torch.autograd.is_multithreading_enabled()
    torch.autograd.set_multithreading_enabled(not enabled)
    return previous


__all__ = ["configure_for_small_batches", "BilinearUnit", "Bilinear", "heavy_linear", "load", "Adam", "clip_grad_norm_",
           "CapturedTrainStep", "LossRing", "config"]
