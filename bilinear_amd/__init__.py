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

__all__ = ["BilinearUnit", "Bilinear", "heavy_linear", "load", "Adam", "clip_grad_norm_",
           "CapturedTrainStep", "LossRing", "config"]
