"""Mirror of the reference's ``model`` package for the lifter
(/root/reference/model/__init__.py star-imports model/bilinear.py)."""
from . import bilinear  # noqa: F401
from .bilinear import Bilinear, BilinearUnit, heavy_linear, load  # noqa: F401
