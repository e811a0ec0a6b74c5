"""Top-level alias so reference scripts that do ``import model`` /
``model.bilinear.load(...)`` (/root/reference/train_bilinear.py:9,45) pick up the
MI355X implementation unchanged.  The hourglass CNN of the reference's ``model``
package is out of scope and not provided."""
from bilinear_amd.model import bilinear  # noqa: F401
from bilinear_amd.model.bilinear import Bilinear, BilinearUnit, heavy_linear, load  # noqa: F401
