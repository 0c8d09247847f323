"""MI355X-native GP-surrogate acquisition step (drop-in for the reference's PointSelector).

Importing the package does not load the HIP library; the first use of `PointSelector` /
`DeviceGP` does, and raises if it (or a GPU) is missing - there is no CPU fallback.
"""
from .point_selector import PointSelector  # noqa: F401
from .gp_device import DeviceGP, ScoreResult  # noqa: F401
from .host_binding import PointSelectorHost  # noqa: F401  (NumPy + ctypes only: no PyTorch needed)

__all__ = ["PointSelector", "PointSelectorHost", "DeviceGP", "ScoreResult"]
