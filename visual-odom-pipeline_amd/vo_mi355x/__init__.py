"""vo_mi355x -- MI355X-native visual-odometry inner loop (host side).

Python + numpy host code over a C-ABI shared library of hand-written HIP kernels
(include/vo_mi355x.h).  Mirrors the reference's `Extractor` / `BundleAdjuster`
interfaces (JonasFrey96/Visual-Odom-Pipeline, src/extractor, src/bundle_adjuster).
"""
from ._lib import LIB_PATH, VoError  # noqa: F401
from .bundle_adjuster import BundleAdjuster  # noqa: F401
from .context import VoContext  # noqa: F401
from .extractor import DMatch, Extractor  # noqa: F401
from .loader import Loader  # noqa: F401
from .state import Keypoint, Landmark, State, Trajectory  # noqa: F401

__all__ = ["VoContext", "VoError", "LIB_PATH", "Extractor", "DMatch", "BundleAdjuster", "Loader", "Keypoint", "Landmark", "State",
           "Trajectory"]
