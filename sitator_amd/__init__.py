"""sitator_amd: the landmark-analysis hot path of sitator, MI355X-native.

Host code is Python and talks to hand-written HIP kernels (gfx950) through the ctypes C-ABI of
``include/sitator_hip.h``.  The public names mirror the reference package for this path::

    from sitator_amd import SiteNetwork, SiteTrajectory, LandmarkAnalysis
    st = LandmarkAnalysis(clustering_algorithm="dotprod").run(sn, frames)
"""
from .errors import (InsufficientSitesError, LandmarkAnalysisError, MultipleOccupancyError,  # noqa: F401
                     StaticLatticeError, ZeroLandmarkError)
from .site_network import SiteNetwork, Structure  # noqa: F401
from .site_trajectory import SiteTrajectory  # noqa: F401
from .pbc import PBCCalculator  # noqa: F401
from .dotprod_classifier import DotProdClassifier, LandmarkVectors  # noqa: F401
from .landmark import LandmarkAnalysis  # noqa: F401
from .dynamics import JumpAnalysis, MergeSitesByDynamics, SmoothSiteTrajectory  # noqa: F401
from .merging import MergeSites, MergeSitesError, MergedSitesTooDistantError  # noqa: F401
from .recenter import RecenterTrajectory  # noqa: F401

__version__ = "0.1.0"
