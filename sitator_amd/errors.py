"""Exception contract of the landmark path.

Same class names, attributes and messages' meaning as the reference
(``sitator/landmark/errors.py:5-40``, ``sitator/errors.py:6-21``), so ``except`` clauses and
attribute reads written against sitator keep working.
"""


class LandmarkAnalysisError(Exception):
    pass


class StaticLatticeError(LandmarkAnalysisError):
    """Static-lattice atoms moved beyond ``static_movement_threshold`` or were left unmatched.

    Attributes: ``lattice_atoms`` (indices into the static lattice), ``frame``.
    """
    TRY_RECENTERING_MSG = "Try recentering the input trajectory (sitator.util.RecenterTrajectory)"

    def __init__(self, message, lattice_atoms=None, frame=None, try_recentering=False):
        if try_recentering:
            message = message + "\n" + StaticLatticeError.TRY_RECENTERING_MSG
        super(StaticLatticeError, self).__init__(message)
        self.lattice_atoms = lattice_atoms
        self.frame = frame


class ZeroLandmarkError(LandmarkAnalysisError):
    """An all-zero landmark vector. Attributes: ``mobile_index``, ``frame``."""

    def __init__(self, mobile_index, frame):
        super(ZeroLandmarkError, self).__init__(
            "Encountered a zero landmark vector for mobile ion %i at frame %i. Try increasing "
            "`cutoff_midpoint` and/or decreasing `cutoff_steepness`." % (mobile_index, frame))
        self.mobile_index = mobile_index
        self.frame = frame


class SiteAnaysisError(Exception):
    """An error occuring as part of site analysis (spelling as in the reference)."""
    pass


class MultipleOccupancyError(SiteAnaysisError):
    """Several mobile atoms on one site in one frame. Attributes: ``mobile_particles``, ``site``, ``frame``."""

    def __init__(self, mobile, site, frame):
        super(MultipleOccupancyError, self).__init__(
            "Multiple mobile particles %s were assigned to site %i at frame %i." % (mobile, site, frame))
        self.mobile_particles = mobile
        self.site = site
        self.frame = frame


class InsufficientSitesError(SiteAnaysisError):
    """Fewer sites than mobile particles. Attributes: ``n_sites``, ``n_mobile``.

    (The reference means to raise this at ``LandmarkAnalysis.py:266-271`` but never imports the
    name, so it surfaces there as a ``NameError``; here the intended exception is raised.)"""

    def __init__(self, verb, n_sites, n_mobile):
        super(InsufficientSitesError, self).__init__(
            "%s resulted in only %i sites for %i mobile particles." % (verb, n_sites, n_mobile))
        self.n_sites = n_sites
        self.n_mobile = n_mobile


class DeviceDomainError(Exception):
    """Internal: a domain error reported by the C-ABI before it is mapped to the classes above."""

    def __init__(self, kind, frame, index):
        super(DeviceDomainError, self).__init__("device domain error kind=%d frame=%d index=%d" % (kind, frame, index))
        self.kind = kind
        self.frame = frame
        self.index = index
