"""Progress reporting, switched as in the reference (``sitator/util/progress.py:3-14``): the environment variable
``SITATOR_PROGRESSBAR`` (``true`` / ``yes`` / ``on``, the default, or anything else for off) decides whether the
long loops show a ``tqdm`` bar.

The loops the reference wraps in ``tqdm`` - over the frames in ``helpers._fill_landmark_vectors`` (``landmark/helpers.pyx:50``,
"Landmark Frame"), over the samples in the clustering - are single kernel launches here, so a bar has one tick: every
stage reports ONE line in ``tqdm``'s format when it is done (to stderr, like ``tqdm``), and nothing when the switch is
off.  ``tqdm(iterable, **kwargs)`` itself is exported with the reference's semantics for code written against it.
"""
import os
import sys

_flag = os.getenv("SITATOR_PROGRESSBAR", "true").lower()
progress = _flag in ("true", "yes", "on")

if progress:
    try:
        from tqdm.autonotebook import tqdm
    except Exception:     # noqa: BLE001 - as the reference: no tqdm, no bars
        def tqdm(iterable, **kwargs):
            return iterable
else:
    def tqdm(iterable, **kwargs):
        return iterable


def enabled():
    """Read per call (a test or a notebook may flip the variable between runs)."""
    return os.getenv("SITATOR_PROGRESSBAR", "true").lower() in ("true", "yes", "on")


def stage(desc, total, seconds, unit="it"):
    """One finished stage in ``tqdm``'s line format: ``desc: 100%|##########| total/total [seconds, rate]``."""
    if not enabled():
        return
    rate = total / seconds if seconds > 0 else float("inf")
    sys.stderr.write("%s: 100%%|##########| %d/%d [%.3fs, %.3g%s/s]\n" % (desc, total, total, seconds, rate, unit))
    sys.stderr.flush()
