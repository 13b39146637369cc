"""TEST INFRASTRUCTURE ONLY - builds the *true* reference (Cython + Python) so that
golden fixtures can be generated from it and the CPU restatement can be pinned.

Works only in the development container, where ``/root/reference`` exists; on the
GPU box it fails soft (``available()`` is False).  Nothing of the reference is
copied into this repository: a scratch tree of *symlinks* to the reference's
``.py``/``.pyx`` files is made under ``$TMPDIR`` and the four hot-path ``.pyx``
modules are cythonized there (SURVEY.md §8c recipe; the stock ``setup.py`` cannot
be used because ``misc/GenerateClampedTrajectory.pyx`` is rejected by Cython 3).

Harness-side shims (this repo's code, the reference is untouched):
  * ``oracle/ase_stub``: a data-holder ``ase.Atoms`` (positions/cell/numbers only;
    no arithmetic on the path goes through it) because ``ase`` is not installed;
  * ``np.int/np.float/np.bool`` aliases removed in numpy >= 1.24;
  * ``SITATOR_PROGRESSBAR=false``, matplotlib ``Agg``.
"""
import os
import subprocess
import sys
import tempfile

REFERENCE = os.environ.get("SITATOR_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
PYX = ["landmark/helpers.pyx", "util/PBCCalculator.pyx", "util/DotProdClassifier.pyx",
       "util/RecenterTrajectory.pyx", "dynamics/SmoothSiteTrajectory.pyx"]


def available():
    return os.path.isdir(os.path.join(REFERENCE, "sitator"))


def scratch_dir():
    return os.path.join(os.environ.get("TMPDIR", tempfile.gettempdir()), "sitator_ref_build")


def build(force=False):
    """Create the symlink tree and compile the .pyx files. Returns the sys.path entry."""
    if not available():
        raise RuntimeError("reference not present at %s" % REFERENCE)
    root = scratch_dir()
    stamp = os.path.join(root, ".built")
    if os.path.exists(stamp) and not force:
        return root
    src = os.path.join(REFERENCE, "sitator")
    for dirpath, _, files in os.walk(src):
        rel = os.path.relpath(dirpath, src)
        dst = os.path.join(root, "sitator", rel)
        os.makedirs(dst, exist_ok=True)
        for f in files:
            if f.endswith((".py", ".pyx", ".pxd")):
                link = os.path.join(dst, f)
                if not os.path.lexists(link):
                    os.symlink(os.path.join(dirpath, f), link)
    script = (
        "import sys, numpy as np, Cython.Compiler.Options as O\n"
        "O.cimport_from_pyx = True\n"
        "from setuptools import setup\n"
        "from Cython.Build import cythonize\n"
        "setup(name='sitator_ref', script_args=['build_ext', '--inplace', '-q'],\n"
        "      ext_modules=cythonize(%r, language_level=3, quiet=True),\n"
        "      include_dirs=[np.get_include()])\n" % [os.path.join("sitator", p) for p in PYX])
    subprocess.check_call([sys.executable, "-c", script], cwd=root,
                          stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)
    open(stamp, "w").write("ok\n")
    return root


def import_reference():
    """Import and return the true reference package ``sitator``."""
    import numpy as np
    for name, typ in (("int", int), ("float", float), ("bool", bool)):
        if not hasattr(np, name):
            setattr(np, name, typ)
    os.environ["SITATOR_PROGRESSBAR"] = "false"
    import matplotlib
    matplotlib.use("Agg")
    root = build()
    for p in (os.path.join(HERE, "ase_stub"), root):
        if p not in sys.path:
            sys.path.insert(0, p)
    import sitator
    return sitator


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
