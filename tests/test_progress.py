"""CPU: the reference's progress switch (sitator/util/progress.py:3-14, landmark/helpers.pyx:50) - SITATOR_PROGRESSBAR
decides whether the stages of LandmarkAnalysis.run report; `verbose` governs the clustering algorithm's own line."""
import numpy as np
import pytest

from tests import golden_util as G


def _run(monkeypatch, **kw):
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, _lib
    from tests.fake_ctx import FakeContext
    monkeypatch.setattr(_lib, "HipContext", FakeContext)        # no GPU here: the oracle-backed test double
    c = G.Case("c1_hex_scgrid")
    sn = SiteNetwork(Structure(c.ref_positions, c.cell), c.static_mask, c.mobile_mask)
    sn.centers = c.centers
    sn.vertices = c.vertices
    st = LandmarkAnalysis(**kw).run(sn, c.frames[:200])
    return st


@pytest.mark.parametrize("flag,verbose,lines", [("true", True, 2), ("on", False, 1), ("false", True, 0), ("0", True, 0)])
def test_progress_switch(monkeypatch, capsys, flag, verbose, lines):
    monkeypatch.setenv("SITATOR_PROGRESSBAR", flag)
    _run(monkeypatch, verbose=verbose)
    err = capsys.readouterr().err
    got = [l for l in err.splitlines() if "100%|" in l]
    assert len(got) == lines, err
    if lines:
        assert got[0].startswith("Landmark Frame: 100%|") and "200/200" in got[0]
    if lines == 2:
        assert got[1].startswith("Clustering (dotprod)")


def test_tqdm_export_follows_the_switch(monkeypatch):
    import importlib
    monkeypatch.setenv("SITATOR_PROGRESSBAR", "false")
    from sitator_amd import progress
    importlib.reload(progress)
    assert list(progress.tqdm(range(3), desc="x")) == [0, 1, 2] and not progress.progress
