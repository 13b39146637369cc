"""GPU tests aimed AT the discontinuities of the path, through the C-ABI, against the CPU oracle.

`k_fill3` replaces the reference's `sqrt`, `/`, `exp` by sequences of its own and takes the cut-off decision
`dist / vcd > cutoff_round_to_zero` (helpers.pyx:196-203) as ONE comparison of the squared distance with an exact
threshold computed on the host; the static-lattice check (helpers.pyx:76-80) compares squared distances first; the wraps
(util/PBCCalculator.pyx:341-366) hinge on `floor`.  Random trajectories never land on those edges, so these tests put
atoms there: an ion is bisected along a ray until the oracle's component flips between zero and non-zero, a static atom
until the oracle starts raising StaticLatticeError, atoms are placed on the cell faces - and then every neighbouring
double (+-8 ulps of a coordinate) is run through both sides.  Cells: diagonal (C2), hexagonal (C1) and triclinic (C1b).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CELLS = ["C2", "C1", "C1b"]


def _basis(cfg):
    from sitator_amd import _lib, synth
    host = synth.config_host(cfg)
    ref_static = np.ascontiguousarray(host.static_pos, dtype=np.float64)
    V = max(len(v) for v in host.vertices)
    verts = np.full((len(host.vertices), V), -1, dtype=np.int64)
    for k, v in enumerate(host.vertices):
        verts[k, :len(v)] = v
    ctx = _lib.HipContext(host.cell)
    vcd = ctx.site_vertex_distances(np.asarray(host.centers), ref_static, verts)
    ctx.set_basis(ref_static, verts, vcd, 1.5, 30, 1.0)
    return host, ctx, ref_static, verts, vcd


def _frames(ref_static, ions):
    """One frame per ion position: the statics at their reference positions, one mobile atom."""
    ions = np.asarray(ions, dtype=np.float64).reshape(-1, 3)
    fr = np.empty((len(ions), len(ref_static) + 1, 3))
    fr[:, :-1] = ref_static
    fr[:, -1] = ions
    return fr


def _neighbours(p, axis, n=8):
    """p with coordinate `axis` moved by -n .. +n ulps."""
    out = []
    for k in range(-n, n + 1):
        q = np.array(p, dtype=np.float64)
        x = q[axis]
        for _ in range(abs(k)):
            x = np.nextafter(x, np.inf if k > 0 else -np.inf)
        q[axis] = x
        out.append(q)
    return out


@pytest.mark.parametrize("cfg", CELLS)
def test_ions_on_the_cutoff_boundary(oracle, cfg):
    """For several landmarks: walk from the landmark's centre away from one of its vertices until the oracle's
    component becomes zero, bisect to adjacent doubles of the ray parameter, then try the 17 neighbouring doubles of
    each coordinate.  The GPU rows must have the oracle's zero pattern entry for entry (the exact-threshold compare
    IS the reference's decision) and its values within the float bar."""
    host, ctx, ref_static, verts, vcd = _basis(cfg)
    S = len(ref_static)
    sidx, midx = np.arange(S), np.array([S])

    def component(p, k):
        fr = _frames(ref_static, [p])
        lv, _ = oracle.fill(host.cell, oracle.wrap_points(host.cell, fr), sidx, midx, ref_static, verts, vcd,
                            check_for_zeros=False)
        return lv[0, k]

    rng = np.random.default_rng(5)
    ions, found = [], 0
    for k in rng.permutation(len(host.centers))[:12]:
        c = np.asarray(host.centers[k], dtype=np.float64)
        v = [x for x in verts[k] if x >= 0]
        d = c - ref_static[v[int(rng.integers(len(v)))]]
        d = d / np.linalg.norm(d) + rng.normal(scale=0.05, size=3)
        lo, hi = 0.0, 6.0
        if component(c + lo * d, k) == 0.0 or component(c + hi * d, k) != 0.0:
            continue
        while np.nextafter(lo, np.inf) < hi:
            mid = 0.5 * (lo + hi)
            if component(c + mid * d, k) != 0.0:
                lo = mid
            else:
                hi = mid
        assert component(c + lo * d, k) != 0.0 and component(c + hi * d, k) == 0.0
        found += 1
        for lam in (lo, hi):
            for axis in range(3):
                ions.extend(_neighbours(c + lam * d, axis))
    assert found >= 6, "too few boundary crossings found"
    fr = _frames(ref_static, ions)
    ctx.set_frames(fr, sidx, midx)
    rc, nz, err = ctx.fill(check_for_zeros=False)
    assert rc == 0 and ctx.info()["fill_kernel"] == 3
    exp, nz_exp = oracle.fill(host.cell, oracle.wrap_points(host.cell, fr), sidx, midx, ref_static, verts, vcd,
                              check_for_zeros=False)
    got = ctx.rows_dense()
    flips = int(np.sum((exp[1:] != 0) != (exp[:-1] != 0)))
    assert flips >= found, "the neighbourhoods must straddle the boundary"
    assert np.array_equal(got != 0, exp != 0), "zero pattern on the cut-off boundary differs from the oracle's"
    assert nz == nz_exp
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=0)
    # diagonal cell: the cut-off is decided on the logistic argument with an error band (~1e-11 wide); ions ON the edge
    # must have fallen into it and gone round again with the reference's arithmetic - counted, so that this test cannot
    # pass by the cheap decision landing on the right side by luck.  (General cells decide exactly throughout.)
    band = ctx.info()["band_redos"]
    assert (band > 0) if cfg == CELLS[0] else (band == 0), (cfg, band)


@pytest.mark.parametrize("cfg", CELLS)
def test_static_atoms_on_the_movement_threshold(oracle, cfg):
    """A static atom is moved away from its reference position until the oracle raises StaticLatticeError
    (helpers.pyx:76-80, distance > static_movement_threshold), bisected to adjacent doubles; for the 17 neighbouring
    doubles of each coordinate the GPU must raise exactly when the oracle does, with its frame and atom."""
    from sitator_amd import _lib
    host, ctx, ref_static, verts, vcd = _basis(cfg)
    S = len(ref_static)
    sidx, midx = np.arange(S), np.array([S])
    ion = np.asarray(host.centers[0], dtype=np.float64)

    def raises(fr):
        try:
            oracle.fill(host.cell, oracle.wrap_points(host.cell, fr), sidx, midx, ref_static, verts, vcd, check_for_zeros=False)
        except oracle.OracleError as e:
            assert e.kind == "StaticLatticeError"
            return (int(e.frame), int(e.lattice_atoms[0]))
        return None

    rng = np.random.default_rng(9)
    n_raise = n_ok = 0
    for atom in rng.permutation(S)[:3]:
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        base = _frames(ref_static, [ion])

        def moved(lam, axis=None, k=0):
            fr = base.copy()
            p = ref_static[atom] + lam * d
            if axis is not None:
                p = _neighbours(p, axis)[8 + k]
            fr[0, atom] = p
            return fr

        lo, hi = 0.5, 1.5
        assert raises(moved(lo)) is None and raises(moved(hi)) is not None
        while np.nextafter(lo, np.inf) < hi:
            mid = 0.5 * (lo + hi)
            if raises(moved(mid)) is None:
                lo = mid
            else:
                hi = mid
        for lam in (lo, hi):
            for axis in range(3):
                for k in range(-8, 9):
                    fr = moved(lam, axis, k)
                    want = raises(fr)
                    ctx.set_frames(fr, sidx, midx)
                    rc, nz, err = ctx.fill(check_for_zeros=False)
                    if want is None:
                        n_ok += 1
                        assert rc == 0, "the GPU raised where the oracle did not (lambda %r axis %d ulp %+d)" % (lam, axis, k)
                    else:
                        n_raise += 1
                        assert rc == _lib.E_STATIC_THRESHOLD and (int(err.frame), int(err.index)) == want
    assert n_raise >= 20 and n_ok >= 20, "both sides of the threshold must be exercised"


@pytest.mark.parametrize("cfg", CELLS)
def test_atoms_on_the_cell_faces(oracle, cfg):
    """Wraps hinge on floor(): points with a fractional coordinate of exactly 0, 1, -1, 2 and their neighbouring
    doubles (in Cartesian coordinates, each axis) must wrap to the oracle's bits (PBCCalculator.wrap_points), and a
    frame whose mobile ion and static atoms sit on faces must give the oracle's landmark vectors."""
    from sitator_amd import PBCCalculator
    host, ctx, ref_static, verts, vcd = _basis(cfg)
    cell = np.asarray(host.cell, dtype=np.float64)
    pts = []
    for f0 in (0.0, 1.0, -1.0, 2.0, 0.5):
        for f1 in (0.0, 1.0, 0.25):
            for f2 in (0.0, 1.0, 0.75):
                p = np.array([f0, f1, f2]) @ cell
                for axis in range(3):
                    pts.extend(_neighbours(p, axis, n=3))
    pts = np.array(pts)
    mine = pts.copy()
    PBCCalculator(cell).wrap_points(mine)
    assert np.array_equal(mine, oracle.wrap_points(cell, pts.reshape(1, -1, 3))[0]), "wrap on a cell face differs in bits"
    # frames: the ion on faces / edges / corners, and (second half) a static atom pushed onto the nearest face
    S = len(ref_static)
    sidx, midx = np.arange(S), np.array([S])
    ions = []
    for f in ((0.0, 0.3, 0.6), (1.0, 0.3, 0.6), (0.4, 0.0, 1.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), (0.7, 1.0, 0.0)):
        p = np.array(f) @ cell
        for axis in range(3):
            ions.extend(_neighbours(p, axis, n=2))
    fr = _frames(ref_static, ions)
    frac = ref_static @ np.linalg.inv(cell)
    height = 1.0 / np.linalg.norm(np.linalg.inv(cell), axis=0)          # perpendicular heights of the cell
    gap = np.minimum(frac % 1.0, 1.0 - frac % 1.0) * height              # distance of every static atom to its faces
    near = [a for a in np.argsort(gap.min(axis=1)) if gap[a].min() < 0.4][:len(fr) // 2]   # well below the threshold
    for i, a in enumerate(near):
        f = frac[a].copy()
        ax = int(np.argmin(gap[a]))
        f[ax] = np.round(f[ax])                      # exactly on the face
        fr[len(fr) // 2 + i, a] = f @ cell
    ctx.set_frames(fr, sidx, midx)
    rc, nz, err = ctx.fill(check_for_zeros=False)
    assert rc == 0
    exp, nz_exp = oracle.fill(cell, oracle.wrap_points(cell, fr), sidx, midx, ref_static, verts, vcd, check_for_zeros=False)
    got = ctx.rows_dense()
    assert nz == nz_exp
    assert np.array_equal(got != 0, exp != 0)
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=0)
