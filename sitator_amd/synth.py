"""Synthetic host lattices and trajectories for the landmark-analysis path.

The reference ships no data and no tests; its example notebooks live
off-repo.  These generators provide the inputs that the oracle, the golden fixtures,
the parity tests and ``bench.py`` all share:

* ``sc_grid``  - throughput family: statics on a simple-cubic grid at fractional
  ``(g + 0.25)/G``, one landmark per grid cell at ``(g + 0.75)/G`` whose vertices are
  the 8 surrounding statics.  Every ``G`` must be >= 3.
* ``bcc_tet``  - parity family: BCC statics, 12 tetrahedral 4-vertex landmarks per
  cubic cell, face-sharing tetrahedra => landmark vectors with ~7 non-zeros.
* ``fcc_mixed`` - ragged family: FCC statics, 8 tetrahedral (4 vertices) + 4
  octahedral (6 vertices) landmarks per cubic cell.

``make_trajectory`` adds the dynamics: thermal jitter on statics and ions, rare hops
of an ion to an unoccupied neighbouring landmark centre along a linear 5-frame
transit (which yields genuine unassigned/transition samples), and unwrapped
coordinates (ions that cross the cell keep going), so wrapping is exercised.

Pure numpy, deterministic for a given seed.
"""
from collections import namedtuple

import numpy as np

Host = namedtuple("Host", ["cell", "static_pos", "centers", "vertices", "name"])
Host.__doc__ = """A static host lattice plus its landmark basis.

cell        (3,3) rows are cell vectors (ASE convention, as ``sn.structure.cell``)
static_pos  (S,3) reference positions of the static atoms
centers     (D,3) landmark (Voronoi-node-like) centres
vertices    list of D lists of static indices (statics-only numbering, SURVEY.md a15)
"""

ORTHO = "orthorhombic"


def _frac_to_real(frac, cell):
    return np.asarray(frac, dtype=np.float64) @ np.asarray(cell, dtype=np.float64)


def mic_displacement(d, cell):
    """Minimum-image displacement by brute force over 27 (or 125) images."""
    cell = np.asarray(cell, dtype=np.float64)
    inv = np.linalg.inv(cell)
    f = d @ inv
    f -= np.round(f)
    base = f @ cell
    rng = np.arange(-1, 2)
    shifts = np.array([[i, j, k] for i in rng for j in rng for k in rng], dtype=np.float64) @ cell
    cand = base[..., None, :] + shifts
    n2 = np.einsum("...k,...k->...", cand, cand)
    best = np.argmin(n2, axis=-1)
    return np.take_along_axis(cand, best[..., None, None], axis=-2)[..., 0, :]


def sc_grid(G=(3, 3, 3), cell=None, spacing=4.0):
    """Simple-cubic-grid host (SURVEY.md §8d ``SCgrid``)."""
    G = tuple(int(g) for g in G)
    if min(G) < 3:
        raise ValueError("every G must be >= 3 (neighbouring landmarks degenerate otherwise)")
    if cell is None:
        cell = np.diag([spacing * g for g in G]).astype(np.float64)
    cell = np.asarray(cell, dtype=np.float64)
    gx, gy, gz = np.meshgrid(np.arange(G[0]), np.arange(G[1]), np.arange(G[2]), indexing="ij")
    g = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1)
    Gf = np.asarray(G, dtype=np.float64)
    static_pos = _frac_to_real((g + 0.25) / Gf, cell)
    centers = _frac_to_real((g + 0.75) / Gf, cell)

    def sidx(a, b, c):
        return ((a % G[0]) * G[1] + (b % G[1])) * G[2] + (c % G[2])

    vertices = []
    for a, b, c in g:
        vertices.append([int(sidx(a + dx, b + dy, c + dz))
                         for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)])
    return Host(cell, static_pos, centers, vertices, "SCgrid%s" % (G,))


def _nearest_vertices(centers, static_pos, cell, n):
    out = []
    for c in centers:
        d = mic_displacement(static_pos - c, cell)
        r = np.sqrt(np.einsum("ij,ij->i", d, d))
        order = np.argsort(np.round(r, 9), kind="stable")
        out.append([int(i) for i in order[:n]])
    return out


_BCC_TET = np.array([
    (0, .5, .25), (0, .5, .75), (.5, 0, .25), (.5, 0, .75),
    (.25, 0, .5), (.75, 0, .5), (.25, .5, 0), (.75, .5, 0),
    (0, .25, .5), (0, .75, .5), (.5, .25, 0), (.5, .75, 0)])


def bcc_tet(G=3, a=4.2, shape=None):
    """BCC host with tetrahedral-interstitial landmarks (SURVEY.md §8d ``BCCtet``).

    ``shape`` is a (3,3) matrix multiplying the cubic cell rows (e.g. a shear for a
    triclinic cell, or diag(1, 1.1, 1.2) for orthorhombic)."""
    G = int(G)
    cell = np.eye(3) * (a * G)
    if shape is not None:
        cell = np.asarray(shape, dtype=np.float64) * (a * G)
    gx, gy, gz = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij")
    g = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1).astype(np.float64)
    off = 0.1
    sfrac = np.concatenate([(g + off) / G, (g + 0.5 + off) / G])
    lfrac = ((g[:, None, :] + _BCC_TET[None, :, :] + off) / G).reshape(-1, 3)
    static_pos = _frac_to_real(sfrac, cell)
    centers = _frac_to_real(lfrac % 1.0, cell)
    vertices = _nearest_vertices(centers, static_pos, cell, 4)
    return Host(cell, static_pos, centers, vertices, "BCCtet(%d,%g)" % (G, a))


_FCC_BASIS = np.array([(0, 0, 0), (0, .5, .5), (.5, 0, .5), (.5, .5, 0)])
_FCC_TET = np.array([(x, y, z) for x in (.25, .75) for y in (.25, .75) for z in (.25, .75)])
_FCC_OCT = np.array([(.5, .5, .5), (.5, 0, 0), (0, .5, 0), (0, 0, .5)])


def fcc_mixed(G=3, a=5.0, shape=None):
    """FCC host with ragged vertex lists: tetrahedral (4) + octahedral (6) landmarks."""
    G = int(G)
    cell = np.eye(3) * (a * G)
    if shape is not None:
        cell = np.asarray(shape, dtype=np.float64) * (a * G)
    gx, gy, gz = np.meshgrid(np.arange(G), np.arange(G), np.arange(G), indexing="ij")
    g = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1).astype(np.float64)
    off = 0.1
    sfrac = ((g[:, None, :] + _FCC_BASIS[None] + off) / G).reshape(-1, 3)
    tfrac = ((g[:, None, :] + _FCC_TET[None] + off) / G).reshape(-1, 3)
    ofrac = ((g[:, None, :] + _FCC_OCT[None] + off) / G).reshape(-1, 3)
    static_pos = _frac_to_real(sfrac % 1.0, cell)
    tcent = _frac_to_real(tfrac % 1.0, cell)
    ocent = _frac_to_real(ofrac % 1.0, cell)
    vertices = _nearest_vertices(tcent, static_pos, cell, 4) + \
        _nearest_vertices(ocent, static_pos, cell, 6)
    centers = np.concatenate([tcent, ocent])
    return Host(cell, static_pos, centers, vertices, "FCC(%d,%g)" % (G, a))


def hexagonal_cell(a=12.0, c=12.0):
    return np.array([[a, 0, 0], [-0.5 * a, a * np.sqrt(3) / 2, 0], [0, 0, c]])


def _neighbour_table(host, n_shell_tol=1.15):
    """For every landmark: the landmarks in its first neighbour shell (MIC) and the
    MIC displacement towards each."""
    D = len(host.centers)
    nbrs, disps = [], []
    for k in range(D):
        d = mic_displacement(host.centers - host.centers[k], host.cell)
        r = np.sqrt(np.einsum("ij,ij->i", d, d))
        r[k] = np.inf
        sel = np.where(r <= r.min() * n_shell_tol)[0]
        nbrs.append(sel)
        disps.append(d[sel])
    return nbrs, disps


class TrajectoryGenerator(object):
    """Stateful generator so long trajectories can be produced block by block.

    frames are float64 C-contiguous ``[F, A, 3]``; atom order is statics (S) then
    mobiles (M) then ``n_spectator`` extra atoms belonging to neither mask, unless
    ``interleave`` shuffles the atom order (masks are returned either way).
    """

    NOISE_BLOCK = 256

    def __init__(self, host, n_mobile, seed=0, sigma_static=0.05, sigma_ion=0.12,
                 p_hop=None, transit=5, n_spectator=0, interleave=False,
                 min_ion_separation=None, threads=8):
        self.host = host
        self.seed = int(seed)
        self.threads = int(threads)
        self.S = len(host.static_pos)
        self.D = len(host.centers)
        self.M = int(n_mobile)
        if self.M > self.D // 2:
            raise ValueError("too many mobile ions for %d landmarks" % self.D)
        self.sigma_static = float(sigma_static)
        self.sigma_ion = float(sigma_ion)
        self.p_hop = (1.0 / (50.0 * self.M)) if p_hop is None else float(p_hop)
        self.transit = int(transit)
        self.n_spec = int(n_spectator)
        self.A = self.S + self.M + self.n_spec
        self.rng = np.random.Generator(np.random.Philox(key=int(seed)))
        self.nbrs, self.disps = _neighbour_table(host)
        self.min_sep = min_ion_separation
        # initial occupation: distinct, mutually non-neighbouring where possible
        order = self.rng.permutation(self.D)
        occ = np.zeros(self.D, dtype=bool)
        site = []
        for k in order:
            if len(site) == self.M:
                break
            if occ[k] or occ[self.nbrs[k]].any():
                continue
            site.append(k)
            occ[k] = True
        for k in order:  # relax the neighbour rule if it could not be met
            if len(site) == self.M:
                break
            if not occ[k]:
                site.append(k)
                occ[k] = True
        self.site = np.array(site, dtype=np.int64)
        self.site0 = self.site.copy()
        self.occ = occ
        self.pos = host.centers[self.site].copy()         # unwrapped base position per ion
        self.t_left = np.zeros(self.M, dtype=np.int64)    # frames of transit remaining
        self.step = np.zeros((self.M, 3))                 # per-frame transit step
        if interleave:
            self.perm = self.rng.permutation(self.A)
        else:
            self.perm = np.arange(self.A)
        inv = np.empty(self.A, dtype=np.int64)
        inv[self.perm] = np.arange(self.A)
        self.static_mask = np.zeros(self.A, dtype=bool)
        self.mobile_mask = np.zeros(self.A, dtype=bool)
        # atom p of the emitted frame is canonical atom perm[p]
        self.static_mask[inv[:self.S]] = True
        self.mobile_mask[inv[self.S:self.S + self.M]] = True
        self.frame0 = 0

    # the reference structure (ideal lattice + ions at their initial sites)
    def reference_positions(self):
        ref = np.zeros((self.A, 3))
        ref[:self.S] = self.host.static_pos
        ref[self.S:self.S + self.M] = self.host.centers[self.site0]
        if self.n_spec:
            ref[self.S + self.M:] = self.host.centers[-self.n_spec:] + 0.3
        return ref[self.perm]

    def _can_land(self, j, k):
        if self.occ[k]:
            return False
        if self.min_sep is None:
            return True
        others = np.delete(np.arange(self.M), j)
        d = mic_displacement(self.host.centers[self.site[others]] - self.host.centers[k], self.host.cell)
        return bool(np.all(np.einsum("ij,ij->i", d, d) > self.min_sep ** 2))

    def generate(self, n_frames):
        F, M, S = int(n_frames), self.M, self.S
        base = np.empty((F, M, 3))
        u = self.rng.random((F, M))
        hop_frames = np.nonzero((u < self.p_hop).any(axis=1))[0]
        cur = 0
        # Between hop-candidate frames ions either sit still or finish a transit.
        def advance(f_to):
            nonlocal cur
            while cur < f_to:
                moving = self.t_left > 0
                if not moving.any():
                    base[cur:f_to] = self.pos
                    cur = f_to
                    return
                self.pos[moving] += self.step[moving]
                self.t_left[moving] -= 1
                base[cur] = self.pos
                cur += 1
        for f in hop_frames:
            advance(f)
            for j in np.nonzero(u[f] < self.p_hop)[0]:
                if self.t_left[j] > 0:
                    continue
                cand = self.nbrs[self.site[j]]
                pick = self.rng.permutation(len(cand))
                for c in pick:
                    k = cand[c]
                    if self._can_land(j, k):
                        self.occ[self.site[j]] = False
                        self.occ[k] = True
                        self.step[j] = self.disps[self.site[j]][c] / self.transit
                        self.t_left[j] = self.transit
                        self.site[j] = k
                        break
            advance(f + 1)
        advance(F)
        frames = np.empty((F, self.A, 3))
        identity = bool((self.perm == np.arange(self.A)).all())
        NB = self.NOISE_BLOCK
        if self.frame0 % NB:
            raise ValueError("generate() calls must start on a multiple of %d frames" % NB)
        spec0 = (self.host.centers[-self.n_spec:] + 0.3) if self.n_spec else None

        def noise_block(b):
            lo, hi = b * NB, min(F, (b + 1) * NB)
            rng = np.random.Generator(np.random.SFC64([self.seed, 0x5eed, self.frame0 // NB + b]))
            blk = frames[lo:hi] if identity else np.empty((hi - lo, self.A, 3))
            rng.standard_normal(out=blk.reshape(-1))
            st = blk[:, :S]
            st *= self.sigma_static
            st += self.host.static_pos
            mo = blk[:, S:S + M]
            mo *= self.sigma_ion
            mo += base[lo:hi]
            if self.n_spec:
                sp = blk[:, S + M:]
                sp *= 0.5
                sp += spec0
            if not identity:
                frames[lo:hi] = blk[:, self.perm]

        nblk = (F + NB - 1) // NB
        if nblk > 4 and self.threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(self.threads) as ex:
                list(ex.map(noise_block, range(nblk)))
        else:
            for b in range(nblk):
                noise_block(b)
        self.frame0 += F
        return frames


def make_trajectory(host, n_mobile, n_frames, seed=0, **kw):
    """One-shot convenience: returns (frames[F,A,3], static_mask, mobile_mask, ref_positions[A,3])."""
    gen = TrajectoryGenerator(host, n_mobile, seed=seed, **kw)
    ref = gen.reference_positions()
    frames = gen.generate(n_frames)
    return frames, gen.static_mask.copy(), gen.mobile_mask.copy(), ref


# ---- the named configurations of BASELINE.json / SURVEY.md §8d -----------------

def config_host(name):
    if name == "C1":
        return sc_grid((3, 3, 3), cell=hexagonal_cell(12.0, 12.0))
    if name == "C1d":       # the C1 host on a DIAGONAL cell (k_fill3's minimum-image arithmetic; C1 and C1b take the general one)
        return sc_grid((3, 3, 3), cell=np.diag([12.0, 13.2, 14.4]))
    if name == "C1b":
        shear = np.array([[1, 0, 0], [-0.15, 1, 0], [0.1, -0.08, 1.0]])
        return bcc_tet(3, 4.2, shape=shear)
    if name == "C2":
        return sc_grid((8, 8, 8), cell=np.diag([32.0, 35.2, 38.4]))
    if name == "C3":
        return sc_grid((8, 8, 17), cell=np.diag([26.0, 26.0, 55.25]))
    if name == "C4":
        return sc_grid((16, 16, 8), cell=np.diag([64.0, 70.4, 38.4]))
    if name == "C5":
        return fcc_mixed(G=4, a=5.4, shape=np.diag([1.0, 1.0, 1.45]))
    # the C2 shape on non-orthogonal cells (full 3x3 wraps in the kernels): hexagonal like C1, triclinic like C1b
    if name == "C2h":
        return sc_grid((8, 8, 8), cell=hexagonal_cell(32.0, 35.2))
    if name == "C2t":
        return sc_grid((8, 8, 8), cell=np.array([[32.0, 0, 0], [-5.3, 34.8, 0], [4.0, -2.7, 38.0]]))
    raise KeyError(name)


CONFIG_MOBILE = {"C1": 4, "C1d": 4, "C1b": 4, "C2": 64, "C3": 448, "C4": 256, "C5": 160, "C2h": 64, "C2t": 64}
CONFIG_FRAMES = {"C1": 2000, "C1d": 2000, "C1b": 1000, "C2": 100000, "C3": 250000, "C4": 1000000, "C5": 500000, "C2h": 100000,
                 "C2t": 100000}
CONFIG_SEED = {"C1": 1, "C1d": 21, "C1b": 11, "C2": 2, "C3": 3, "C4": 4, "C5": 5, "C2h": 12, "C2t": 13}
CONFIG_TEXT = {
    "C1": "C1: LiAlSiO4-like hexagonal cell a=b=12 A, c=12 A, SCgrid(3,3,3), S=D=27 (V=8), M=4 (BASELINE configs[0])",
    "C1d": "C1d: SCgrid(3,3,3) on an orthorhombic cell 12.0x13.2x14.4 A, S=D=27 (V=8), M=4 (the C1 host on a diagonal cell)",
    "C1b": "C1b: triclinic BCCtet(3, 4.2 A), S=54, D=324 (V=4), M=4 (the rich-overlap parity host)",
    "C2": "C2: SCgrid(8,8,8) orthorhombic 32.0x35.2x38.4 A, S=D=512 (V=8), M=64, A=576 (BASELINE configs[1])",
    "C3": "C3: LLZO-like SCgrid(8,8,17) 26.0x26.0x55.25 A, S=D=1088 (V=8), M=448, A=1536 (BASELINE configs[2])",
    "C4": "C4: SCgrid(16,16,8) orthorhombic 64.0x70.4x38.4 A, S=D=2048 (V=8), M=256, A=2304 (BASELINE configs[3])",
    "C5": "C5: LGPS-like tetragonal FCC host with tetrahedral (V=4) and octahedral (V=6) landmarks, ragged, S=256, "
          "D=768, M=160 (BASELINE configs[4])",
    "C2h": "C2h: the C2 shape on a hexagonal cell a=b=32.0 A, gamma=120, c=35.2 A (full 3x3 wraps), S=D=512, M=64",
    "C2t": "C2t: the C2 shape on a triclinic cell (32,0,0),(-5.3,34.8,0),(4,-2.7,38) A, S=D=512, M=64",
}
