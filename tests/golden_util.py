"""Loading helpers for the golden fixtures in tests/golden (data generated from the true
reference by oracle/make_fixtures.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PIPELINE_CASES = ["c1_hex_scgrid", "c1b_tri_bcctet", "c2_cut_ortho", "c5_cut_fcc_ragged", "bcc_ortho",
                  "c1_variants", "c1_static_swap", "c1_zero_lvecs", "err_static_threshold",
                  "err_static_unassigned", "err_multiple_occupancy", "err_insufficient_sites"]
# Long cuts of the BASELINE configurations (hops, unassigned transition samples, late clusters, jumps).  Their frames
# are regenerated from the stored recipe and checked against the stored digest; landmark vectors are stored for a
# leading block of frames only.  C3 / C4 are too heavy for the dense CPU oracle and are GPU-only.
LONG_CASES = ["c2_long_ortho", "c5_long_fcc_ragged", "c3_long", "c4_long", "c2h_long", "c2t_long"]
LONG_CASES_CPU = ["c2_long_ortho", "c5_long_fcc_ragged", "c2h_long", "c2t_long"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def vertices_of(verts_np):
    return [[int(v) for v in row if v >= 0] for row in verts_np]


class Case(object):
    def __init__(self, name):
        z = load(name)
        self.name = name
        self.z = z
        self.cell = z["cell"]
        self.ref_positions = z["ref_positions"]
        self.static_mask = z["static_mask"]
        self.mobile_mask = z["mobile_mask"]
        self.centers = z["centers"]
        self.verts_np = z["verts_np"]
        self.vertices = vertices_of(self.verts_np)
        self.site_vert_dists = z["site_vert_dists"]
        if "frames" in z.files:
            self.frames = z["frames"]
        else:
            self.frames = regenerate_frames(z)
        self.wrapped_head = z["wrapped_head"]
        self.tags = [str(t) for t in z["tags"]]

    def kwargs(self, tag):
        return json.loads(str(self.z[tag + "/kwargs"]))

    def out(self, tag):
        pre = tag + "/"
        return {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre) and not k.endswith("/kwargs")}


def regenerate_frames(z):
    """Frames of a long case from its recipe (synth.make_trajectory arguments); the digest must match."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sitator_amd import synth
    rec = json.loads(str(z["frames_recipe"]))
    frames, sm, mm, ref = synth.make_trajectory(synth.config_host(rec["config"]), rec["n_mobile"], rec["n_frames"],
                                                seed=rec["seed"], **rec["kw"])
    got = hashlib.sha256(np.ascontiguousarray(frames).tobytes()).hexdigest()
    if got != str(z["frames_sha256"]) or not np.array_equal(frames[:2], z["frames_head"]):
        raise AssertionError("regenerated frames of a golden case differ from the ones the reference was run on")
    return frames


def existing(names):
    return [n for n in names if os.path.exists(os.path.join(GOLDEN, n + ".npz"))]


def long_runs(names=None):
    runs = []
    for name in existing(LONG_CASES if names is None else names):
        z = load(name)
        for t in z["tags"]:
            runs.append((name, str(t)))
    return runs


def all_runs():
    runs = []
    for name in PIPELINE_CASES:
        z = load(name)
        for t in z["tags"]:
            runs.append((name, str(t)))
    return runs
