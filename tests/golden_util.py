"""Loading helpers for the golden fixtures in tests/golden (data generated from the true
reference by oracle/make_fixtures.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PIPELINE_CASES = ["c1_hex_scgrid", "c1b_tri_bcctet", "c2_cut_ortho", "c5_cut_fcc_ragged", "bcc_ortho",
                  "c1_variants", "c1_static_swap", "c1_zero_lvecs", "err_static_threshold",
                  "err_static_unassigned", "err_multiple_occupancy", "err_insufficient_sites"]


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
        self.frames = z["frames"]
        self.wrapped_head = z["wrapped_head"]
        self.tags = [str(t) for t in z["tags"]]

    def kwargs(self, tag):
        return json.loads(str(self.z[tag + "/kwargs"]))

    def out(self, tag):
        pre = tag + "/"
        return {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre) and not k.endswith("/kwargs")}


def all_runs():
    runs = []
    for name in PIPELINE_CASES:
        z = load(name)
        for t in z["tags"]:
            runs.append((name, str(t)))
    return runs
