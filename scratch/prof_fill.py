"""Small driver for profiling the fill kernel: C2 host, F frames, fused assign (no stored rows)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import _lib, synth, LandmarkAnalysis, SiteNetwork, Structure
F = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = sys.argv[3] if len(sys.argv) > 3 else "C2"
host = synth.config_host(cfg); M = synth.CONFIG_MOBILE[cfg]
gen = synth.TrajectoryGenerator(host, M, seed=2)
ref = gen.reference_positions()
frames = gen.generate(F)
sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
la = LandmarkAnalysis(verbose=False)
la.run(sn, np.ascontiguousarray(frames[::max(1, F // 2000)][:2000]))
centers = np.asarray(la.cluster_centers_)
ctx = _lib.HipContext(host.cell)
ref_static = ref[gen.static_mask]
V = max(len(v) for v in host.vertices)
verts = np.full((len(host.vertices), V), -1, dtype=np.int64); vcd = np.full(verts.shape, np.nan)
for k, v in enumerate(host.vertices):
    verts[k, :len(v)] = v; vcd[k, :len(v)] = la._ctx.distances(host.centers[k], ref_static[np.asarray(v)])
ctx.set_basis(ref_static, verts, vcd, 1.5, 30, 1.0)
ctx.set_frames(frames, np.where(gen.static_mask)[0], np.where(gen.mobile_mask)[0])
ctx.set_centers(centers / np.linalg.norm(centers, axis=1)[:, None], True)
for i in range(steps):
    rc, nz, err = ctx.fill(assign=True, predict_threshold=0.8, store_rows=False)
    print("step", i, "rc", rc, "fill ms", ctx.timers()["fill"], "predict ms", ctx.timers()["predict"])
print(ctx.info(), "K", len(centers))
rc, nz, err = ctx.fill()
nnz, idx, val = ctx.rows_sparse()
print("nnz mean %.3f max %d" % (nnz.mean(), nnz.max()), "store fill ms", ctx.timers()["fill"])
