"""Where the set-up time of one analysis goes (context, basis / loose table, tight table)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import _lib, synth
host = synth.config_host(sys.argv[1] if len(sys.argv) > 1 else "C2")
M = 64
frames, sm, mm, ref = synth.make_trajectory(host, M, 2000, seed=2)
for rep in range(2):
    t0 = time.time(); ctx = _lib.HipContext(host.cell); ctx.synchronize(); t1 = time.time()
    V = max(len(v) for v in host.vertices)
    verts = np.full((len(host.vertices), V), -1, dtype=np.int64)
    for k, v in enumerate(host.vertices): verts[k, :len(v)] = v
    vcd = ctx.site_vertex_distances(host.centers, ref[sm], verts); t2 = time.time()
    ctx.set_basis(ref[sm], verts, vcd, 1.5, 30, 1.0); ctx.synchronize(); t3 = time.time()
    ctx.set_frames(frames, np.where(sm)[0], np.where(mm)[0]); ctx.synchronize(); t4 = time.time()
    ctx.fill(); t5 = time.time()
    ctx.fill(); t6 = time.time()
    print("rep %d: create %.1f ms, vertex dists %.1f, set_basis %.1f, set_frames %.1f, first fill %.1f, second fill %.1f"
          % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t5 - t4), 1e3 * (t6 - t5)))
    ctx.close()
