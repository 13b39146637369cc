"""A/B of the fill + assignment pass in one process: scratch/ab_fill.py [config] [frames] [steps]
Variants are environment switches read at every launch (SITATOR_FUSE, SITATOR_FILL_DMA, SITATOR_F3_NVU) and call
options (store_rows, defer); every variant's labels are compared with the first one's."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import _lib, synth, LandmarkAnalysis, SiteNetwork, Structure

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
host = synth.config_host(cfg); M = synth.CONFIG_MOBILE[cfg]
gen = synth.TrajectoryGenerator(host, M, seed=synth.CONFIG_SEED.get(cfg, 2), threads=16)
ref = gen.reference_positions()
frames = gen.generate(F)
sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
la = LandmarkAnalysis(verbose=False)
st = la.run(sn, frames)
e2e_labels = st.traj.reshape(-1).copy()
centers = np.asarray(la.cluster_centers_)
ctx = _lib.HipContext(host.cell)
ref_static = ref[gen.static_mask]
V = max(len(v) for v in host.vertices)
verts = np.full((len(host.vertices), V), -1, dtype=np.int64); vcd = np.full(verts.shape, np.nan)
for k, v in enumerate(host.vertices):
    verts[k, :len(v)] = v; vcd[k, :len(v)] = la._ctx.distances(host.centers[k], ref_static[np.asarray(v)])
ctx.set_basis(ref_static, verts, vcd, 1.5, 30, 1.0)
ctx.set_frames(frames, np.where(gen.static_mask)[0], np.where(gen.mobile_mask)[0])
with np.errstate(divide="ignore", invalid="ignore"):
    ctx.set_centers(centers / np.linalg.norm(centers, axis=1)[:, None], True)

VARIANTS = [
    ("two kernels, rows stored, blocking", dict(SITATOR_FUSE="0"), dict(store_rows=True, defer=False)),
    ("two kernels, no DMA", dict(SITATOR_FUSE="0", SITATOR_FILL_DMA="0"), dict(store_rows=True, defer=False)),
    ("two kernels, nv looked up", dict(SITATOR_FUSE="0", SITATOR_F3_NVU="0"), dict(store_rows=True, defer=False)),
    ("two kernels, deferred", dict(SITATOR_FUSE="0"), dict(store_rows=True, defer=True)),
    ("fused, rows stored, blocking", dict(), dict(store_rows=True, defer=False)),
    ("fused, rows not stored, blocking", dict(), dict(store_rows=False, defer=False)),
    ("fused, rows not stored, deferred", dict(), dict(store_rows=False, defer=True)),
]
sel = os.environ.get("AB_ONLY")
if sel:
    VARIANTS = [VARIANTS[int(i)] for i in sel.split(",")]
KEYS = sorted({k for _, e, _ in VARIANTS for k in e})


def run(env, kw, n):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    for _ in range(n):
        rc, nz, err = ctx.fill(False, False, True, assign=True, predict_threshold=0.8, **kw)
        assert rc == 0, (rc, err.frame, err.index, ctx.message())
    if kw.get("defer"):
        rc, nz, err = ctx.fill_result()
        assert rc == 0, (rc, err.frame, err.index, ctx.message())
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(*VARIANTS[0][1:], 60)              # clocks
base = None
res = {name: [] for name, _, _ in VARIANTS}
for rnd in range(3):
    for name, env, kw in VARIANTS:
        run(env, kw, 5)
        tot0 = ctx.timer_totals()
        ms = run(env, kw, steps)
        tot1 = ctx.timer_totals()
        lap = {k: (tot1[k][0] - tot0[k][0]) / max(1, tot1[k][1] - tot0[k][1]) for k in ("fill", "predict")}
        res[name].append((ms, lap["fill"], lap["predict"]))
        if rnd == 0:
            labels, confs, counts = ctx.assignments()
            ok_e2e = bool(np.array_equal(labels, e2e_labels))
            if base is None:
                base = (labels.copy(), confs.copy(), counts.copy())
                same = "(reference)"
            else:
                same = "labels %s confs %s counts %s" % (np.array_equal(labels, base[0]), np.array_equal(confs, base[1]), np.array_equal(counts, base[2]))
            print("%-40s equal to e2e run: %s; %s; info %s" % (name, ok_e2e, same, {k: ctx.info()[k] for k in ("fill_kernel", "survivors_per_wave", "waves_per_workgroup", "task_table_per_wave", "frames_per_workgroup")}), flush=True)
print("\n%-40s %10s %10s %10s   (ms per step: wall, fill events, assignment events; best of 3 rounds / all)" % ("variant", "wall", "fill", "assign"))
for name, _, _ in VARIANTS:
    r = res[name]
    best = min(r)
    print("%-40s %10.4f %10.4f %10.4f   %s" % (name, best[0], best[1], best[2], " ".join("%.4f" % x[0] for x in r)))
print("N = %d vectors; %.3e lvec/s at the best wall" % (F * M, F * M / (min(min(r)[0] for r in res.values()) * 1e-3)))
