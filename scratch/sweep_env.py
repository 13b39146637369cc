"""Times the fill + assignment pass under environment settings, interleaved in one process:
    scratch/sweep_env.py <config> <frames> "K=V,K=V" "K=V" ...      ("" = the defaults)
Prints ms per pass (wall, fill events, assignment events), best of three rounds, and whether the labels equal the first
setting's."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sitator_amd import _lib, synth, LandmarkAnalysis, SiteNetwork, Structure

cfg, F = sys.argv[1], int(sys.argv[2])
specs = sys.argv[3:] or [""]
host = synth.config_host(cfg); M = synth.CONFIG_MOBILE[cfg]
gen = synth.TrajectoryGenerator(host, M, seed=synth.CONFIG_SEED.get(cfg, 2), threads=16)
ref = gen.reference_positions()
frames = gen.generate(F)
sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask); sn.centers = host.centers; sn.vertices = host.vertices
kw = {"clustering_algorithm": "mcl", "max_mobile_per_site": 2} if cfg == "C5" and os.environ.get("SWEEP_MCL") else {}
la = LandmarkAnalysis(verbose=False, **kw)
la.run(sn, np.ascontiguousarray(frames[:min(F, int(os.environ.get("SWEEP_FIT_FRAMES", "20000")))]))
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
envs = [dict(kv.split("=", 1) for kv in s.split(",") if kv) for s in specs]
KEYS = sorted({k for e in envs for k in e})


def run(env, n):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    for _ in range(n):
        rc, nz, err = ctx.fill(False, False, True, assign=True, predict_threshold=0.8, store_rows=False, defer=True)
        assert rc == 0, (rc, err.frame, err.index, ctx.message())
    rc, nz, err = ctx.fill_result()
    assert rc == 0, (rc, err.frame, err.index, ctx.message())
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


steps = int(os.environ.get("SWEEP_STEPS", "20"))
run(envs[0], max(10, int(60e6 / (F * M)) * 6))             # clocks
res = [[] for _ in envs]
base = None
for rnd in range(3):
    for i, env in enumerate(envs):
        run(env, 3)
        tot0 = ctx.timer_totals()
        ms = run(env, steps)
        tot1 = ctx.timer_totals()
        lap = {k: (tot1[k][0] - tot0[k][0]) / max(1, tot1[k][1] - tot0[k][1]) for k in ("fill", "predict")}
        res[i].append((ms, lap["fill"], lap["predict"]))
        if rnd == 0:
            labels, confs, counts = ctx.assignments()
            if base is None:
                base = labels.copy()
            inf = ctx.info()
            print("%-60s labels equal: %s; shape nw %d fpb %d rcap %d tt %d fused %s" % (specs[i] or "(defaults)", np.array_equal(labels, base),
                  inf["waves_per_workgroup"], inf["frames_per_workgroup"], inf["survivors_per_wave"], inf["task_table_per_wave"], inf["assignment_fused"]), flush=True)
print("\n%-60s %9s %9s %9s" % ("setting", "wall", "fill", "assign"))
for i, s in enumerate(specs):
    b = min(res[i])
    print("%-60s %9.4f %9.4f %9.4f   %s" % (s or "(defaults)", b[0], b[1], b[2], " ".join("%.4f" % x[1] for x in res[i])))
