#!/usr/bin/env python3
"""Headline benchmark: landmark vectors per second of the fill + site-assignment pass.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): synthetic 64-mobile / 512-landmark orthorhombic cell
(SCgrid(8,8,8), 32.0 x 35.2 x 38.4 A, A = 576 atoms), 100 000 frames PER GPU (weak scaling:
rank r holds frames [r*F, (r+1)*F) of an N*F-frame trajectory).  One "step" = one pass of the hot
path over the resident trajectory: wrap + static-lattice check + landmark vector of every
(frame, mobile ion) + cosine assignment to the fitted site centres -> int64 label + float64
confidence per (frame, ion).  Frames are resident in HBM before the timed region; the site
centres come from the product's own end-to-end `run()` on the same trajectory (outside the timed
region; its wall time is reported as `end_to_end_run`).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (k_fill_rows) with the
algorithmic bytes of SURVEY.md section 8(d): B = 24*A/M + 16 bytes per landmark vector.
`cpu_baseline` = the oracle's C port of the same pass, single thread, on a bounded cut.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=100000, help="frames per GPU")
    ap.add_argument("--config", default="C2")
    ap.add_argument("--cpu-frames", type=int, default=2000, help="frames of the CPU-baseline cut (0 = skip)")
    ap.add_argument("--compare-v1", action="store_true",
                    help="also time the first-generation kernels (fill + predict), interleaved in this process")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    backend = os.environ.get("SITATOR_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 path on a 1-GPU box
    if world > 1:
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        if backend == "nccl":
            if local >= ndev:
                raise RuntimeError("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local, ndev))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            local = local % max(ndev, 1)
            torch.cuda.set_device(local)
            dist.init_process_group(backend)
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth, _lib

    host = synth.config_host(args.config)
    M = synth.CONFIG_MOBILE[args.config]
    S, D = len(host.static_pos), len(host.centers)
    A = S + M
    F = args.frames
    gen = synth.TrajectoryGenerator(host, M, seed=synth.CONFIG_SEED[args.config] + 1000 * rank,
                                    threads=max(1, min(16, (os.cpu_count() or 8) // max(1, min(world, 8)))))
    ref = gen.reference_positions()
    t0 = time.time()
    frames = gen.generate(F)
    t_gen = time.time() - t0
    sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask)
    sn.centers = host.centers
    sn.vertices = host.vertices

    # --- end-to-end: the product's own LandmarkAnalysis.run() on this rank's whole trajectory (upload,
    #     tables, fill, exact fit_centers, two predict passes, site centres, occupancy check).  It is
    #     reported beside the headline and supplies the fitted site centres for the timed pass. ---
    t0 = time.time()
    la = LandmarkAnalysis(verbose=False, device=local)
    st_full = la.run(sn, frames)
    t_e2e = time.time() - t0
    e2e = {"frames": F, "seconds": round(t_e2e, 4), "lvec_per_s": round(F * M / t_e2e, 1),
           "wall_s": {k: round(v, 4) for k, v in la.wall_timings.items()},
           "sites": int(st_full.site_network.n_sites), "unassigned_frac": float(st_full.percent_unassigned),
           "fit": {k: v for k, v in la._ctx.info().items() if k.startswith("fit_")}}
    e2e_labels = st_full.traj.reshape(-1)

    # --- resident context for the timed pass ---
    ctx = _lib.HipContext(host.cell, device=local)
    pb_ctx = la._ctx
    verts = np.full((D, max(len(v) for v in host.vertices)), -1, dtype=np.int64)
    vcd = np.full(verts.shape, np.nan)
    ref_static = ref[gen.static_mask]
    for k, v in enumerate(host.vertices):
        verts[k, :len(v)] = v
        vcd[k, :len(v)] = pb_ctx.distances(host.centers[k], ref_static[np.asarray(v)])
    ctx.set_basis(ref_static, verts, vcd, 1.5, 30, 1.0)
    ctx.set_frames(frames, np.where(gen.static_mask)[0], np.where(gen.mobile_mask)[0], frame0=rank * F)
    h2d_ms = ctx.timers()["h2d"]
    # centres fitted on the cut: representative vectors of the sites found
    fit_ctx_centers = np.asarray(la.cluster_centers_)
    with np.errstate(divide="ignore", invalid="ignore"):
        normed = fit_ctx_centers / np.linalg.norm(fit_ctx_centers, axis=1)[:, None]
    ctx.set_centers(normed, True)

    def step():
        rc, nz, err = ctx.fill(False, False, True, assign=True, predict_threshold=0.8)
        if rc != 0:
            raise RuntimeError("fill failed rc=%d frame=%d index=%d: %s" % (rc, err.frame, err.index, ctx.message()))

    def sync_all():
        ctx.synchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()
    fill_ms, pred_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = ctx.timers()
        fill_ms.append(tm["fill"])
        pred_ms.append(tm["predict"])
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ab = None
    if args.compare_v1:
        os.environ["SITATOR_FILL_KERNEL"] = "1"
        ctx1 = _lib.HipContext(host.cell, device=local)
        ctx1.set_basis(ref_static, verts, vcd, 1.5, 30, 1.0)
        os.environ.pop("SITATOR_FILL_KERNEL")
        ctx1.set_frames(frames, np.where(gen.static_mask)[0], np.where(gen.mobile_mask)[0], frame0=rank * F)
        ctx1.set_centers(normed, True)
        t_v1, t_v2 = [], []
        for _ in range(max(3, args.steps)):
            ctx1.fill(False, False, True, assign=True, predict_threshold=0.8)
            tm = ctx1.timers()
            t_v1.append(tm["fill"] + tm["predict"])
            step()
            t_v2.append(ctx.timers()["fill"] + ctx.timers()["predict"])
        l1, c1, n1 = ctx1.assignments()
        l2, c2, n2 = ctx.assignments()
        ab = {"v1_fill_plus_predict_ms": {"median": float(np.median(t_v1)), "min": float(np.min(t_v1))},
              "v2_fill_plus_predict_ms": {"median": float(np.median(t_v2)), "min": float(np.min(t_v2))},
              "labels_identical": bool(np.array_equal(l1, l2)), "confs_identical": bool(np.array_equal(c1, c2))}
        ctx1.close()

    labels, confs, counts = ctx.assignments()
    checks = {"unassigned_frac": float(np.mean(labels < 0)), "sites": int(len(counts)),
              "labels_equal_end_to_end_run": bool(np.array_equal(labels, e2e_labels)),
              "label_checksum": int(np.sum(labels[labels >= 0] * 7 + 1) % 1000003)}

    if rank == 0:
        n_lvec = world * F * M * args.steps
        value = n_lvec / elapsed
        bytes_per_lvec = 24.0 * A / M + 16.0
        fill_avg_ms = float(np.mean(fill_ms))
        achieved = (F * M * bytes_per_lvec) / (fill_avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("k_fill2_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "landmark-vectors/sec (frames x mobile atoms), fill + site assignment",
            "value": value, "unit": "lvec/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: SCgrid(8,8,8) orthorhombic 32.0x35.2x38.4 A, S=D=512 (V=8), M=64, "
                                   "A=576, %d frames per GPU" % F if args.config == "C2" else args.config,
                       "frames_per_gpu": F, "n_mobile": M, "n_static": S, "landmark_dim": D,
                       "parallelism": "frame-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_fill2", "kernel_ms": fill_avg_ms,
                         "algorithmic_bytes_per_lvec": bytes_per_lvec},
            "stages_ms": {"fill": fill_avg_ms, "predict": float(np.mean(pred_ms)), "h2d_frames": h2d_ms,
                          "generate_s": round(t_gen, 2)},
            "end_to_end_run": e2e,
            "checks": checks,
        }
        if ab is not None:
            out["ab_kernels"] = ab
        if args.cpu_frames > 0 and world == 1:          # a reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(host, gen, frames[:min(F, args.cpu_frames)], ref, fit_ctx_centers, M)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(host, gen, frames, ref, centers, M):
    """The oracle's C port of the same pass (wrap, static check, dense landmark vectors, predict),
    one thread, on a leading cut of the same trajectory."""
    from oracle import oracle as orc
    orc.lib()
    t0 = time.perf_counter()
    wrapped = orc.wrap_points(host.cell, frames)
    ref_static = ref[gen.static_mask]
    verts, vcd = orc.site_vertex_distances(host.cell, host.centers, host.vertices, ref_static)
    lv, _ = orc.fill(host.cell, wrapped, np.where(gen.static_mask)[0], np.where(gen.mobile_mask)[0],
                     ref_static, verts, vcd)
    orc.predict(lv, centers, 0.8, True)
    dt = time.perf_counter() - t0
    return {"value": len(frames) * M / dt, "unit": "lvec/s", "cores": 1, "kind": "port",
            "sample": "leading %d frames of the same workload (%d landmark vectors), %.1f s"
                      % (len(frames), len(frames) * M, dt)}


if __name__ == "__main__":
    main()
