#!/usr/bin/env python3
"""Headline benchmark: landmark vectors per second of the fill + site-assignment pass.

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5|C2h|C2t] [--frames F] [--algo dotprod|mcl]

Workload: BASELINE.json configs[1], "C2": synthetic 64-mobile / 512-landmark orthorhombic cell (SCgrid(8,8,8), 32.0 x
35.2 x 38.4 A, A = 576 atoms), 100 000 frames PER GPU at every N (weak scaling: rank r holds frames [r*F, (r+1)*F) of
an N*F-frame trajectory, so that a 1 -> 8 series compares like with like).  `--config C4` times BASELINE configs[3]
instead (256-mobile / 2 048-landmark cell, 1 000 000 frames over 8 GPUs = 125 000 per GPU; the N = 1 line carries that
share as `scale_ref`); `--config` picks any other (C3: 448 mobile, 250 000 frames;
C5: the ragged FCC host, 62 500 frames per GPU, end-to-end run with the mcl plugin and jump detection; C2h / C2t: the
C2 shape on a hexagonal / triclinic cell).  One "step" = one pass of the hot path over the resident trajectory: wrap +
static-lattice check + landmark vector of every (frame, mobile ion) + cosine assignment to the fitted site centres ->
int64 label + float64 confidence per (frame, ion).  Frames are resident in HBM before the timed region; the site
centres come from the product's own end-to-end `run()` on the same trajectory (outside the timed region; it is run
three times - `cold_seconds` is the first run of the process, `end_to_end_run.seconds` the faster of the two warm runs
behind it incl. jump detection, both listed in `warm_runs_seconds`).
At N = 1 the default line also carries the other BASELINE configurations a GPU can hold (`configs`: C3, one GPU's share
of C4 and of C5), each timed by a child process running `bench.py --config <it>` - a fresh process, because tens of GB of
device buffers of another size class in the same process cost their hipFree / hipMalloc in whoever runs next.

N > 1: one process per GPU.  Under `python -m torch.distributed.run ... bench.py --gpus N` the ranks come from
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; a bare `python bench.py --gpus N` starts the N ranks itself (fresh child
processes, before anything touches a GPU).  The ranks talk through the library's own RCCL entry points
(`sit_comm_*`, sitator_amd/sharding.py `RcclComm`) - the end-to-end run exercises every exchange step of the path
(first-offender keys, counts, the ordered fit relay, site-centre anchors and sums, occupancy); the timed pass has no
collective in it (frames shard embarrassingly) and is bracketed by a barrier on both sides.  The set-up (RCCL unique
id, the ranks' agreement that every one of them got its communicator) is a TCP channel of the package's own
(`sharding.Control`): no torch in this file or in the package.  A rank without a communicator ends the run non-zero.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (k_fill3 unless the tables force the general
fallback) with the algorithmic bytes of SURVEY.md section 8(d): B = 24*A/M + 16 bytes per landmark vector;
`roofline.frac_step` is the same for the whole step (ms_per_step: fill + assignment + launch gaps); `roofline.limiter`
/ `roofline.valu` say what the counters say keeps the kernel below that roofline.  `cpu_baseline` = the oracle's C port of the
same pass on a bounded cut, one thread (the reference's execution model) and all host cores.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("SITATOR_PROGRESSBAR", "false")      # the reference's switch: no per-stage lines on stderr

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU (default: the configuration's own, below)")
    ap.add_argument("--config", default=None, help="C2 (BASELINE.json configs[1]) at every N; C4 = configs[3]'s per-GPU share")
    ap.add_argument("--algo", default=None, help="clustering plugin of the end-to-end run: dotprod, or mcl (the default of C5)")
    ap.add_argument("--cpu-frames", type=int, default=1000, help="frames per core of the CPU-baseline cut (0 = skip)")
    ap.add_argument("--no-scale-ref", action="store_true", help="default workload: skip the extra passes over the other BASELINE configurations (C3, C4's per-GPU share, C5)")
    ap.add_argument("--repeats", type=int, default=4, help="further timed regions of K steps behind the reported one (min / median of ms_per_step)")
    return ap.parse_args()


# frames per GPU: BASELINE.json's trajectory lengths, the 8-GPU configurations cut into their per-GPU share
FRAMES_PER_GPU = {"C1": 2000, "C1b": 1000, "C2": 100000, "C3": 250000, "C4": 125000, "C5": 62500, "C2h": 100000, "C2t": 100000}


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (this process has
    not touched a GPU and never will), pass rank 0's JSON line through."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].stdout
    rc = 0
    try:
        for line in out0:
            sys.stdout.write(line.decode())
            sys.stdout.flush()
        for p in procs:
            rc = max(rc, p.wait())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries the one JSON line and nothing else: whatever the libraries print (gloo's connection notes, RCCL's
    # version banner) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    backend = os.environ.get("SITATOR_BENCH_BACKEND", "rccl")     # "tcp": rehearsal of the N>1 path on a 1-GPU box
    import numpy as np
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth, _lib, sharding

    comm = None
    exchange = "none (single process)"
    if world > 1:
        # One process per GPU; the exchange steps run on RCCL through the library's own entry points.  The set-up
        # (unique id, the ranks' agreement that every one of them has its communicator) is a TCP channel of the
        # package's own - no torch.  A rank that cannot get its communicator ends the run, on every rank, non-zero.
        if backend == "tcp":                   # rehearsal on a development box: the ranks share what GPUs there are
            comm = sharding.TcpComm.from_env()
            local = local % max(_lib.device_count(), 1)
            exchange = "tcp (rehearsal: ranks share a GPU, no RCCL communicator)"
        else:
            try:
                comm = sharding.RcclComm.from_env(device=local)
            except RuntimeError as e:
                print("[rank %d/%d] %s" % (rank, world, e), file=sys.stderr, flush=True)
                os._exit(3)                    # a fresh process; _exit because a thread may still sit inside RCCL
            exchange = "rccl"
            print("[rank %d/%d] RCCL communicator up on GPU %d" % (rank, world, local), file=sys.stderr, flush=True)

    if args.config is None:
        # the same workload at every N (weak scaling: every GPU takes configs[1]'s 100 000 frames), so that a 1 -> 8
        # series compares like with like; `--config C4` times BASELINE configs[3]'s per-GPU share instead
        args.config = "C2"
    if args.frames is None:
        args.frames = FRAMES_PER_GPU.get(args.config, 100000)
    if args.algo is None:
        args.algo = "mcl" if args.config == "C5" else "dotprod"
    # C5 with the mcl plugin's defaults puts two ions on one merged site of the synthetic FCC host (the reference
    # raises MultipleOccupancyError there too: a golden); max_mobile_per_site=2 lets the full pipeline through
    la_kw = {"clustering_algorithm": args.algo}
    if args.algo == "mcl":
        la_kw["max_mobile_per_site"] = 2
    host = synth.config_host(args.config)
    M = synth.CONFIG_MOBILE[args.config]
    S, D = len(host.static_pos), len(host.centers)
    A = S + M
    F = args.frames
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    gen = synth.TrajectoryGenerator(host, M, seed=synth.CONFIG_SEED[args.config] + 1000 * rank,
                                    threads=max(1, min(16, ncpu // max(1, min(world, 8)))))
    ref = gen.reference_positions()
    t0 = time.time()
    frames = gen.generate(F)
    t_gen = time.time() - t0
    sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask)
    sn.centers = host.centers
    sn.vertices = host.vertices

    # --- end-to-end: the product's own LandmarkAnalysis.run() on the whole (N*F-frame) trajectory, this rank's
    #     shard resident on its GPU (upload, tables, fill, exact fit_centers, two predict passes, site centres,
    #     occupancy check; with N > 1 every exchange step of the path runs on RCCL).  It is reported beside the
    #     headline and supplies the fitted site centres for the timed pass. ---
    # The first run of a process also pays for the HIP runtime, the code objects and the first allocations
    # (`cold_seconds`); the second is what a long-lived analysis process sees per trajectory.
    t0 = time.time()
    LandmarkAnalysis(verbose=False, device=local, comm=comm, **la_kw).run(sn, frames)
    t_cold = time.time() - t0
    if comm is not None:
        comm.barrier()
    # two warm runs, the faster one reported and both listed: the SECOND run of a process comes out 5-8 ms slower than
    # the third and later ones on some boxes (its uploads are slower; scratch/e2e_walls.py shows the series)
    warm = []
    la = st_full = None
    for _ in range(2):
        del la, st_full
        if comm is not None:
            comm.barrier()
        t0 = time.time()
        la = LandmarkAnalysis(verbose=False, device=local, comm=comm, **la_kw)
        st_full = la.run(sn, frames)
        t_run_i = time.time() - t0
        n_jumps = sum(1 for _ in st_full.jumps())      # jump detection (SiteTrajectory.py:307-373): configs[4]'s last step
        warm.append((time.time() - t0, t_run_i, dict(la.wall_timings)))
    if comm is not None:                                # every rank reports the same run: the one rank 0 found faster
        pick = int(comm.bcast(np.array([0 if warm[0][0] <= warm[1][0] else 1], dtype=np.int64))[0])
    else:
        pick = 0 if warm[0][0] <= warm[1][0] else 1
    t_e2e, t_run, walls = warm[pick]
    e2e = {"frames": F * world, "seconds": round(t_e2e, 4), "cold_seconds": round(t_cold, 4), "warm_runs_seconds": [round(w[0], 4) for w in warm],
           "algo": args.algo, "run_seconds": round(t_run, 4), "jump_detection_seconds": round(t_e2e - t_run, 4),
           "jumps": n_jumps, "lvec_per_s": round(world * F * M / t_e2e, 1),
           "wall_s": {k: round(v, 4) for k, v in walls.items()},
           "sites": int(st_full.site_network.n_sites), "unassigned_frac": float(st_full.percent_unassigned),
           "fit": {k: v for k, v in la._ctx.info().items() if k.startswith("fit_")},
           "exchange": exchange}
    e2e_labels = st_full.traj.reshape(-1)
    if world > 1 and args.algo == "dotprod":
        # how the ranks spent the fit of the exact (relayed) run, and the same run with fit_mode="shard-merge" (every rank
        # fits its shard, all-gather of the clusters' sufficient statistics, identical merge everywhere): seconds, and
        # the positions whose site differs from the exact run's once the numberings are matched
        ft = la.fit_timings or {}
        e2e["fit_per_rank"] = {k: [round(float(v), 4) for v in comm.allgather(np.array([ft.get(k, 0.0)]))[:, 0]]
                               for k in ("fit_s", "exchange_s", "merge_s")}
        comm.barrier()
        t0 = time.time()
        la_m = LandmarkAnalysis(verbose=False, device=local, comm=comm, fit_mode="shard-merge", **la_kw)
        st_m = la_m.run(sn, frames)
        t_m = time.time() - t0
        fm = la_m.fit_timings or {}
        k_exact, k_merge = int(st_full.site_network.n_sites), int(st_m.site_network.n_sites)
        e2e["shard_merge"] = {
            "run_seconds": round(t_m, 4), "sites": k_merge,
            "lvec_per_s": round(world * F * M / t_m, 1),
            "positions_differing_from_exact": sharding.partition_mismatch_sharded(comm, st_full.traj, st_m.traj, k_exact, k_merge),
            "positions": int(world * F * M),
            "clusters_per_rank": fm.get("clusters_per_rank"),
            "fit_per_rank": {k: [round(float(v), 4) for v in comm.allgather(np.array([fm.get(k, 0.0)]))[:, 0]]
                             for k in ("fit_s", "exchange_s", "merge_s")}}
        del la_m, st_m

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
    # centres of the sites found by the run above
    fit_ctx_centers = np.asarray(la.cluster_centers_)
    with np.errstate(divide="ignore", invalid="ignore"):
        normed = fit_ctx_centers / np.linalg.norm(fit_ctx_centers, axis=1)[:, None]
    ctx.set_centers(normed, True)

    def check(rc, err):
        if rc != 0:
            raise RuntimeError("fill failed rc=%d frame=%d index=%d: %s" % (rc, err.frame, err.index, ctx.message()))

    def step():
        # one pass, enqueued (sit_fill with defer = 1: no host synchronisation per pass; the error word of every pass is
        # still read back and decoded - by a later call, or by fill_result() below)
        rc, nz, err = ctx.fill(False, False, True, assign=True, predict_threshold=0.8, store_rows=False, defer=True)
        check(rc, err)

    def sync_all():
        rc, nz, err = ctx.fill_result()        # waits for the passes in flight; the first failure among them, if any
        check(rc, err)
        ctx.synchronize()
        if comm is not None:
            comm.barrier()

    # The GPU raises its clocks over the first tens of milliseconds of sustained load (k_fill3 at C2: 0.97 -> 0.865 ms
    # over ~30 back-to-back launches): untimed passes until the pass time has settled, so that the W warm-up steps and
    # the K timed steps see the state a long-running analysis sees.  Reported as `clock_ramp_steps`.
    ramp, ramp_ms, t_r0 = 0, [], time.perf_counter()
    while ramp < 100 and time.perf_counter() - t_r0 < 0.4:
        step()
        ramp += 1
        ramp_ms.append(ctx.timers()["fill"])
        if ramp >= 12 and max(ramp_ms[-6:]) <= 1.012 * min(ramp_ms[-6:]):
            break
    for _ in range(args.warmup):
        step()
    sync_all()
    # the library sums the HIP-event laps of its stages: read before and after, not once per pass
    tot0 = ctx.timer_totals()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    tot1 = ctx.timer_totals()
    laps = {k: ((tot1[k][0] - tot0[k][0]) / max(1, tot1[k][1] - tot0[k][1]), tot1[k][1] - tot0[k][1]) for k in ("fill", "predict")}
    assert laps["fill"][1] == args.steps and laps["predict"][1] == args.steps, laps
    elapsed_own = elapsed

    def max_over_ranks(x):
        if comm is None:
            return float(x)
        if hasattr(comm, "allreduce_max"):
            return float(comm.allreduce_max(np.array([x]))[0])
        return float(np.max(comm.allgather(np.array([x]))))

    elapsed = max_over_ranks(elapsed)
    # The reported region is 20 x 0.8 ms: one hiccup is 5 %.  Further regions of K steps each (every one bracketed like
    # the first; the maximum over the ranks each), for a minimum and a median beside the value.
    regions = [1e3 * elapsed / args.steps]
    for _ in range(max(0, args.repeats)):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync_all()
        regions.append(1e3 * max_over_ranks(time.perf_counter() - t0) / args.steps)
    info = ctx.info()

    # N > 1: the line proves its own rank count - what every rank's RCCL communicator says about itself (read back
    # from the library, not what the launcher asked for) and every rank's own rate
    rccl_info = per_rank_value = None
    if comm is not None:
        mine = comm.info() if hasattr(comm, "info") else {"ranks": comm.size, "rank": comm.rank, "device": local, "rccl_version": 0}
        got = comm.allgather(np.array([mine["ranks"], mine["rank"], mine["device"], mine.get("rccl_version", 0)], dtype=np.int64))
        rccl_info = {"backend": exchange, "ranks": [int(x) for x in got[:, 0]], "rank": [int(x) for x in got[:, 1]],
                     "device": [int(x) for x in got[:, 2]], "rccl_version": int(got[0, 3])}
        per_rank_value = [float(x) for x in comm.allgather(np.array([F * M * args.steps / elapsed_own]))[:, 0]]
    labels, confs, counts = ctx.assignments()
    checks = {"unassigned_frac": float(np.mean(labels < 0)), "sites": int(len(counts)),
              "labels_equal_end_to_end_run": bool(np.array_equal(labels, e2e_labels)),
              "label_checksum": int(np.sum(labels[labels >= 0] * 7 + 1) % 1000003)}

    # N > 1, default workload: every rank also times one GPU's share of configs[3] (C4: 125 000 of the 1e6 frames) - the
    # configuration BASELINE.json names for the 8-GPU run - bracketed by barriers; the record carries the sum over the ranks
    c4 = None
    if world > 1 and args.config == "C2" and F == FRAMES_PER_GPU["C2"] and not args.no_scale_ref:
        del frames, la, st_full
        comm.barrier()
        c4 = other_config(args, local, "C4")
        vals = comm.allgather(np.array([c4["value"], c4["ms_per_step"]]))
        c4["value_per_rank"] = [float(x) for x in vals[:, 0]]
        c4["value"] = float(np.sum(vals[:, 0]) * np.min(vals[:, 1]) / np.max(vals[:, 1]))     # all ranks' vectors over the slowest rank's time
        c4["ms_per_step"] = float(np.max(vals[:, 1]))
        c4["workload"] = c4["workload"].replace("(one GPU's share of the 8-GPU run)", "per GPU, %d GPUs" % world)
    if rank == 0:
        n_lvec = world * F * M * args.steps
        value = n_lvec / elapsed
        bytes_per_lvec = 24.0 * A / M + 16.0
        fill_avg_ms = float(laps["fill"][0])
        step_ms = 1e3 * elapsed / args.steps
        achieved = (F * M * bytes_per_lvec) / (fill_avg_ms * 1e-3) / 1e9
        kernel = "k_fill%d" % info["fill_kernel"] if info["fill_kernel"] > 1 else "k_fill_rows"
        valu = measured_valu(kernel, args.config)
        out = {
            "metric": "landmark-vectors/sec (frames x mobile atoms), fill + site assignment",
            "value": value, "unit": "lvec/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "ms_per_step_regions": {"n": len(regions), "min": min(regions), "median": float(np.median(regions)), "max": max(regions),
                                    "note": "regions of K steps each; the first is the reported one"},
            # how the pass was run: enqueued with defer = 1 (no host synchronisation per pass, every pass's
            # error word collected before the clock stops), labels + confidences written, landmark rows not kept
            "pass_mode": {"workload_id": args.config, "deferred": True, "store_rows": False, "assign": True},
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s, %d frames per GPU" % (synth.CONFIG_TEXT.get(args.config, args.config), F),
                       "frames_per_gpu": F, "n_mobile": M, "n_static": S, "landmark_dim": D,
                       "parallelism": "frame-sharded x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(kernel),
                         "kernel": kernel, "kernel_ms": fill_avg_ms,
                         "algorithmic_bytes_per_lvec": bytes_per_lvec,
                         # the whole step (fill + assignment + launch gaps = ms_per_step) against the same bytes
                         "frac_step": (F * M * bytes_per_lvec) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         # `bound` names the roofline the kernel is priced against (its bytes are 1.09x the algorithmic
                         # ones: no wasted traffic); what the counters say holds it below that roofline
                         # (profiles/pmc_valu.json, taken with this build; null if the library has changed since):
                         "limiter": limiter_of(valu), "valu": valu,
                         # SURVEY 8(d)'s other roofline: FP64 vector lane-instructions per second of the kernel against
                         # the FP64 vector peak (256 CUs x 4 SIMDs x 16 FP64 lanes per clock x 2.4 GHz = 39.3e12 lane-
                         # instructions/s = the guide's 78.6 TFLOP/s with an fma as two); from the same counter set
                         "valu_frac": valu_fraction(valu, F * M, fill_avg_ms)},
            "clock_ramp_steps": ramp,
            "fill_shape": {k: info.get(k) for k in ("waves_per_workgroup", "frames_per_workgroup", "survivors_per_wave", "task_table_per_wave", "assignment_fused")},
            "stages_ms": {"fill": fill_avg_ms, "predict": float(laps["predict"][0]), "h2d_frames": h2d_ms,
                          "generate_s": round(t_gen, 2)},
            "end_to_end_run": e2e,
            "checks": checks,
        }
        if comm is not None:
            out["rccl"] = rccl_info
            out["value_per_rank"] = per_rank_value
        if args.cpu_frames > 0 and world == 1:          # a reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(host, gen, frames, ref, fit_ctx_centers, M, args.cpu_frames, ncpu)
        if world == 1 and args.config == "C2" and F == FRAMES_PER_GPU["C2"] and not args.no_scale_ref:
            # The other BASELINE configurations in the same line: C3 (configs[2], the larger one-GPU configuration, at its
            # stated size), one GPU's share of C4 (configs[3]; also `scale_ref`: what `--gpus N --config C4` times on N GPUs)
            # and of C5 (configs[4]: the end-to-end run is Markov clustering + jump detection).
            # Each in a fresh process: tens of GB of device buffers of ANOTHER size class in the same process cost their
            # hipFree / hipMalloc in whoever runs next (C4's warm run() 0.30 s behind C3 in one process, 0.20 s alone), and a
            # child's record is by construction what `bench.py --config C3` prints on its own (profiles/*_bench_C3.json).
            del frames, ctx, la, st_full
            _lib.release_cached_memory()
            out["configs"] = [other_config_child(args, c) for c in ("C3", "C4", "C5")]
            if "error" not in out["configs"][1]:
                out["scale_ref"] = {k: v for k, v in out["configs"][1].items() if k in ("workload", "value", "unit", "ms_per_step", "steps")}
                out["scale_ref"]["end_to_end_run_seconds_cold"] = out["configs"][1]["end_to_end_run"]["cold_seconds"]
        if c4 is not None:
            out["configs"] = [c4]
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.barrier()
        if hasattr(comm, "close"):
            comm.close()


def other_config_child(args, cfg):
    """`bench.py --config cfg` in a child process (started, not exec'ed: this process keeps its GPU), its line cut down to
    the sub-record of `configs`."""
    cmd = [sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--cpu-frames", "0", "--repeats", "1", "--no-scale-ref"]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, env=dict(os.environ), timeout=600)
        line = p.stdout.decode().strip().splitlines()[-1] if p.stdout.strip() else ""
        if p.returncode != 0 or not line.startswith("{"):
            raise RuntimeError("exit %d" % p.returncode)
        d = json.loads(line)
    except Exception as e:      # noqa: BLE001 - the headline line must not depend on a side record
        print("bench.py --config %s failed: %s" % (cfg, e), file=sys.stderr, flush=True)
        return {"workload": cfg, "error": "%s: %s" % (type(e).__name__, e), "process": "own (`bench.py --config %s`)" % cfg}
    e2e = d["end_to_end_run"]
    return {"workload": d["config"]["workload"] + (" (one GPU's share of the 8-GPU run)" if cfg in ("C4", "C5") else ""),
            "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
            "ms_per_step_regions": d.get("ms_per_step_regions"), "stages_ms": {k: d["stages_ms"][k] for k in ("fill", "predict")},
            "roofline": {k: d["roofline"].get(k) for k in ("bound", "kernel", "kernel_ms", "algorithmic_bytes_per_lvec", "peak", "unit", "frac", "frac_step")},
            "fill_shape": d.get("fill_shape"),
            "end_to_end_run": {k: e2e.get(k) for k in ("algo", "seconds", "cold_seconds", "run_seconds", "jumps", "sites", "lvec_per_s", "wall_s")},
            "process": "own (`bench.py --config %s`)" % cfg}


def other_config(args, device, cfg):
    """Another BASELINE configuration on this GPU, in brief: the end-to-end `run()` (cold, then warm) and the timed pass
    with its roofline fractions."""
    import numpy as np
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth
    F = FRAMES_PER_GPU[cfg]
    host = synth.config_host(cfg)
    M = synth.CONFIG_MOBILE[cfg]
    A = len(host.static_pos) + M
    gen = synth.TrajectoryGenerator(host, M, seed=synth.CONFIG_SEED[cfg], threads=16)
    ref = gen.reference_positions()
    frames = gen.generate(F)
    sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask)
    sn.centers = host.centers
    sn.vertices = host.vertices
    kw = {"clustering_algorithm": "mcl", "max_mobile_per_site": 2} if cfg == "C5" else {}
    t0 = time.time()
    LandmarkAnalysis(verbose=False, device=device, **kw).run(sn, frames)
    t_cold = time.time() - t0
    warm = []
    for _ in range(2):                                       # two warm runs, the faster one reported (both listed)
        t0 = time.time()
        la = LandmarkAnalysis(verbose=False, device=device, **kw)
        st = la.run(sn, frames)
        t_run_i = time.time() - t0
        n_jumps = sum(1 for _ in st.jumps())
        warm.append((time.time() - t0, t_run_i))
    t_e2e, t_run = min(warm)
    ctx = la._ctx
    centers = np.asarray(la.cluster_centers_)
    with np.errstate(divide="ignore", invalid="ignore"):
        ctx.set_centers(centers / np.linalg.norm(centers, axis=1)[:, None], True)

    def passes(n):
        for _ in range(n):
            rc, nz, err = ctx.fill(False, False, True, assign=True, predict_threshold=0.8, store_rows=False, defer=True)
            assert rc == 0, rc
        rc, nz, err = ctx.fill_result()
        assert rc == 0, rc
        ctx.synchronize()

    passes(12)
    steps = max(3, args.steps // 2)
    tot0 = ctx.timer_totals()
    t0 = time.perf_counter()
    passes(steps)
    dt = time.perf_counter() - t0
    tot1 = ctx.timer_totals()
    lap = {k: (tot1[k][0] - tot0[k][0]) / max(1, tot1[k][1] - tot0[k][1]) for k in ("fill", "predict")}
    bpl = 24.0 * A / M + 16.0
    info = ctx.info()
    return {"workload": "%s, %d frames%s" % (synth.CONFIG_TEXT.get(cfg, cfg), F, " (one GPU's share of the 8-GPU run)" if cfg in ("C4", "C5") else ""),
            "value": F * M * steps / dt, "unit": "lvec/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "stages_ms": {"fill": float(lap["fill"]), "predict": float(lap["predict"])},
            "roofline": {"bound": "hbm", "kernel": "k_fill%d" % info["fill_kernel"], "kernel_ms": float(lap["fill"]),
                         "algorithmic_bytes_per_lvec": bpl, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": F * M * bpl / (lap["fill"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_step": F * M * bpl / dt * steps / 1e9 / HBM_PEAK_GBS},
            "fill_shape": {k: info.get(k) for k in ("waves_per_workgroup", "frames_per_workgroup", "assignment_fused")},
            "end_to_end_run": {"algo": kw.get("clustering_algorithm", "dotprod"), "seconds": round(t_e2e, 4), "cold_seconds": round(t_cold, 4),
                               "run_seconds": round(t_run, 4), "warm_runs_seconds": [round(w[0], 4) for w in warm],
                               "jumps": n_jumps, "sites": int(st.site_network.n_sites),
                               "lvec_per_s": round(F * M / t_e2e, 1)}}


def lib_sha():
    p = os.path.join(ROOT, "sitator_amd", "lib", "libsitator_hip.so")
    h = hashlib.sha256()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def measured_traffic(kernel):
    """HBM bytes per launch of the fill kernel from the PMC passes kept under profiles/ (FETCH_SIZE x 2 + WRITE_SIZE,
    MI355X_MICROARCH.md) - only if they were taken with THIS build of the library, else null."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(tpath))
        if rec.get("lib_sha16") == lib_sha() and rec.get("kernel") == kernel:
            return rec.get("bytes_per_launch")
    except Exception:
        pass
    return None


FP64_LANE_INSTS_PER_S = 256 * 4 * 16 * 2.4e9


def valu_fraction(valu, n_lvec, kernel_ms):
    if not valu or "floor_insts_per_ion" not in valu or not kernel_ms:
        return None
    return valu["floor_insts_per_ion"] * n_lvec * 64.0 / (kernel_ms * 1e-3) / FP64_LANE_INSTS_PER_S


def limiter_of(valu):
    """What the SQ counters say keeps the kernel below its HBM roofline: the SIMDs' vector-issue slots (issue_frac
    >= 0.7), or the waves' waits (memory / LDS latency that the resident waves do not cover)."""
    if not valu or "issue_frac" not in valu:
        return None
    if valu["issue_frac"] >= 0.7:
        return "valu-issue (SIMDs issue a vector instruction in %.0f %% of the cycles)" % (100 * valu["issue_frac"])
    return "latency (vector issue in %.0f %% of the cycles, waves waiting %.0f %% of their time)" % (
        100 * valu["issue_frac"], 100 * valu.get("wait_frac", float("nan")))


def measured_valu(kernel, config):
    """Vector-instruction counters of the fill kernel from the PMC passes kept under profiles/ (SURVEY.md section 8d:
    the FP64-VALU side of the roofline): wave instructions per landmark vector, the FP64 arithmetic among them (the
    floor), and the fraction of the kernel's cycles in which a SIMD issues a vector instruction - only if they were
    taken with THIS build of the library and this configuration, else null."""
    vpath = os.path.join(ROOT, "profiles", "pmc_valu.json")
    try:
        rec = json.load(open(vpath))
        if rec.get("lib_sha16") == lib_sha() and rec.get("kernel") == kernel and rec.get("config") == config:
            return {k: rec[k] for k in ("insts_per_ion", "floor_insts_per_ion", "issue_frac", "wait_frac", "salu_per_ion") if k in rec}
    except Exception:
        pass
    return None


def cpu_baseline(host, gen, frames, ref, centers, M, frames_per_core, ncpu):
    """The oracle's C port of the same pass (wrap, static check, dense landmark vectors, predict) on leading cuts of
    the same trajectory: one thread (how the reference runs), and one cut per host core in parallel (ctypes releases
    the GIL; the reference itself has no threading)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    orc.lib()
    ref_static = ref[gen.static_mask]
    verts, vcd = orc.site_vertex_distances(host.cell, host.centers, host.vertices, ref_static)
    sidx, midx = np.where(gen.static_mask)[0], np.where(gen.mobile_mask)[0]

    def one(cut):
        wrapped = orc.wrap_points(host.cell, cut)
        lv, _ = orc.fill(host.cell, wrapped, sidx, midx, ref_static, verts, vcd)
        orc.predict(lv, centers, 0.8, True)
        return len(cut)

    n1 = min(len(frames), frames_per_core)
    t0 = time.perf_counter()
    one(frames[:n1])
    dt1 = time.perf_counter() - t0
    nthr = min(ncpu, 16)                      # the CPU share of a one-GPU box
    cuts = [frames[i * n1:(i + 1) * n1] for i in range(nthr) if (i + 1) * n1 <= len(frames)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=len(cuts)) as ex:
        nall = sum(ex.map(one, cuts))
    dta = time.perf_counter() - t0
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n1 * M / dt1, "unit": "lvec/s", "cores": 1, "kind": "port",
            "sample": "leading %d frames of the same workload (%d landmark vectors), %.1f s" % (n1, n1 * M, dt1),
            "all_cores": {"value": nall * M / dta, "cores": len(cuts),
                          "sample": "%d cuts of %d frames in parallel threads, %.1f s" % (len(cuts), n1, dta)},
            "cpu_model": model, "host_cores": ncpu}


if __name__ == "__main__":
    main()
