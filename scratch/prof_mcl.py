"""Where the time of the mcl plugin goes (host sections timed with perf_counter): python scratch/prof_mcl.py [config] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup
from sitator_amd import synth
from sitator_amd.dotprod_classifier import LandmarkVectors
from sitator_amd.cluster import mcl
from sitator_amd.markov import markov_clustering
from scipy.sparse.linalg import eigsh
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
host = synth.config_host(cfg)
ctx, *_ = _setup(host, synth.CONFIG_MOBILE[cfg], F, seed=2)
assert ctx.fill()[0] == 0
X = LandmarkVectors(ctx)
T = time.perf_counter
t0 = T(); gram, seen = ctx.gram(); t1 = T()
cov = gram / X.shape[0]; graph = np.clip(mcl.cov2corr(cov), 0, None)
for i in range(X.shape[1]):
    if graph[i, i] == 0: graph[i, i] = 1
t2 = T(); groups = markov_clustering(graph, inflation=4); t3 = T()
groups = [list(g) for g in groups if seen[g[0]] > 0]
te = tb = 0.0
for g in groups:
    a = T()
    c = np.zeros(X.shape[1])
    if len(g) == 1: c[g] = 1.0
    else:
        _, vec = eigsh(cov[g][:, g], k=1); c[g] = vec.T
    b = T(); ctx.best_match(c); d = T()
    te += b - a; tb += d - b
print(cfg, "groups", len(groups), "gram %.3f  graph %.3f  markov %.3f  eigsh %.3f  best_match %.3f" % (t1 - t0, t2 - t1, t3 - t2, te, tb))
t = T(); out = mcl.do_landmark_clustering(X, {}, 0.01 / synth.CONFIG_MOBILE[cfg], False); print("whole plugin %.3f s" % (T() - t))
