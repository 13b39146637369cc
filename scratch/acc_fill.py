"""Largest relative difference between the rows of k_fill3 and of the general kernel (library sqrt / exp / pow, the
reference's expressions) on a few hosts, and whether the zero patterns agree:  scratch/acc_fill.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests.test_gpu_kernels import _setup
from sitator_amd import synth
for cfg, M, F in (("C2", 64, 400), ("C1b", 4, 500), ("C5", 160, 60), ("C3", 448, 20)):
    host = synth.config_host(cfg)
    out = []
    for kern in ("1", "3"):
        ctx, *_ = _setup(host, M, F, seed=77, kernel=kern)
        rc, nz, err = ctx.fill()
        assert rc == 0
        out.append(ctx.rows_dense())
    a, b = out
    m = a != 0
    rel = np.abs(b[m] / a[m] - 1)
    print("%-4s pattern equal %s, entries %d, max rel %.3g, mean rel %.3g, 99.9%% %.3g" % (cfg, np.array_equal(m, b != 0), m.sum(), rel.max(), rel.mean(), np.quantile(rel, 0.999)), flush=True)
