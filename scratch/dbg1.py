import sys, os; sys.path.insert(0, '.')
import numpy as np
from tests.test_gpu_kernels import _setup
from oracle import oracle
from sitator_amd import synth
host = synth.config_host("C2")
ctx, frames, sm, mm, ref = _setup(host, 64, 300, seed=5)
assert ctx.fill()[0] == 0
X = ctx.rows_dense()
centers = oracle.fit_centers(X, 0.45)
normed = centers / np.linalg.norm(centers, axis=1)[:, None]
ctx.set_centers(normed, True)
lab_a, conf_a, cnt_a = ctx.predict(0.8)
rc, _, _ = ctx.fill(assign=True, predict_threshold=0.8, store_rows=False)
lab_b, conf_b, cnt_b = ctx.assignments()
bad = np.where(lab_a != lab_b)[0]
print('K', len(centers), 'mismatch', len(bad), bad[:20])
for r in bad[:8]:
    print(r, lab_a[r], conf_a[r], lab_b[r], conf_b[r], 'nnz', np.count_nonzero(X[r]), np.nonzero(X[r])[0], X[r][X[r]!=0])
print('conf diff', np.abs(conf_a-conf_b).max())
