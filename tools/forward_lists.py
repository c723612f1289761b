"""Distribution of the forward lists of one batch (rows the scan forwards per query, rows the exact rerank scores in the
end): python tools/forward_lists.py [workload]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from similaritysearchbyrdf_amd._lib import DpfIndex

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
params, X, Q, pick, steps = bench.workload(wl, 1000000, 10000)
ix = DpfIndex(params, metric="l2", top_k=10)
ix.fit(X)
for rep in range(3):
    ix.query_topk(pick.astype(np.int32), Q, steps=steps, k=10)
f, k = ix.last_forward_counts(Q.shape[0])
for name, a in (("forwarded", f), ("kept", k)):
    print("%s %s: mean %.1f, median %d, p90 %d, p99 %d, p99.9 %d, max %d; queries above 128: %d, above 256: %d, above 512: %d" % (
        wl, name, a.mean(), np.median(a), np.percentile(a, 90), np.percentile(a, 99), np.percentile(a, 99.9), a.max(),
        (a > 128).sum(), (a > 256).sum(), (a > 512).sum()))
