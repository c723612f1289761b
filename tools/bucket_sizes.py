"""Bucket-size statistics of a fitted forest: python tools/bucket_sizes.py <workload> [base rows] [dim]
(the size-biased mean is the expected size of the bucket a random stored row - or a query near one - falls into)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from similaritysearchbyrdf_amd._lib import DpfIndex

wl = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
params, X, Q, pick, steps = bench.workload(wl, n, 100)
ix = DpfIndex(params, metric="l2", top_k=10)
ix.fit(X)
roots, dnodes, leaf_off, ids = ix.snapshot()
sz = np.diff(leaf_off.astype(np.int64))
sz = sz[sz > 0]
print("%s: %d tables, %d buckets, mean %.1f, size-biased mean %.1f, max %d" % (
    wl, ix.L, sz.size, sz.mean(), (sz * sz).sum() / sz.sum(), sz.max()), flush=True)
