#!/usr/bin/env python3
"""How many queries would a ROW-major scan score each 64-row tile against?  Rows are laid out in the bucket
order of tree 0 (neighbouring rows share most buckets of the other trees too); a tile's query set is the union
of the queries that have any of its rows among their candidates.  Prints the mean and quantiles of that set's
size for a workload of bench.py, from a sample of the batch's queries scaled up to the whole batch.

  python3 tools/union_sizes.py [workload = c3] [sampled queries = 1000]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from similaritysearchbyrdf_amd._lib import DpfIndex  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
params, X, Q, pick, steps = bench.workload(wl, 1000000, 10000)
n, nq = X.shape[0], Q.shape[0]
ix = DpfIndex(params, metric="l2", dtype=np.float32)
ix.fit(X)
roots, dnodes, leaf_off, ids = ix.snapshot()
first = np.asarray(ids[:n])
assert np.array_equal(np.sort(first), np.arange(n)), "the first n bucket entries are not tree 0's"
tile_of = np.empty(n, dtype=np.int64)
tile_of[first] = np.arange(n) // 64
ntiles = (n + 63) // 64
per_tile = np.zeros(ntiles, dtype=np.int64)
pairs = 0
for s in range(0, m, 250):
    off, cand = ix.query_batch(pick[s:s + 250].astype(np.int32), Q[s:s + 250], steps=steps)
    for q in range(len(off) - 1):
        t = np.unique(tile_of[cand[off[q]:off[q + 1]]])
        per_tile[t] += 1
        pairs += off[q + 1] - off[q]
scale = nq / float(m)
sizes = per_tile * scale
print("%s: %d queries sampled; candidates per query %.0f; per 64-row tile (rows in tree 0's bucket order) the "
      "union holds mean %.0f queries (median %.0f, p90 %.0f, max %.0f) of %d; scored pairs %.3g against %.3g candidate "
      "pairs" % (wl, m, pairs / float(m), sizes.mean(), np.median(sizes), np.quantile(sizes, 0.9), sizes.max(), nq,
                 sizes.sum() * 64, pairs * scale))
