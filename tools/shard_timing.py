"""Per-rank batch time of a sharded handle on one GPU (collectives excluded):
python tools/shard_timing.py <mode: tables|rows> <shard groups> [workload] [query groups] [seed sample]
times the rank that holds the largest shard (most tables / most rows) on its share of the queries, rotating
through four distinct query batches."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from similaritysearchbyrdf_amd import synthetic
from similaritysearchbyrdf_amd._lib import DpfIndex
from similaritysearchbyrdf_amd.dist import table_bounds, shard_bounds, num_tables_of

mode = sys.argv[1]
sg = int(sys.argv[2])
wl = sys.argv[3] if len(sys.argv) > 3 else "c3"
qg = int(sys.argv[4]) if len(sys.argv) > 4 else 1
seed_sample = int(sys.argv[5]) if len(sys.argv) > 5 else 0
n_base, n_q = (1250000, 12500) if wl == "c5" else (1000000, 10000)
params, X, Q, pick, steps = bench.workload(wl, n_base, n_q)
L = num_tables_of(params)
kw = {}
if mode == "tables":
    spans = [table_bounds(L, sg, r) for r in range(sg)]
    tb, te = max(spans, key=lambda s: s[1] - s[0])
    kw = dict(table_begin=tb, table_end=te)
    what = "tables [%d, %d) of %d" % (tb, te, L)
else:
    lo, hi = shard_bounds(X.shape[0], sg, 0)
    kw = dict(row_begin=lo, row_end=hi)
    what = "rows [%d, %d) of %d" % (lo, hi, X.shape[0])
dev = torch.device("cuda", 0)
ix = DpfIndex(params, metric="l2", top_k=10, dtype=np.float32, device=0, **kw)
if seed_sample:
    ix.set_option("seed_sample", seed_sample)
ix.fit(X)
nq, k = Q.shape[0] // qg, 10
batches = [(Q[:nq], pick[:nq])] + [synthetic.more_queries(bench.QUERY_RECIPE[wl], X, nq, seed=9000 + b) for b in range(1, 4)]
dB = [(torch.from_numpy(np.ascontiguousarray(q)).to(dev), torch.from_numpy(np.ascontiguousarray(p).astype(np.int32)).to(dev)) for q, p in batches]
d_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
d_sc = torch.empty((nq, k), dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for it in range(4):
    ix.query_topk_device(dB[it % 4][1].data_ptr(), dB[it % 4][0].data_ptr(), nq, steps, k, d_ids.data_ptr(), d_sc.data_ptr(), stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for it in range(n):
    ix.query_topk_device(dB[it % 4][1].data_ptr(), dB[it % 4][0].data_ptr(), nq, steps, k, d_ids.data_ptr(), d_sc.data_ptr(), stream)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
st = ix.stats()
print("%s %s: query groups %d x shard groups %d: %s, %d queries: %.3f ms per batch, forwarded %.0f per query, path %d" % (
    wl, mode, qg, sg, what, nq, ms, st["last_forwarded"] / nq, st["last_path"]), flush=True)
