"""Per-rank batch time of a sharded handle on one GPU: python tools/shard_timing.py <table groups> [workload]
[query groups] times the rank that holds the most tables, on its share of the queries (collectives excluded)."""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from similaritysearchbyrdf_amd._lib import DpfIndex
from similaritysearchbyrdf_amd.dist import table_bounds, num_tables_of

world = int(sys.argv[1])
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
qg = int(sys.argv[3]) if len(sys.argv) > 3 else 1
params, X, Q, pick, steps = bench.workload(wl, 1000000, 10000)
Q, pick = Q[:Q.shape[0] // qg], pick[:Q.shape[0] // qg]
L = num_tables_of(params)
spans = [table_bounds(L, world, r) for r in range(world)]
tb, te = max(spans, key=lambda s: s[1] - s[0])
dev = torch.device("cuda", 0)
ix = DpfIndex(params, metric="l2", top_k=10, dtype=np.float32, device=0, table_begin=tb, table_end=te)
ix.fit(X)
nq, k = Q.shape[0], 10
dQ = torch.from_numpy(Q).to(dev)
dqid = torch.from_numpy(pick.astype(np.int32)).to(dev)
d_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
d_sc = torch.empty((nq, k), dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for it in range(3):
    ix.query_topk_device(dqid.data_ptr(), dQ.data_ptr(), nq, steps, k, d_ids.data_ptr(), d_sc.data_ptr(), stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for it in range(n):
    ix.query_topk_device(dqid.data_ptr(), dQ.data_ptr(), nq, steps, k, d_ids.data_ptr(), d_sc.data_ptr(), stream)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
st = ix.stats()
print("query groups %d, table groups %d: tables [%d, %d) of %d: %.3f ms per batch, forwarded %.0f per query, path %d" % (
    qg, world, tb, te, L, ms, st["last_forwarded"] / nq, st["last_path"]))
