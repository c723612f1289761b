"""Throughput of the Set-returning batch query (dpf_query_batch = DensevectorRDFInit.NewMultiThreadQueryBatch: host
query matrix in, per-query sorted unique candidate lists out in host memory):
python tools/sets_timing.py [workload] [queries]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from similaritysearchbyrdf_amd._lib import DpfIndex

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
params, X, Q, pick, steps = bench.workload(wl, 1000000, nq)
ix = DpfIndex(params, metric="l2", top_k=10)
ix.fit(X)
pick = pick.astype(np.int32)
for rep in range(4):
    t0 = time.perf_counter()
    off, ids = ix.query_batch(pick, Q, steps=steps, copy=False)
    dt = time.perf_counter() - t0
    print("%s: %d queries, %d-step search: %.1f ms per batch = %.0f queries/s; %.1f k ids per Set, %.2f GB of ids to the host" % (
        wl, nq, steps, dt * 1e3, nq / dt, off[-1] / nq / 1e3, off[-1] * 4 / 1e9), flush=True)
t0 = time.perf_counter()
off, ids = ix.query_batch(pick, Q, steps=steps, copy=True)
print("with a copy out of the handle's buffer: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
t0 = time.perf_counter()
off = ix.query_batch_resident(pick, Q, steps=steps)
t1 = time.perf_counter()
some = ix.sets_fetch(0, 100)
t2 = time.perf_counter()
ti, ts = ix.sets_topk(10)
t3 = time.perf_counter()
print("resident Sets: sizes in %.1f ms; the first 100 queries' lists (%.1f MB) in %.2f ms; top-10 of all %d Sets (gather rerank) in %.1f ms" % (
    (t1 - t0) * 1e3, some.nbytes / 1e6, (t2 - t1) * 1e3, nq, (t3 - t2) * 1e3), flush=True)
