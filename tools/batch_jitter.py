"""Wall time of every batch of a workload, with the per-stage event timings of the slow ones."""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from similaritysearchbyrdf_amd._lib import DpfIndex

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sync_each = len(sys.argv) > 3 and sys.argv[3] == "sync"
params, X, Q, pick, steps = bench.workload(wl, 1000000, 10000)
dev = torch.device("cuda", 0)
ix = DpfIndex(params, metric="l2", top_k=10, dtype=np.float32, device=0)
ix.fit(X)
ix.set_profiling(True)
nq, k = Q.shape[0], 10
dQ = torch.from_numpy(Q).to(dev)
dqid = torch.from_numpy(pick.astype(np.int32)).to(dev)
d_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
d_sc = torch.empty((nq, k), dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
walls = []
for b in range(nb):
    t0 = time.perf_counter()
    ix.query_topk_device(dqid.data_ptr(), dQ.data_ptr(), nq, steps, k, d_ids.data_ptr(), d_sc.data_ptr(), stream)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    walls.append((t2 - t0) * 1e3)
    if walls[-1] > 1.5 * min(walls):
        st = ix.stats()
        print("batch %d: %.2f ms (call returned after %.2f ms): hash %.2f probe %.2f dedupe %.2f rerank %.2f scan %.2f, forwarded %d" % (
            b, walls[-1], (t1 - t0) * 1e3, st["last_ms_hash"], st["last_ms_probe"], st["last_ms_dedupe"], st["last_ms_rerank"],
            st["last_ms_scan"], st["last_forwarded"]), flush=True)
w = np.array(walls)
print("batches %d: min %.2f median %.2f p90 %.2f max %.2f ms" % (nb, w.min(), np.median(w), np.percentile(w, 90), w.max()))
