"""Fit timing on the c3 base (1M x 100-d, 30 trees): device forest build against the host builder."""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from similaritysearchbyrdf_amd._lib import DpfIndex

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
params, X, Q, qids, _ = bench.workload(wl, 1000000, 10000)
for flags, name in ((0, "device"), (4, "host"), (0, "device")):
    ix = DpfIndex(dict(params, flags=flags), metric="l2")
    t0 = time.time()
    ix.fit(X)
    dt = time.time() - t0
    st = ix.stats()
    print("%s forest: fit %.1f ms (hash+copy %.1f, forest %.1f, total %.1f); %d directories, %d buckets, max %d, misflag %d" % (
        name, dt * 1e3, st["fit_ms_hash"], st["fit_ms_forest"], st["fit_ms_total"], st["num_dnodes"], st["num_leaves"],
        st["max_leaf"], st["misflag_events"]), flush=True)
    del ix
