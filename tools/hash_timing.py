"""Time of the two hashing paths on one GPU: python tools/hash_timing.py <rows> <dim> <functions> [family]
(the matrix-core projection with its exact re-evaluation against the exact vector-ALU kernel alone)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from similaritysearchbyrdf_amd import synthetic
from similaritysearchbyrdf_amd._lib import DpfIndex

n, dim, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
family = sys.argv[4] if len(sys.argv) > 4 else "angle"
params = synthetic.make_family(dim, table_num=10, permutation_num=3 if family == "angle" else 1, chain_length=32, family=family,
                               family_size=F, partition_bits=3, orthogonal=(family == "angle"))
rng = np.random.default_rng(1)
X = rng.standard_normal((n, dim), dtype=np.float32)
order = (("exact vector-ALU kernel", 0), ("matrix-core projection", 1))
if len(sys.argv) > 5 and sys.argv[5] == "swap":
    order = order[::-1]
for name, opt in order:
    ix = DpfIndex(params, dtype=np.float32)
    ix.set_option("projection_min_rows", opt)
    ix.fit(X[:1000])
    t0 = time.perf_counter()
    ix.fit(X)
    st = ix.stats()
    flop = 2.0 * n * dim * F
    print("%s: %d x %d rows against %d functions: copy-in + hashing %.1f ms (%.1f TFLOP/s incl. the copies), undecided rows %d" % (
        name, n, dim, F, st["fit_ms_hash"], flop / (st["fit_ms_hash"] * 1e-3) / 1e12, st["hash_undecided_rows"]), flush=True)
    ix.close()
