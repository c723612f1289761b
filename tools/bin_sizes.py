"""usage (GPU box): python tools/bin_sizes.py <c2|c3|c4> : the bins the binned list building would cut from one batch of the
workload (the rule of k_bin_seed restated on the host over dpf_last_incidences' bucket counts): how even they come out."""
import sys
import numpy as np
sys.path.insert(0, '.')
import bench
from similaritysearchbyrdf_amd._lib import DpfIndex

wl = sys.argv[1]
params, X, Q, pick, steps = bench.workload(wl, 1000000, 10000)
ix = DpfIndex(params, metric="l2", top_k=10)
ix.fit(X)
ix.query_topk(pick.astype(np.int32), Q, steps=steps, k=10)
bc, raw, br, bl = ix.last_incidences(10000)
nl = len(bc)
shift = 4
while ((nl + (1 << shift) - 1) >> shift) > 8192:
    shift += 1
G = 1 << shift
pad = np.zeros(((nl + G - 1) // G) * G, dtype=np.int64)
pad[:nl] = bc
gran = pad.reshape(-1, G).sum(1)
target = max(int(gran.sum()) // 224, 1024)
if target > 32000:                     # (half of what k_bin_lists stages in LDS, unless that made more than about 700 bins)
    target = max(32000, int(gran.sum()) // 700)
max_gran = 8192 >> shift
excl = np.concatenate([[0], np.cumsum(gran)[:-1]])
key = excl // target                    # granules with one key form a run; a run is cut every max_gran granules
bins, acc, n_in, prev = [], 0, 0, None
for c, kx in zip(gran, key):
    if n_in > 0 and (kx != prev or n_in == max_gran):
        bins.append((acc, n_in)); acc = 0; n_in = 0
    acc += int(c); n_in += 1; prev = kx
if n_in:
    bins.append((acc, n_in))
pairs = np.array([b[0] for b in bins]); grans = np.array([b[1] for b in bins])
print(wl, "buckets", nl, "granule", G, "granules", len(gran), "largest granule", int(gran.max()), "target", target)
print("  bins", len(bins), "pairs/bin mean %.0f max %d min %d p90 %.0f" % (pairs.mean(), pairs.max(), pairs.min(), np.percentile(pairs, 90)),
      "buckets/bin max", int(grans.max()) * G, "largest bucket count", int(bc.max()))
