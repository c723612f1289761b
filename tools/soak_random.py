"""One-off soak: the seeded configuration sweep of tests/test_gpu_parity.py over many more seeds
(scan path == gather path == oracle on every draw): python tools/soak_random.py <first seed> <last seed> [long]
(`long`: rows of 300 to 1000 dimensions instead of 8 to 256)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_gpu_parity import test_random_configurations

lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
long_rows = len(sys.argv) > 3 and sys.argv[3] == "long"
for seed in range(lo, hi):
    if long_rows:
        test_random_configurations(seed, dims=(300, 400, 512, 640, 768, 1000))
    else:
        test_random_configurations(seed)
    if seed % 50 == 0:
        print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
print("seeds %d..%d: all equal" % (lo, hi - 1), flush=True)
