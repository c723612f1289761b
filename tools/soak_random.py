"""One-off soak: the seeded configuration sweep of tests/test_gpu_parity.py over many more seeds
(scan path == gather path == oracle on every draw): python tools/soak_random.py <first seed> <last seed>"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests.test_gpu_parity import test_random_configurations

lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
for seed in range(lo, hi):
    test_random_configurations(seed)
    if seed % 50 == 0:
        print("seed %d ok (%.0f s)" % (seed, time.time() - t0), flush=True)
print("seeds %d..%d: all equal" % (lo, hi - 1), flush=True)
