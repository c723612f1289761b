"""Runs the scan's parity tests against a -DDPF_DEBUG_BOUNDS build of the library (make -C similaritysearchbyrdf_amd/csrc
debug-bounds -> tools/bin/libdpf_dbg.so), whose tile and query loaders count every index outside its array instead of
following it, and reports the count summed over every handle the tests made:
python tools/debug_bounds_run.py [pytest selection ...]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import pytest
import similaritysearchbyrdf_amd._lib as L

L.LIB_PATH = os.path.abspath(os.path.join(ROOT, "tools", "bin", "libdpf_dbg.so"))
total = {"handles": 0, "violations": 0, "topk_batches": 0}
_close = L.DpfIndex.close


def close(self):
    if getattr(self, "h", None) and self.h.value:
        try:
            st = self.stats()
            total["handles"] += 1
            total["violations"] += st["debug_bound_violations"]
            total["topk_batches"] += st["total_topk_batches"]
        except Exception as e:              # a handle that never fitted
            print("stats failed:", e)
    _close(self)


L.DpfIndex.close = close
L.DpfIndex.__del__ = lambda self: close(self)
sel = sys.argv[1:] or ["tests/test_gpu_parity.py", "-k", "bucket_major_rerank_equals_gather_rerank or random_configurations_long_rows or cut_rows"]
rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", "--timeout", "300", "--timeout-method", "thread"] + sel)
import gc
gc.collect()
print("debug-bounds build: %d handles, %d top-k batches, %d out-of-bounds indices counted" % (total["handles"], total["topk_batches"], total["violations"]))
sys.exit(int(rc) or (1 if total["violations"] else 0))
