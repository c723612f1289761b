"""ctypes binding of the CPU oracle (oracle/libdpf_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by the smoke check and by the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdpf_oracle.so")

FAMILY = {"angle": 0, "pStable": 1}
TYPE_OF_INDEX = {"original": 0, "sampling": 1, "continueBitsCount": 2, "angleNewMethod": 3}
SELF = {"compat": 0, "always": 1, "never": 2}
METRIC = {"dot": 0, "l2": 1}


class OrcConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "family", "dim", "num_functions", "num_tables", "chain_length",
        "partition_bits", "bucket_bits", "dir_node_size", "bucket_overflow",
        "type_of_index", "exclude_self")]


def build(force=False):
    """Compile the oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "dpf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    dp, ip, i64p = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    L.orc_dot_dense.restype = C.c_double
    L.orc_dot_dense.argtypes = [dp, dp, C.c_int32]
    L.orc_dot_sparse_sparse.restype = C.c_double
    L.orc_dot_sparse_sparse.argtypes = [C.c_int32, ip, dp, C.c_int32, ip, dp, C.c_int32]
    L.orc_angle_hash_dense.restype = C.c_int32
    L.orc_angle_hash_dense.argtypes = [dp, C.c_int32, C.c_int32, dp]
    L.orc_pstable_hash_dense.restype = C.c_int32
    L.orc_pstable_hash_dense.argtypes = [dp, dp, ip, C.c_int32, C.c_int32, dp]
    L.orc_pstable_hash_from_sums.restype = C.c_int32
    L.orc_pstable_hash_from_sums.argtypes = [dp, dp, ip, C.c_int32]
    L.orc_d2i.restype = C.c_int32
    L.orc_d2i.argtypes = [C.c_double]
    L.orc_bytes_hashcode.restype = C.c_int32
    L.orc_bytes_hashcode.argtypes = [C.POINTER(C.c_int8), C.c_int32]
    L.orc_continue_bits_count.restype = C.c_int32
    L.orc_continue_bits_count.argtypes = [C.c_int32, ip]
    L.orc_sampling_one_key.restype = C.c_int32
    L.orc_sampling_one_key.argtypes = [C.c_int32, ip]
    L.orc_scala_shuffle_0_31.restype = None
    L.orc_scala_shuffle_0_31.argtypes = [C.c_int64, ip]
    L.orc_angle_distance.restype = C.c_double
    L.orc_angle_distance.argtypes = [C.c_int32]
    L.orc_angle_new_method.restype = C.c_int32
    L.orc_angle_new_method.argtypes = [C.c_int32]
    L.orc_dir_offset_from_slot.restype = C.c_int32
    L.orc_dir_offset_from_slot.argtypes = [ip, C.c_int32, C.c_int32]
    L.orc_step_sub_indexes.restype = None
    L.orc_step_sub_indexes.argtypes = [C.c_int32, C.c_int32, C.c_int32, ip, ip]
    L.orc_probe_count.restype = C.c_int32
    L.orc_probe_count.argtypes = [C.c_int32]
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.POINTER(OrcConfig), dp, dp, ip, ip, dp, dp, ip, ip]
    L.orc_destroy.restype = None
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_hash.restype = C.c_int32
    L.orc_hash.argtypes = [C.c_void_p, C.c_int32, dp]
    L.orc_partition.restype = C.c_int32
    L.orc_partition.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.orc_fit.restype = C.c_int
    L.orc_fit.argtypes = [C.c_void_p, dp, C.c_int64, C.c_int32]
    L.orc_size.restype = C.c_int64
    L.orc_size.argtypes = [C.c_void_p]
    L.orc_misflag_events.restype = C.c_int64
    L.orc_misflag_events.argtypes = [C.c_void_p]
    L.orc_short_hash_events.restype = C.c_int64
    L.orc_short_hash_events.argtypes = [C.c_void_p]
    L.orc_get_similar.restype = C.c_int64
    L.orc_get_similar.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dp, C.c_int32, ip, C.c_int64]
    L.orc_query_batch.restype = C.c_int
    L.orc_query_batch.argtypes = [C.c_void_p, ip, dp, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                  i64p, C.POINTER(ip)]
    L.orc_free.restype = None
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_topk.restype = C.c_int
    L.orc_topk.argtypes = [C.c_void_p, dp, C.c_int64, i64p, ip, C.c_int32, C.c_int32, C.c_int32, ip, dp]
    L.orc_table_stats.restype = None
    L.orc_table_stats.argtypes = [C.c_void_p, C.c_int32, i64p, i64p, i64p]
    L.orc_set_probe_own.restype = None
    L.orc_set_probe_own.argtypes = [C.c_void_p, C.c_int32]
    L.orc_set_subindex_shard.restype = None
    L.orc_set_subindex_shard.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.orc_fitted_hashes.restype = None
    L.orc_fitted_hashes.argtypes = [C.c_void_p, C.c_int32, ip, ip]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


class OracleIndex:
    """The reference's DPF forest + batch query, restated on the CPU.

    params: dict with keys family, dim, fns [F,d], b [F], w [F], chain_fn [L,m],
    part_fns [L,pb,32], part_b [L,pb], part_w [L,pb], partition_bits,
    bucket_bits, dir_node_size, bucket_overflow, type_of_index, exclude_self,
    sampling_index [32].
    """

    def __init__(self, params):
        p = params
        self.lib = lib()
        self.fns = _f64(p["fns"])
        self.chain_fn = _i32(p["chain_fn"])
        F, d = self.fns.shape
        Lt, m = self.chain_fn.shape
        pb = int(p.get("partition_bits", 3))
        self.cfg = OrcConfig(
            family=FAMILY[p.get("family", "angle")], dim=d, num_functions=F, num_tables=Lt,
            chain_length=m, partition_bits=pb, bucket_bits=int(p.get("bucket_bits", 28)),
            dir_node_size=int(p.get("dir_node_size", 32)),
            bucket_overflow=int(p.get("bucket_overflow", 500)),
            type_of_index=TYPE_OF_INDEX[p.get("type_of_index", "original")],
            exclude_self=SELF[p.get("exclude_self", "compat")])
        self.b = _f64(p.get("b"))
        self.w = _i32(p.get("w"))
        self.part_fns = _f64(p.get("part_fns"))
        self.part_b = _f64(p.get("part_b"))
        self.part_w = _i32(p.get("part_w"))
        self.sampling_index = _i32(p.get("sampling_index"))
        if pb > 0:
            assert self.part_fns.shape == (Lt, pb, 32), self.part_fns.shape
        self.dim, self.L, self.k_default = d, Lt, 10
        self.h = self.lib.orc_create(C.byref(self.cfg), _dp(self.fns), _dp(self.b), _ip(self.w),
                                     _ip(self.chain_fn), _dp(self.part_fns), _dp(self.part_b),
                                     _ip(self.part_w), _ip(self.sampling_index))

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_probe_own(self, on):
        """True: later queries walk the hash itself only (RandomDrawTreeMap.java:630-675)."""
        self.lib.orc_set_probe_own(self.h, 1 if on else 0)

    def set_subindex_shard(self, group, groups):
        self.lib.orc_set_subindex_shard(self.h, group, groups)

    def hash(self, table, v):
        v = _f64(v)
        return int(self.lib.orc_hash(self.h, table, _dp(v)))

    def hashes(self, X):
        X = _f64(X)
        out = np.empty((X.shape[0], self.L), dtype=np.int32)
        for i in range(X.shape[0]):
            for t in range(self.L):
                out[i, t] = self.lib.orc_hash(self.h, t, _dp(X[i]))
        return out

    def partition(self, table, h):
        return int(self.lib.orc_partition(self.h, table, int(np.int32(h))))

    def fit(self, X, threads=8):
        X = _f64(X)
        assert X.ndim == 2 and X.shape[1] == self.dim
        self.lib.orc_fit(self.h, _dp(X), X.shape[0], threads)

    @property
    def size(self):
        return int(self.lib.orc_size(self.h))

    @property
    def misflag_events(self):
        return int(self.lib.orc_misflag_events(self.h))

    @property
    def short_hash_events(self):
        return int(self.lib.orc_short_hash_events(self.h))

    def fitted_hashes(self, table):
        n = self.size
        h = np.empty(n, dtype=np.int32)
        p = np.empty(n, dtype=np.int32)
        self.lib.orc_fitted_hashes(self.h, table, _ip(h), _ip(p))
        return h, p

    def get_similar(self, table, key, v, steps):
        v = _f64(v)
        cap = 1 << 16
        while True:
            out = np.empty(cap, dtype=np.int32)
            n = self.lib.orc_get_similar(self.h, table, key, _dp(v), steps, _ip(out), cap)
            if n <= cap:
                return out[:n].copy()
            cap = int(n)

    def query_batch(self, qids, Q, steps=0, threads=5, mode=0):
        """-> (offsets[nq+1] int64, ids int32): per query the sorted unique Set."""
        qids = _i32(qids)
        Q = _f64(Q)
        nq = Q.shape[0]
        offsets = np.zeros(nq + 1, dtype=np.int64)
        pids = C.POINTER(C.c_int32)()
        self.lib.orc_query_batch(self.h, _ip(qids), _dp(Q), nq, steps, threads, mode,
                                 _i64p(offsets), C.byref(pids))
        total = int(offsets[nq])
        ids = np.ctypeslib.as_array(pids, shape=(max(total, 1),))[:total].copy()
        self.lib.orc_free(pids)
        return offsets, ids

    def topk(self, Q, offsets, cand, k=10, metric="dot", threads=8):
        Q = _f64(Q)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        cand = _i32(cand)
        nq = Q.shape[0]
        out_ids = np.empty((nq, k), dtype=np.int32)
        out_sc = np.empty((nq, k), dtype=np.float64)
        self.lib.orc_topk(self.h, _dp(Q), nq, _i64p(offsets), _ip(cand), k, METRIC[metric], threads,
                          _ip(out_ids), _dp(out_sc))
        return out_ids, out_sc

    def table_stats(self, table):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self.lib.orc_table_stats(self.h, table, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value
