"""ctypes binding of the C ABI declared in include/dpf.h (libdpf.so).

There is no CPU path: importing works anywhere, but creating an index needs a
HIP device, and a missing libdpf.so raises at first use.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdpf.so")

FAMILY = {"angle": 0, "pStable": 1}
TYPE_OF_INDEX = {"original": 0, "sampling": 1, "continueBitsCount": 2, "angleNewMethod": 3}
SELF = {"compat": 0, "always": 1, "never": 2}
METRIC = {"dot": 0, "l2": 1}
DTYPE = {np.dtype(np.float32): 0, np.dtype(np.float64): 1}

DPF_OK, DPF_EINVAL, DPF_ENOTFIT, DPF_ENOFAMILY, DPF_EHIP, DPF_ENOMEM, DPF_ECAPACITY, DPF_EBUSY = 0, -1, -2, -3, -4, -5, -6, -7


class DpfConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "family", "dim", "num_functions", "num_tables", "chain_length", "partition_bits",
        "bucket_bits", "dir_node_size", "bucket_overflow", "type_of_index", "exclude_self",
        "metric", "top_k", "dtype", "device", "flags", "table_begin", "table_end", "subindex_group", "subindex_groups")] + [
        ("row_begin", C.c_int64), ("row_end", C.c_int64)]


class DpfStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "num_vectors", "num_dnodes", "num_leaves", "max_leaf", "misflag_events", "short_hash_events",
        "last_raw_candidates", "last_unique_candidates")] + [
        (n, C.c_double) for n in ("last_ms_hash", "last_ms_probe", "last_ms_dedupe", "last_ms_rerank",
                                  "last_ms_scan")] + [
        (n, C.c_int64) for n in ("last_path", "last_pairs", "last_forwarded", "last_scan_launches", "snapshot_bytes",
                                 "base_bytes")] + [
        (n, C.c_double) for n in ("fit_ms_hash", "fit_ms_forest", "fit_ms_total")] + [
        (n, C.c_int64) for n in ("last_overflowed", "last_sizes_carried", "total_topk_batches", "total_sized_on_host",
                                 "total_carried", "total_rerun", "total_overflowed", "total_gather_batches",
                                 "base16_bytes", "tile_rows", "num_tiles", "resident_ids", "hash_undecided_rows",
                                 "base8_bytes", "last_scan_int8", "bucket_order_bytes", "last_longest_list", "debug_bound_violations")]


EXPORTS = (
    "dpf_create", "dpf_destroy", "dpf_last_error", "dpf_set_hash_family", "dpf_set_partitioner",
    "dpf_set_sampling_index", "dpf_fit", "dpf_hash_batch", "dpf_query_batch", "dpf_free",
    "dpf_query_topk", "dpf_query_topk_device", "dpf_rerank", "dpf_merge_topk_device", "dpf_merge_topk_packed_device",
    "dpf_subindex_distribution", "dpf_get_stats", "dpf_last_rerank_profile", "dpf_set_profiling",
    "dpf_save", "dpf_load", "dpf_get_config", "dpf_snapshot_sizes", "dpf_snapshot_copy", "dpf_set_option", "dpf_query_topk_device_static",
    "dpf_load_shard", "dpf_set_probe_mode", "dpf_host_alloc", "dpf_host_free", "dpf_last_incidences",
    "dpf_query_batch_view", "dpf_query_batch_resident", "dpf_sets_fetch", "dpf_sets_topk",
    "dpf_sets_queries", "dpf_sets_count", "dpf_last_forward_counts",
)

# dpf_set_option keys (include/dpf.h)
OPTION = {"emit_cap": 1, "bucket_list_cap": 2, "sync_sizes": 3, "first_pass_permille": 4, "seed_sample": 5,
          "short_rerank": 6, "projection_min_rows": 7, "scan_passes": 8, "early_pass_permille": 9, "exact_seeds": 10, "keep_forwarded": 11, "side_stream": 12, "probe_lanes": 13,
          "defer_build": 14, "expected_rows": 15, "list_bins": 16}

_lib = None


class DpfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dpf error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Loads libdpf.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `make -C similaritysearchbyrdf_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, ip, dp, i64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64)
    L.dpf_create.restype = C.c_int
    L.dpf_create.argtypes = [C.POINTER(DpfConfig), C.POINTER(vp)]
    L.dpf_destroy.restype = None
    L.dpf_destroy.argtypes = [vp]
    L.dpf_last_error.restype = C.c_char_p
    L.dpf_last_error.argtypes = [vp]
    L.dpf_set_hash_family.restype = C.c_int
    L.dpf_set_hash_family.argtypes = [vp, dp, dp, ip, ip]
    L.dpf_set_partitioner.restype = C.c_int
    L.dpf_set_partitioner.argtypes = [vp, dp, dp, ip]
    L.dpf_set_sampling_index.restype = C.c_int
    L.dpf_set_sampling_index.argtypes = [vp, ip]
    L.dpf_fit.restype = C.c_int
    L.dpf_fit.argtypes = [vp, vp, C.c_int64]
    L.dpf_hash_batch.restype = C.c_int
    L.dpf_hash_batch.argtypes = [vp, vp, C.c_int64, ip, ip]
    L.dpf_query_batch.restype = C.c_int
    L.dpf_query_batch.argtypes = [vp, ip, vp, C.c_int64, C.c_int32, i64p, C.POINTER(ip)]
    L.dpf_query_batch_view.restype = C.c_int
    L.dpf_query_batch_view.argtypes = [vp, ip, vp, C.c_int64, C.c_int32, i64p, C.POINTER(ip)]
    L.dpf_query_batch_resident.restype = C.c_int
    L.dpf_query_batch_resident.argtypes = [vp, ip, vp, C.c_int64, C.c_int32, i64p]
    L.dpf_sets_fetch.restype = C.c_int
    L.dpf_sets_fetch.argtypes = [vp, C.c_int64, C.c_int64, ip]
    L.dpf_sets_queries.restype = C.c_int64
    L.dpf_sets_queries.argtypes = [vp]
    L.dpf_sets_count.restype = C.c_int64
    L.dpf_sets_count.argtypes = [vp, C.c_int64, C.c_int64]
    L.dpf_sets_topk.restype = C.c_int
    L.dpf_sets_topk.argtypes = [vp, C.c_int32, ip, dp]
    L.dpf_free.restype = None
    L.dpf_free.argtypes = [vp]
    L.dpf_query_topk.restype = C.c_int
    L.dpf_query_topk.argtypes = [vp, ip, vp, C.c_int64, C.c_int32, C.c_int32, ip, dp]
    L.dpf_query_topk_device.restype = C.c_int
    L.dpf_query_topk_device.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp]
    L.dpf_query_topk_device_static.restype = C.c_int
    L.dpf_query_topk_device_static.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, vp, vp, vp]
    L.dpf_rerank.restype = C.c_int
    L.dpf_rerank.argtypes = [vp, vp, C.c_int64, i64p, ip, C.c_int32, ip, dp]
    L.dpf_merge_topk_device.restype = C.c_int
    L.dpf_merge_topk_device.argtypes = [vp, vp, vp, C.c_int32, C.c_int64, C.c_int32, vp, vp, vp]
    L.dpf_merge_topk_packed_device.restype = C.c_int
    L.dpf_merge_topk_packed_device.argtypes = [vp, vp, C.c_int64, C.c_int32, C.c_int64, C.c_int32, vp, vp, vp]
    L.dpf_subindex_distribution.restype = C.c_int
    L.dpf_subindex_distribution.argtypes = [vp, dp]
    L.dpf_get_stats.restype = C.c_int
    L.dpf_get_stats.argtypes = [vp, C.POINTER(DpfStats)]
    L.dpf_last_rerank_profile.restype = C.c_int
    L.dpf_last_rerank_profile.argtypes = [vp, dp, i64p]
    L.dpf_set_profiling.restype = C.c_int
    L.dpf_set_profiling.argtypes = [vp, C.c_int32]
    L.dpf_set_option.restype = C.c_int
    L.dpf_set_option.argtypes = [vp, C.c_int32, C.c_int64]
    L.dpf_save.restype = C.c_int
    L.dpf_save.argtypes = [vp, C.c_char_p]
    L.dpf_load.restype = C.c_int
    L.dpf_load.argtypes = [C.c_char_p, C.c_int32, C.POINTER(vp)]
    L.dpf_load_shard.restype = C.c_int
    L.dpf_load_shard.argtypes = [C.c_char_p, C.c_int32, C.POINTER(DpfConfig), C.POINTER(vp)]
    L.dpf_set_probe_mode.restype = C.c_int
    L.dpf_set_probe_mode.argtypes = [vp, C.c_int32]
    L.dpf_snapshot_sizes.restype = C.c_int
    L.dpf_snapshot_sizes.argtypes = [vp, i64p]
    L.dpf_snapshot_copy.restype = C.c_int
    L.dpf_snapshot_copy.argtypes = [vp, vp, vp, vp, vp]
    L.dpf_get_config.restype = C.c_int
    L.dpf_get_config.argtypes = [vp, C.POINTER(DpfConfig)]
    L.dpf_last_incidences.restype = C.c_int
    L.dpf_last_incidences.argtypes = [vp, vp, vp, vp, vp, C.c_int32]
    L.dpf_last_forward_counts.restype = C.c_int
    L.dpf_last_forward_counts.argtypes = [vp, vp, vp]
    L.dpf_host_alloc.restype = C.c_int
    L.dpf_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.dpf_host_free.restype = None
    L.dpf_host_free.argtypes = [vp]
    _lib = L
    return L


class _PinnedBlock:
    """Owner of one dpf_host_alloc block; numpy arrays made over it keep it alive through their .base chain."""

    def __init__(self, nbytes):
        self.L_ = lib()
        self.ptr = C.c_void_p()
        rc = self.L_.dpf_host_alloc(max(int(nbytes), 1), C.byref(self.ptr))
        if rc != 0:
            raise DpfError(rc, self.L_.dpf_last_error(None).decode())
        self.nbytes = max(int(nbytes), 1)

    def __del__(self):
        try:
            if self.ptr.value:
                self.L_.dpf_host_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


def pinned_array(shape, dtype):
    """A numpy array in pinned host memory (dpf_host_alloc): the host-pointer entry points copy it by DMA from / to
    where it lies instead of staging it."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    block = _PinnedBlock(n * dtype.itemsize)
    buf = (C.c_char * block.nbytes).from_address(block.ptr.value)
    buf._owner = block                         # a from_address buffer does not own its memory: the array keeps buf, buf the block
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


class _HandleView(np.ndarray):
    """A numpy view of memory a handle owns: holds a reference to the handle so that it is not destroyed under the view."""

    def __new__(cls, arr, owner):
        obj = np.asarray(arr).view(cls)
        obj._owner = owner
        return obj

    def __array_finalize__(self, obj):
        self._owner = getattr(obj, "_owner", None)


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


class DpfIndex:
    """One forest + base shard on one GPU, driven through the C ABI.

    params: dict with fns [F,d], chain_fn [L,m], optional b/w [F], part_fns
    [L,pb,32], part_b/part_w [L,pb], family, partition_bits, bucket_bits,
    dir_node_size, bucket_overflow, type_of_index, exclude_self,
    sampling_index; plus metric, top_k, dtype, device, row_begin, row_end (row shard),
    table_begin, table_end (table shard).
    """

    def __init__(self, params, metric="dot", top_k=10, dtype=np.float32, device=0,
                 row_begin=0, row_end=-1, table_begin=0, table_end=0, subindex_group=0, subindex_groups=0):
        self.L_ = lib()
        p = params
        fns = _f64(p["fns"])
        chain_fn = _i32(p["chain_fn"])
        F, d = fns.shape
        Lt, m = chain_fn.shape
        pb = int(p.get("partition_bits", 3))
        self.np_dtype = np.dtype(dtype)
        self.cfg = DpfConfig(
            family=FAMILY[p.get("family", "angle")], dim=d, num_functions=F, num_tables=Lt, chain_length=m,
            partition_bits=pb, bucket_bits=int(p.get("bucket_bits", 28)),
            dir_node_size=int(p.get("dir_node_size", 32)), bucket_overflow=int(p.get("bucket_overflow", 500)),
            type_of_index=TYPE_OF_INDEX[p.get("type_of_index", "original")],
            exclude_self=SELF[p.get("exclude_self", "compat")], metric=METRIC[metric], top_k=top_k,
            dtype=DTYPE[self.np_dtype], device=device, flags=int(p.get('flags', 0)), table_begin=table_begin, table_end=table_end,
            subindex_group=subindex_group, subindex_groups=subindex_groups, row_begin=row_begin, row_end=row_end)
        self.dim, self.L, self.pb, self.top_k = d, Lt, pb, top_k
        self.h = C.c_void_p()
        rc = self.L_.dpf_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise DpfError(rc, self.L_.dpf_last_error(None).decode())
        self._check(self.L_.dpf_set_hash_family(self.h, _dp(fns), _dp(_f64(p.get("b"))), _ip(_i32(p.get("w"))),
                                                _ip(chain_fn)))
        if pb > 0:
            part_fns = _f64(p["part_fns"])
            assert part_fns.shape == (Lt, pb, 32), part_fns.shape
            self._check(self.L_.dpf_set_partitioner(self.h, _dp(part_fns), _dp(_f64(p.get("part_b"))),
                                                    _ip(_i32(p.get("part_w")))))
        if p.get("sampling_index") is not None:
            self._check(self.L_.dpf_set_sampling_index(self.h, _ip(_i32(p["sampling_index"]))))

    def _check(self, rc):
        if rc != 0:
            raise DpfError(rc, self.L_.dpf_last_error(self.h).decode())

    def save(self, path):
        """Writes the fitted index (configuration, hash family, hashes, resident rows) to one file."""
        self._check(self.L_.dpf_save(self.h, os.fsencode(path)))

    @classmethod
    def load(cls, path, device=-1, **shard):
        """A handle equal to the one `save` was called on; device < 0 keeps the saved ordinal.  With shard keywords
        (row_begin, row_end, table_begin, table_end, subindex_group, subindex_groups) only that shard of the saved
        index is brought up (dpf_load_shard)."""
        self = cls.__new__(cls)
        self.L_ = lib()
        self.h = C.c_void_p()
        if shard:
            sh = DpfConfig(row_begin=shard.get("row_begin", 0), row_end=shard.get("row_end", -1),
                           table_begin=shard.get("table_begin", 0), table_end=shard.get("table_end", 0),
                           subindex_group=shard.get("subindex_group", 0), subindex_groups=shard.get("subindex_groups", 0))
            rc = self.L_.dpf_load_shard(os.fsencode(path), device, C.byref(sh), C.byref(self.h))
        else:
            rc = self.L_.dpf_load(os.fsencode(path), device, C.byref(self.h))
        if rc != 0:
            raise DpfError(rc, self.L_.dpf_last_error(None).decode())
        self.cfg = DpfConfig()
        self._check(self.L_.dpf_get_config(self.h, C.byref(self.cfg)))
        self.np_dtype = np.dtype({v: k for k, v in DTYPE.items()}[self.cfg.dtype])
        self.dim, self.L, self.pb, self.top_k = self.cfg.dim, self.cfg.num_tables, self.cfg.partition_bits, self.cfg.top_k
        return self

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L_.dpf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _rows(self, X):
        X = np.ascontiguousarray(X, dtype=self.np_dtype)
        assert X.ndim == 2 and X.shape[1] == self.dim, X.shape
        return X

    def fit(self, X):
        X = self._rows(X)
        self._check(self.L_.dpf_fit(self.h, X.ctypes.data, X.shape[0]))

    def fit_stream(self, chunks, total_rows=0):
        """Fits a base that arrives chunk by chunk (an iterable of [rows, dim] arrays): every chunk is hashed and the
        rows of this handle's shard kept as it passes; the forest is built once, after the last chunk."""
        if total_rows:
            self.set_option("expected_rows", total_rows)
        self.set_option("defer_build", 1)
        try:
            for X in chunks:
                self.fit(X)
        finally:
            self.set_option("defer_build", 0)
        self._check(self.L_.dpf_fit(self.h, None, 0))

    def hash_batch(self, Q):
        Q = self._rows(Q)
        nq = Q.shape[0]
        h = np.empty((nq, self.L), dtype=np.int32)
        p = np.empty((nq, self.L), dtype=np.int32)
        self._check(self.L_.dpf_hash_batch(self.h, Q.ctypes.data, nq, _ip(h), _ip(p)))
        return h, p

    def query_batch(self, qids, Q, steps=0, copy=True):
        """NewMultiThreadQueryBatch -> (offsets [nq+1] int64, ids int32).  copy=True: through dpf_query_batch (a block of
        the caller's own).  copy=False: dpf_query_batch_view - the ids are a view of the handle's pinned result buffer,
        valid until the handle's next call that returns Sets (the array keeps the handle alive, not the contents)."""
        Q = self._rows(Q)
        nq = Q.shape[0]
        qids = _i32(qids)
        offsets = np.zeros(nq + 1, dtype=np.int64)
        pids = C.POINTER(C.c_int32)()
        fn = self.L_.dpf_query_batch if copy else self.L_.dpf_query_batch_view
        self._check(fn(self.h, _ip(qids), Q.ctypes.data, nq, steps, offsets.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(pids)))
        total = int(offsets[nq])
        ids = np.ctypeslib.as_array(pids, shape=(max(total, 1),))[:total]
        if copy:
            ids = ids.copy()
            self.L_.dpf_free(pids)
        else:
            ids = _HandleView(ids, self)
        return offsets, ids

    def query_batch_resident(self, qids, Q, steps=0):
        """The Sets stay on the device; -> offsets [nq+1] (the Set sizes).  sets_fetch / sets_topk read them."""
        Q = self._rows(Q)
        nq = Q.shape[0]
        offsets = np.zeros(nq + 1, dtype=np.int64)
        self._check(self.L_.dpf_query_batch_resident(self.h, _ip(_i32(qids)), Q.ctypes.data, nq, steps,
                                                     offsets.ctypes.data_as(C.POINTER(C.c_int64))))
        self._resident_offsets = offsets
        return offsets

    def sets_fetch(self, q0, q1):
        """The sorted unique id lists of the queries [q0, q1) of the resident Sets, back to back."""
        off = self._resident_offsets
        out = np.empty(max(int(off[q1] - off[q0]), 1), dtype=np.int32)
        self._check(self.L_.dpf_sets_fetch(self.h, q0, q1, _ip(out)))
        return out[:int(off[q1] - off[q0])]

    def sets_topk(self, k=None):
        nq = len(self._resident_offsets) - 1
        k = k or self.top_k
        ids = np.empty((nq, k), dtype=np.int32)
        sc = np.empty((nq, k), dtype=np.float64)
        self._check(self.L_.dpf_sets_topk(self.h, k, _ip(ids), _dp(sc)))
        return ids, sc

    def query_topk(self, qids, Q, steps=0, k=None, out=None):
        """out = (ids [nq, k] int32, scores [nq, k] float64) to have the results written into given arrays
        (pinned ones, pinned_array(), are filled by DMA)."""
        Q = self._rows(Q)
        nq = Q.shape[0]
        k = k or self.top_k
        qids = _i32(qids)
        if out is not None:
            ids, sc = out
            assert ids.dtype == np.int32 and sc.dtype == np.float64 and ids.shape == (nq, k) and sc.shape == (nq, k)
            assert ids.flags.c_contiguous and sc.flags.c_contiguous
        else:
            ids = np.empty((nq, k), dtype=np.int32)
            sc = np.empty((nq, k), dtype=np.float64)
        self._check(self.L_.dpf_query_topk(self.h, _ip(qids), Q.ctypes.data, nq, steps, k, _ip(ids), _dp(sc)))
        return ids, sc

    def query_topk_device(self, d_qids, d_Q, nq, steps, k, d_ids, d_scores, stream=None):
        """All arguments are device addresses (ints).  Blocking: the call waits on `stream` at least once
        per batch and may grow workspaces; the final rerank may still be running on `stream` on return."""
        self._check(self.L_.dpf_query_topk_device(self.h, d_qids, d_Q, nq, steps, k, d_ids, d_scores, stream))

    def query_topk_device_static(self, d_qids, d_Q, nq, steps, k, d_ids, d_scores, stream=None):
        """query_topk_device through the entry point that allocates nothing (fixed scratch, limited batch)."""
        self._check(self.L_.dpf_query_topk_device_static(self.h, d_qids, d_Q, nq, steps, k, d_ids, d_scores, stream))

    def merge_topk_device(self, in_ids, in_scores, parts, nq, k, d_ids, d_scores, stream=None):
        self._check(self.L_.dpf_merge_topk_device(self.h, in_ids, in_scores, parts, nq, k, d_ids, d_scores, stream))

    def merge_topk_packed_device(self, blocks, part_bytes, parts, nq, k, d_ids, d_scores, stream=None):
        self._check(self.L_.dpf_merge_topk_packed_device(self.h, blocks, part_bytes, parts, nq, k, d_ids, d_scores, stream))

    def rerank(self, Q, offsets, cand, k=None):
        Q = self._rows(Q)
        nq = Q.shape[0]
        k = k or self.top_k
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        cand = _i32(cand)
        ids = np.empty((nq, k), dtype=np.int32)
        sc = np.empty((nq, k), dtype=np.float64)
        self._check(self.L_.dpf_rerank(self.h, Q.ctypes.data, nq, offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                                       _ip(cand), k, _ip(ids), _dp(sc)))
        return ids, sc

    def subindex_distribution(self):
        out = np.zeros(1 << self.pb, dtype=np.float64)
        self._check(self.L_.dpf_subindex_distribution(self.h, _dp(out)))
        return out

    def snapshot(self):
        """The flattened forest as host arrays: (roots, dnodes, leaf_off, ids)."""
        sz = np.zeros(4, dtype=np.int64)
        self._check(self.L_.dpf_snapshot_sizes(self.h, sz.ctypes.data_as(C.POINTER(C.c_int64))))
        roots = np.zeros(sz[0], dtype=np.uint32)
        dnodes = np.zeros(sz[1], dtype=np.uint32)
        leaf_off = np.zeros(sz[2], dtype=np.uint32)
        ids = np.zeros(sz[3], dtype=np.int32)
        self._check(self.L_.dpf_snapshot_copy(self.h, roots.ctypes.data, dnodes.ctypes.data, leaf_off.ctypes.data,
                                              ids.ctypes.data))
        return roots, dnodes, leaf_off, ids

    def set_option(self, name, value):
        """Tuning / test switches of this handle (dpf_set_option); none changes a result."""
        self._check(self.L_.dpf_set_option(self.h, OPTION[name], int(value)))

    def set_probe_mode(self, mode):
        """"flips" (RandomDrawTreeMap.java:742-797, default) or "own" (the keyed walk, RandomDrawTreeMap.java:630-675)."""
        self._check(self.L_.dpf_set_probe_mode(self.h, {"flips": 0, "own": 1}[mode]))

    def set_profiling(self, on):
        self._check(self.L_.dpf_set_profiling(self.h, 1 if on else 0))

    def stats(self):
        st = DpfStats()
        self._check(self.L_.dpf_get_stats(self.h, C.byref(st)))
        return {n: getattr(st, n) for n, _ in DpfStats._fields_}

    def last_incidences(self, nq, list_cap=0):
        """(bucket_counts [num_leaves] uint32, ids_visited [nq] uint64, buckets_reached [nq] int32, bucket_lists
        [nq, list_cap] uint32) of the last bucket-major top-k batch of nq queries."""
        nl = self.stats()["num_leaves"]
        bc = np.zeros(nl, dtype=np.uint32)
        raw = np.zeros(nq, dtype=np.uint64)
        br = np.zeros(nq, dtype=np.int32)
        bl = np.zeros((nq, max(list_cap, 1)), dtype=np.uint32)
        self._check(self.L_.dpf_last_incidences(self.h, bc.ctypes.data, raw.ctypes.data, br.ctypes.data,
                                                bl.ctypes.data if list_cap > 0 else None, list_cap))
        return bc, raw, br, bl

    def last_forward_counts(self, nq):
        """(forwarded [nq], kept [nq]) of the last bucket-major top-k batch."""
        f = np.zeros(nq, dtype=np.int32)
        kpt = np.zeros(nq, dtype=np.int32)
        self._check(self.L_.dpf_last_forward_counts(self.h, f.ctypes.data, kpt.ctypes.data))
        return f, kpt

    def last_rerank_profile(self):
        ms = C.c_double()
        by = C.c_int64()
        self._check(self.L_.dpf_last_rerank_profile(self.h, C.byref(ms), C.byref(by)))
        return ms.value, by.value
