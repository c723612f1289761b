"""MI355X parity at the benchmark's own sizes: the workloads bench.py times (BASELINE.json `configs`),
built by bench.workload itself, against the CPU oracle fitted on the same full base.

The oracle answers a seeded sample of the batch's queries (it needs seconds per hundred queries at
these sizes); the HIP path answers the whole batch through the C ABI the way the benchmark calls it
(bucket-major scan for the top-k), and the sampled rows of that answer are compared.  Reference
behaviour checked: DensevectorRDFInit.scala:335-360 (candidate Sets), :472-507 (top-k order),
RandomDrawTreeMap.java:742-797 (probes, multi-step sub-indexes), :1719-1768 (split / tag quirk count).
"""
import time

import numpy as np
import pytest

import bench
from oracle import oracle as O
from similaritysearchbyrdf_amd._lib import DpfIndex
from tests.common import assert_same_sets, assert_topk_equal

pytestmark = pytest.mark.gpu


def _fit_both(name, n, nq, dim=None):
    params, X, Q, pick, steps = bench.workload(name, n, nq, dim)
    t0 = time.time()
    orc = O.OracleIndex(params)
    orc.fit(X.astype(np.float64), threads=16)
    t1 = time.time()
    gpu = DpfIndex(params, metric="l2", top_k=10)
    gpu.fit(X)
    print("%s: oracle fit %.1f s, device fit %.2f s" % (name, t1 - t0, time.time() - t1), flush=True)
    st = gpu.stats()
    assert st["num_vectors"] == n == orc.size
    assert st["misflag_events"] == orc.misflag_events
    assert st["num_leaves"] == sum(orc.table_stats(t)[1] for t in range(orc.L))
    assert st["max_leaf"] == max(orc.table_stats(t)[2] for t in range(orc.L))
    return params, X, Q, pick, steps, orc, gpu


def _check_batch(params, X, Q, pick, steps, orc, gpu, sample, set_steps, dot_too=True):
    nq = Q.shape[0]
    idx = np.sort(np.random.default_rng(2024).choice(nq, size=min(sample, nq), replace=False))
    Qs, ps = np.ascontiguousarray(Q[idx]), np.ascontiguousarray(pick[idx]).astype(np.int32)
    Qs64 = Qs.astype(np.float64)
    want_sets = {}
    for s in set_steps:
        want_sets[s] = orc.query_batch(ps, Qs64, steps=s, threads=16, mode=1)
        assert_same_sets(gpu.query_batch(ps, Qs, steps=s), want_sets[s])
    assert steps in want_sets
    off, ids = want_sets[steps]
    assert off[-1] > 0
    # the whole batch the way the benchmark runs it (bucket-major scan + exact rerank), sampled rows checked
    got_ids, got_sc = gpu.query_topk(pick.astype(np.int32), Q, steps=steps, k=10)
    st = gpu.stats()
    assert st["last_path"] == 1, "the benchmark's top-k path is the bucket-major scan"
    want_ids, want_sc = orc.topk(Qs64, off, ids, k=10, metric="l2", threads=16)
    assert_topk_equal(got_ids[idx], got_sc[idx], want_ids, want_sc)
    # every query of the batch: ascending distances, no repeated id, padding only at the tail
    valid = got_ids >= 0
    assert np.all(valid[:, :-1] >= valid[:, 1:])
    sc = np.where(valid, got_sc, np.inf)
    assert np.all(np.diff(sc, axis=1) >= 0)
    srt = np.sort(np.where(valid, got_ids, -np.arange(1, 11)[None, :]), axis=1)
    assert np.all(np.diff(srt, axis=1) != 0)
    # a second batch through the same handle (list sizes carried over) gives the same answer
    again_ids, again_sc = gpu.query_topk(pick.astype(np.int32), Q, steps=steps, k=10)
    assert np.array_equal(again_ids, got_ids) and np.array_equal(again_sc, got_sc, equal_nan=True)
    if dot_too:
        gdot = DpfIndex(params, metric="dot", top_k=10)
        gdot.fit(X)
        d_ids, d_sc = gdot.query_topk(ps, Qs, steps=steps, k=10)
        w_ids, w_sc = orc.topk(Qs64, off, ids, k=10, metric="dot", threads=16)
        assert_topk_equal(d_ids, d_sc, w_ids, w_sc)
        gdot.close()
    return st


def test_c3_full_forest_1m_base_10k_queries():
    # the configuration the headline metric is quoted on: 100-d, 1M base / 10k queries, 30 trees,
    # 3 partition bits, bucket overflow 500; candidate Sets at 0..3 search steps, top-10 at 1 step
    params, X, Q, pick, steps, orc, gpu = _fit_both("c3", 1_000_000, 10_000)
    assert steps == 1 and orc.L == 30
    _check_batch(params, X, Q, pick, steps, orc, gpu, sample=256, set_steps=(0, 1, 2, 3))
    assert gpu.stats()["short_hash_events"] == orc.short_hash_events


def test_c2_single_tree_1m_base_10k_queries():
    # one tree, 0-step search, all 10 000 queries: the oracle on a sample, set properties on every query
    params, X, Q, pick, steps, orc, gpu = _fit_both("c2", 1_000_000, 10_000)
    assert steps == 0 and orc.L == 1
    _check_batch(params, X, Q, pick, steps, orc, gpu, sample=512, set_steps=(0, 1, 3))
    o0, i0 = gpu.query_batch(pick.astype(np.int32), Q, steps=0)
    o3, i3 = gpu.query_batch(pick.astype(np.int32), Q, steps=3)
    assert o0.shape == (10_001,) and o0[-1] > 0
    # sorted unique lists: the only descents are at list starts
    for off, ids in ((o0, i0), (o3, i3)):
        desc = np.flatnonzero(np.diff(ids) <= 0) + 1
        assert np.isin(desc, off).all()
    # more steps only add sub-indexes: every 0-step set is inside the 3-step set of the same query
    q_of0 = np.repeat(np.arange(10_000, dtype=np.int64), np.diff(o0))
    q_of3 = np.repeat(np.arange(10_000, dtype=np.int64), np.diff(o3))
    assert np.isin(q_of0 * (1 << 32) + i0, q_of3 * (1 << 32) + i3).all()


def test_c4_sift_shaped_1m_base_10k_queries():
    # 128-d integer-valued 0..255 rows (SIFT-shaped), forest as c3
    params, X, Q, pick, steps, orc, gpu = _fit_both("c4", 1_000_000, 10_000)
    assert X.shape[1] == 128 and float(X.max()) == 255.0
    st = _check_batch(params, X, Q, pick, steps, orc, gpu, sample=128, set_steps=(0, 1))
    # integer-valued rows and queries inside one window of 256 values: scanned in int8 (exact integer products)
    assert st["last_scan_int8"] == 1 and gpu.stats()["base8_bytes"] == 1_000_000 * 128


def test_c5_share_768d_unit_gaussian():
    # one GPU's eighth of the 10M / 100k configuration: 1.25M unit-normalised 768-d rows, 12.5k queries
    params, X, Q, pick, steps, orc, gpu = _fit_both("c5", 1_250_000, 12_500)
    assert X.shape[1] == 768
    _check_batch(params, X, Q, pick, steps, orc, gpu, sample=96, set_steps=(1,), dot_too=False)


def test_c1_readme_simple_test_at_size():
    # BASELINE configs[0] = the reference's README "simple test" (README.md:32-43) at its stated size: 20 000 100-d
    # vectors, the default forest (tableNum 10 x permutationNum 3, 3 partition bits, bucket overflow 500), two query
    # vectors with components uniform in [0, 1) and ids 0 and 1 through NewMultiThreadQueryBatch
    # (DensevectorRDFInit.scala:335-360; README: 0 steps, here every step count) and the descending-inner-product
    # top-10 of DensevectorRDFInit.scala:472-507.
    rec = bench.c1_record(type("A", (), {})(), 10)
    assert rec["oracle_check"] is True, rec
    for st_ in ("0", "1", "2", "3"):
        assert rec["steps"][st_]["same_sets_as_oracle"] and rec["steps"][st_]["same_top10_as_oracle"], rec
    # more steps only add sub-indexes
    for q in range(2):
        sizes = [rec["steps"][str(s)]["set_sizes"][q] for s in range(4)]
        assert sizes == sorted(sizes), sizes
    # the same flow through the mirrored objects of the reference (text file, LSHServer globals, Set results)
    import os
    import tempfile
    from similaritysearchbyrdf_amd import synthetic
    from similaritysearchbyrdf_amd.host import Config, DenseVector, DensevectorRDFInit, LSH, LSHServer
    X, _, _ = synthetic.make_clustered(20_000, 100, 2)
    Q, _ = synthetic.fresh_queries("uniform", X, 2, seed=5)
    from tests.test_host_mirror import conf, _write_dense_file, _oracle_for_mirror
    c = conf()                                            # the reference's TestSettings defaults: 10 x 3 trees, overflow 500
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "glove.like.20k.DenseVector.txt")
        _write_dense_file(path, X.astype(np.float64), ids=np.arange(20_000))
        LSHServer.lshEngine = None
        LSHServer.isUseDense = True
        LSHServer.lshEngine = LSH(c)
        assert DensevectorRDFInit.newMultiThreadFit(path, c) == 20_000
    results = DensevectorRDFInit.NewMultiThreadQueryBatch([0, 1], [DenseVector(0, Q[0].astype(np.float64)), DenseVector(1, Q[1].astype(np.float64))], 0)
    orc = _oracle_for_mirror(c, 30, 3, 500)
    orc.fit(X.astype(np.float64))
    off, ids = orc.query_batch([0, 1], Q.astype(np.float64), steps=0)
    assert results == [set(ids[off[i]:off[i + 1]].tolist()) for i in range(2)]
    DensevectorRDFInit.clearAndClose()
