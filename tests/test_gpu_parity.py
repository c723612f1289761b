"""MI355X parity: the HIP path, called through the C ABI, against the CPU oracle."""
import numpy as np
import pytest

from oracle import oracle as O
from similaritysearchbyrdf_amd._lib import DpfIndex, DpfError, DPF_ENOTFIT
from tests.common import make_problem, assert_same_sets, assert_topk_equal

pytestmark = pytest.mark.gpu


def both(params, X, dtype=np.float32, **kw):
    orc = O.OracleIndex(params)
    orc.fit(np.asarray(X, dtype=np.float64), threads=8)
    gpu = DpfIndex(params, dtype=dtype, **kw)
    gpu.fit(X)
    return orc, gpu


@pytest.mark.parametrize("family,dtype,toi", [
    ("angle", np.float32, "original"), ("angle", np.float64, "original"),
    ("pStable", np.float32, "original"), ("pStable", np.float64, "original"),
    ("angle", np.float32, "sampling"), ("angle", np.float32, "continueBitsCount"),
    ("angle", np.float32, "angleNewMethod"), ("pStable", np.float32, "sampling"),
])
def test_hash_and_partition_bit_exact(family, dtype, toi):
    params, X, Q, _ = make_problem(n=3000, nq=64, family=family, dtype=dtype, type_of_index=toi)
    orc = O.OracleIndex(params)
    gpu = DpfIndex(params, dtype=dtype)
    h, p = gpu.hash_batch(X)
    X64 = X.astype(np.float64)
    want = orc.hashes(X64[:500])
    assert np.array_equal(h[:500], want)
    for i in range(0, 500, 7):
        for t in range(orc.L):
            assert p[i, t] == orc.partition(t, want[i, t])
    # every hash of the whole set through the fit path
    orc.fit(X64)
    for t in range(orc.L):
        oh, op = orc.fitted_hashes(t)
        assert np.array_equal(h[:, t], oh)
        assert np.array_equal(p[:, t], op)


@pytest.mark.parametrize("family,dtype,dim,fsize", [
    ("angle", np.float32, 100, 100), ("angle", np.float64, 100, 100), ("pStable", np.float32, 100, 100),
    ("pStable", np.float64, 33, 70), ("angle", np.float32, 3, 40), ("angle", np.float32, 130, 200),
    ("angle", np.float32, 768, 768), ("pStable", np.float32, 257, 129), ("angle", np.float64, 31, 33),
])
def test_matrix_core_projection_equals_exact_kernel_and_oracle(family, dtype, dim, fsize):
    # the fp64 MFMA contraction with its guard band + exact re-evaluation against the exact vector-ALU kernel
    # (every row) and the oracle (a sample): hashes and sub-index ids identical
    kw = dict(table_num=3, permutation_num=2) if family == "angle" else dict(table_num=4, permutation_num=1)
    params, X, Q, _ = make_problem(n=5000, nq=8, dim=dim, family=family, dtype=dtype, family_size=fsize, **kw)
    X[7] = 0                                            # an all-zero row: every dot product is exactly 0
    X[11] = X[10]
    mfma = DpfIndex(params, dtype=dtype)
    mfma.set_option("projection_min_rows", 1)
    exact = DpfIndex(params, dtype=dtype)
    exact.set_option("projection_min_rows", 0)
    hm, pm = mfma.hash_batch(X)
    he, pe = exact.hash_batch(X)
    assert np.array_equal(hm, he) and np.array_equal(pm, pe)
    # (the all-zero row is undecided for the angle family: sign(0); a pStable bucket index of b / w is decided)
    assert mfma.stats()["hash_undecided_rows"] >= (1 if family == "angle" else 0) and exact.stats()["hash_undecided_rows"] == 0
    orc = O.OracleIndex(params)
    want = orc.hashes(X[:200].astype(np.float64))
    assert np.array_equal(hm[:200], want)
    # odd batch sizes (partial tiles)
    for n in (1, 15, 129, 1000):
        h2, p2 = mfma.hash_batch(X[:n])
        assert np.array_equal(h2, he[:n]) and np.array_equal(p2, pe[:n])


@pytest.mark.parametrize("family", ["angle", "pStable"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_matrix_core_projection_on_decision_boundaries(family, dtype):
    # vectors built so that one dot product per row sits on a decision boundary up to rounding noise: orthogonal to
    # a pool function (angle: the sign is decided by the last bits of the left-to-right sum) or landing on an
    # integer bucket boundary (pStable).  The guard band must catch every such row and the exact kernel decide it.
    dim = 64
    params, X, Q, _ = make_problem(n=4000, nq=8, dim=dim, family=family, dtype=np.float64, family_size=96, pstable_w=3,
                                   **(dict(table_num=3, permutation_num=1)))
    fns = np.asarray(params["fns"], dtype=np.float64)
    rng = np.random.default_rng(3)
    X = X.astype(np.float64)
    for i in range(X.shape[0]):
        a = fns[rng.integers(0, fns.shape[0])]
        if family == "angle":
            target = 0.0
        else:
            f = int(np.flatnonzero((fns == a).all(axis=1))[0])
            target = float(rng.integers(-3, 4)) * float(params["w"][f]) - float(params["b"][f])
        X[i] = X[i] + (target - a @ X[i]) / (a @ a) * a
    X = np.ascontiguousarray(X, dtype=dtype)            # float32 rounding moves the rows off the boundary a little: both kinds occur
    mfma = DpfIndex(params, dtype=dtype)
    mfma.set_option("projection_min_rows", 1)
    hm, pm = mfma.hash_batch(X)
    orc = O.OracleIndex(params)
    want = orc.hashes(X.astype(np.float64))
    assert np.array_equal(hm, want)
    if dtype == np.float64:
        assert mfma.stats()["hash_undecided_rows"] > 100     # the boundary rows were recognised as such


def test_fit_through_matrix_core_projection_builds_the_same_forest():
    params, X, Q, pick = make_problem(n=30000, nq=64, bucket_overflow=40, table_num=3, permutation_num=2)
    a = DpfIndex(params, metric="l2")
    a.set_option("projection_min_rows", 1)
    b = DpfIndex(params, metric="l2")
    b.set_option("projection_min_rows", 0)
    a.fit(X)
    b.fit(X)
    for x, y in zip(a.snapshot(), b.snapshot()):
        assert np.array_equal(x, y)
    ai, as_ = a.query_topk(pick, Q, steps=1, k=10)
    bi, bs = b.query_topk(pick, Q, steps=1, k=10)
    assert np.array_equal(ai, bi) and np.array_equal(as_, bs, equal_nan=True)


def test_hash_exact_zero_dot_products():
    # sign(0) = 0 (AngleHashFamily.scala:184): integer data makes exact zeros likely
    rng = np.random.default_rng(5)
    dim = 16
    fns = rng.integers(-2, 3, size=(32, dim)).astype(np.float64)
    params = dict(family="angle", fns=fns, chain_fn=np.arange(32, dtype=np.int32).reshape(1, 32),
                  part_fns=rng.integers(-1, 2, size=(1, 3, 32)).astype(np.float64), partition_bits=3)
    X = rng.integers(-2, 3, size=(4000, dim)).astype(np.float32)
    orc = O.OracleIndex(params)
    gpu = DpfIndex(params)
    h, p = gpu.hash_batch(X)
    want = orc.hashes(X.astype(np.float64))
    assert np.array_equal(h, want)
    viamfma = DpfIndex(params)
    viamfma.set_option("projection_min_rows", 1)         # exact zeros through the matrix cores: all undecided, all re-evaluated
    h2, p2 = viamfma.hash_batch(X)
    assert np.array_equal(h2, want) and np.array_equal(p2, p)
    assert ((X.astype(np.float64) @ fns.T) == 0).any()
    for i in range(0, 4000, 13):
        assert p[i, 0] == orc.partition(0, want[i, 0])


CASES = [
    # n, B, dir, bb, pb, family
    (2000, 500, 32, 28, 3, "angle"),
    (2000, 4, 32, 28, 3, "angle"),
    (6000, 16, 32, 28, 2, "angle"),
    (6000, 16, 128, 28, 3, "angle"),
    (6000, 16, 64, 30, 3, "angle"),
    (6000, 16, 32, 30, 0, "angle"),
    (6000, 8, 32, 28, 3, "pStable"),
    (20000, 50, 32, 28, 3, "angle"),
]


@pytest.mark.parametrize("n,B,dns,bb,pb,family", CASES)
def test_candidate_sets_bit_exact(n, B, dns, bb, pb, family):
    params, X, Q, pick = make_problem(n=n, nq=48, family=family, bucket_overflow=B, dir_node_size=dns,
                                      bucket_bits=bb, partition_bits=pb, n_clusters=20, pstable_w=2)
    orc, gpu = both(params, X)
    st = gpu.stats()
    leaves = sum(orc.table_stats(t)[1] for t in range(orc.L))
    max_leaf = max(orc.table_stats(t)[2] for t in range(orc.L))
    assert st["num_leaves"] == leaves
    assert st["max_leaf"] == max_leaf
    assert st["misflag_events"] == orc.misflag_events
    # query ids on both sides of the boxed-Integer cache boundary (RDT:982), some equal to indexed ids
    qids = np.array([0, 5, 127, 128, 129, 1000, -1, -128, -129] + list(pick[9:48]), dtype=np.int32)
    Qm = Q.copy()
    Qm[:9] = X[[0, 5, 127, 128, 129, 1000, 3, 4, 6]]    # exact copies: own bucket is reachable
    for steps in range(0, pb + 1):
        want = orc.query_batch(qids, Qm.astype(np.float64), steps=steps)
        got = gpu.query_batch(qids, Qm, steps=steps)
        assert_same_sets(got, want)
        if steps == 0:
            assert want[0][-1] > 0
    st = gpu.stats()
    assert st["last_unique_candidates"] == int(want[0][-1])
    assert st["last_raw_candidates"] >= st["last_unique_candidates"]


@pytest.mark.parametrize("mode", ["compat", "always", "never"])
def test_self_exclusion_modes(mode):
    params, X, Q, _ = make_problem(n=1500, nq=8, exclude_self=mode, bucket_overflow=30)
    orc, gpu = both(params, X)
    qids = np.array([3, 100, 127, 128, 500, 1499, -5, 7], dtype=np.int32)
    Qm = X[[3, 100, 127, 128, 500, 1499, 5, 7]]
    want = orc.query_batch(qids, Qm.astype(np.float64), steps=1)
    got = gpu.query_batch(qids, Qm, steps=1)
    assert_same_sets(got, want)


def test_thread_count_independence_and_idempotence():
    # TestSingleRDFSuite.scala:36-61: multi-thread result == single-thread result
    params, X, Q, pick = make_problem(n=5000, nq=32, bucket_overflow=40, table_num=3, permutation_num=2)
    orc, gpu = both(params, X)
    w1 = orc.query_batch(pick, Q.astype(np.float64), steps=1, threads=1)
    w5 = orc.query_batch(pick, Q.astype(np.float64), steps=1, threads=5)
    w7 = orc.query_batch(pick, Q.astype(np.float64), steps=1, threads=7, mode=1)
    assert_same_sets(w1, w5)
    assert_same_sets(w1, w7)
    g1 = gpu.query_batch(pick, Q, steps=1)
    g2 = gpu.query_batch(pick, Q, steps=1)
    assert_same_sets(g1, w1)
    assert_same_sets(g2, w1)


@pytest.mark.parametrize("metric", ["dot", "l2"])
@pytest.mark.parametrize("dtype,dim", [(np.float32, 100), (np.float64, 100), (np.float32, 50), (np.float32, 128),
                                        (np.float32, 300), (np.float32, 768), (np.float64, 33)])
def test_topk_matches_oracle(metric, dtype, dim):
    params, X, Q, pick = make_problem(n=4000, nq=40, dim=dim, dtype=dtype, bucket_overflow=60, table_num=3,
                                      permutation_num=1, n_clusters=10)
    orc, gpu = both(params, X, dtype=dtype, metric=metric, top_k=10)
    Q64 = Q.astype(np.float64)
    off, ids = orc.query_batch(pick, Q64, steps=1)
    want_ids, want_sc = orc.topk(Q64, off, ids, k=10, metric=metric)
    got_ids, got_sc = gpu.query_topk(pick, Q, steps=1, k=10)
    assert_topk_equal(got_ids, got_sc, want_ids, want_sc)
    # the rerank stage alone, over the oracle's candidate lists
    r_ids, r_sc = gpu.rerank(Q, off, ids, k=10)
    assert_topk_equal(r_ids, r_sc, want_ids, want_sc)
    for k in (1, 3, 64):
        wi, ws = orc.topk(Q64, off, ids, k=k, metric=metric)
        gi, gs = gpu.rerank(Q, off, ids, k=k)
        assert_topk_equal(gi, gs, wi, ws)


def test_topk_with_duplicate_vectors_breaks_ties_by_id():
    params, X, Q, pick = make_problem(n=3000, nq=20, bucket_overflow=100, n_clusters=5)
    X[1000:2000] = X[0:1000]            # every vector twice: exact score ties
    orc, gpu = both(params, X, metric="l2")
    Q64 = Q.astype(np.float64)
    off, ids = orc.query_batch(pick, Q64, steps=2)
    want = orc.topk(Q64, off, ids, k=10, metric="l2")
    got = gpu.query_topk(pick, Q, steps=2, k=10)
    assert np.array_equal(got[0], want[0])


def test_row_shards_and_merge():
    import torch
    params, X, Q, pick = make_problem(n=6000, nq=24, bucket_overflow=50, table_num=2, permutation_num=2)
    orc = O.OracleIndex(params)
    orc.fit(X.astype(np.float64))
    Q64 = Q.astype(np.float64)
    off, ids = orc.query_batch(pick, Q64, steps=1)
    want = orc.topk(Q64, off, ids, k=10, metric="l2")
    bounds = [(0, 1500), (1500, 3000), (3000, 4500), (4500, -1)]
    parts_ids, parts_sc = [], []
    for lo, hi in bounds:
        g = DpfIndex(params, metric="l2", row_begin=lo, row_end=hi)
        g.fit(X)
        so, si = g.query_batch(pick, Q, steps=1)
        hi_eff = 6000 if hi < 0 else hi
        for q in range(24):
            w = ids[off[q]:off[q + 1]]
            assert np.array_equal(si[so[q]:so[q + 1]], w[(w >= lo) & (w < hi_eff)])
        pi, ps = g.query_topk(pick, Q, steps=1, k=10)
        parts_ids.append(pi)
        parts_sc.append(ps)
    dev = torch.device("cuda:0")
    t_ids = torch.from_numpy(np.stack(parts_ids)).to(dev)
    t_sc = torch.from_numpy(np.stack(parts_sc)).to(dev)
    o_ids = torch.empty((24, 10), dtype=torch.int32, device=dev)
    o_sc = torch.empty((24, 10), dtype=torch.float64, device=dev)
    g.merge_topk_device(t_ids.data_ptr(), t_sc.data_ptr(), 4, 24, 10, o_ids.data_ptr(), o_sc.data_ptr())
    torch.cuda.synchronize()
    assert_topk_equal(o_ids.cpu().numpy(), o_sc.cpu().numpy(), want[0], want[1])
    # the same lists as packed blocks (scores, then ids, per part), the form one all-gather delivers;
    # k = 7 leaves a block that needs padding to 8 bytes (24 * 7 * 12 = 2016 + 0; 23 queries: 1932 -> 1936)
    from similaritysearchbyrdf_amd.dist import packed_bytes, packed_views
    for nq, k in ((24, 10), (23, 7)):
        pb = packed_bytes(nq, k)
        blocks = torch.zeros((4, pb), dtype=torch.uint8, device=dev)
        b_sc, b_ids = packed_views(blocks, nq, k)
        b_sc.copy_(t_sc[:, :nq, :k])
        b_ids.copy_(t_ids[:, :nq, :k])
        p_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
        p_sc = torch.empty((nq, k), dtype=torch.float64, device=dev)
        g.merge_topk_packed_device(blocks.data_ptr(), pb, 4, nq, k, p_ids.data_ptr(), p_sc.data_ptr())
        c_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
        c_sc = torch.empty((nq, k), dtype=torch.float64, device=dev)
        s_ids, s_sc = t_ids[:, :nq, :k].contiguous(), t_sc[:, :nq, :k].contiguous()
        g.merge_topk_device(s_ids.data_ptr(), s_sc.data_ptr(), 4, nq, k, c_ids.data_ptr(), c_sc.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(p_ids, c_ids) and np.array_equal(p_sc.cpu().numpy(), c_sc.cpu().numpy(), equal_nan=True)


def test_incremental_fit_equals_one_fit():
    params, X, Q, pick = make_problem(n=5000, nq=16, bucket_overflow=12)
    orc = O.OracleIndex(params)
    orc.fit(X[:2000].astype(np.float64))
    orc.fit(X[2000:].astype(np.float64))
    gpu = DpfIndex(params)
    gpu.fit(X[:2000])
    gpu.fit(X[2000:3000])
    gpu.fit(X[3000:])
    one = DpfIndex(params)
    one.fit(X)
    want = orc.query_batch(pick, Q.astype(np.float64), steps=1)
    assert_same_sets(gpu.query_batch(pick, Q, steps=1), want)
    assert_same_sets(one.query_batch(pick, Q, steps=1), want)
    assert gpu.stats()["num_vectors"] == 5000


def test_device_entry_point_matches_host_entry_point():
    import torch
    params, X, Q, pick = make_problem(n=3000, nq=33, bucket_overflow=25)
    gpu = DpfIndex(params, metric="l2")
    gpu.fit(X)
    hi, hs = gpu.query_topk(pick, Q, steps=1, k=10)
    dev = torch.device("cuda:0")
    dq = torch.from_numpy(Q).to(dev)
    dqid = torch.from_numpy(pick.astype(np.int32)).to(dev)
    di = torch.empty((33, 10), dtype=torch.int32, device=dev)
    ds = torch.empty((33, 10), dtype=torch.float64, device=dev)
    s = torch.cuda.Stream()
    gpu.set_profiling(True)
    with torch.cuda.stream(s):
        gpu.query_topk_device(dqid.data_ptr(), dq.data_ptr(), 33, 1, 10, di.data_ptr(), ds.data_ptr(), s.cuda_stream)
    s.synchronize()
    assert np.array_equal(di.cpu().numpy(), hi)
    assert np.array_equal(ds.cpu().numpy(), hs)
    ms, by = gpu.last_rerank_profile()
    assert ms > 0 and by > 0


def test_edge_cases():
    params, X, Q, pick = make_problem(n=300, nq=4)
    gpu = DpfIndex(params)
    off = np.zeros(5, dtype=np.int64)
    with pytest.raises(DpfError) as ei:
        gpu.query_batch(pick, Q, steps=0)
    assert ei.value.code == DPF_ENOTFIT          # "need to fit the data first", Dense:422-424
    gpu.fit(X)
    o, i = gpu.query_batch(np.zeros(0, dtype=np.int32), np.zeros((0, 100), dtype=np.float32))
    assert o.tolist() == [0] and i.size == 0
    # a query far from everything still returns a (possibly empty) sorted unique list
    far = np.full((1, 100), 1e6, dtype=np.float32)
    o, i = gpu.query_batch(np.array([9], dtype=np.int32), far, steps=3)
    assert np.all(np.diff(i) > 0)
    # empty candidate lists give -1 / NaN padded rows
    ids, sc = gpu.rerank(Q, np.zeros(5, dtype=np.int64), np.zeros(0, dtype=np.int32), k=5)
    assert (ids == -1).all() and np.isnan(sc).all()
    with pytest.raises(DpfError):
        DpfIndex(dict(params, dir_node_size=48))


def test_short_hash_is_counted_not_fatal():
    # RDT:753: 0 <= hash < 8 makes `new int[negative]` throw in the reference; here
    # such (query, table) pairs return nothing and are counted
    fns = np.array([[1.0], [-1.0]])
    chain = np.array([[1] * 32], dtype=np.int32)       # every bit 0 -> h = 0
    params = dict(family="angle", fns=fns, chain_fn=chain, part_fns=np.zeros((1, 2, 32)), partition_bits=2,
                  bucket_overflow=10)
    X = np.ones((50, 1), dtype=np.float32)
    orc, gpu = both(params, X)
    want = orc.query_batch(np.arange(3), np.ones((3, 1)), steps=1)
    got = gpu.query_batch(np.arange(3), np.ones((3, 1), dtype=np.float32), steps=1)
    assert_same_sets(got, want)
    assert want[0][-1] == 0
    assert gpu.stats()["short_hash_events"] == 3 == orc.short_hash_events


def test_full_size_properties():
    # 1M x 100 (the benchmark's base size): properties that need no oracle run
    params, X, Q, pick = make_problem(n=1_000_000, nq=512, table_num=1, permutation_num=1, n_clusters=1000,
                                      gaussian=True)
    gpu = DpfIndex(params, metric="l2")
    gpu.fit(X)
    st = gpu.stats()
    assert st["num_vectors"] == 1_000_000 and st["num_leaves"] > 1000 and st["max_leaf"] >= 500
    o0, i0 = gpu.query_batch(pick, Q, steps=0)
    o1, i1 = gpu.query_batch(pick, Q, steps=1)
    o3, i3 = gpu.query_batch(pick, Q, steps=3)
    for q in range(512):
        a, b, c = i0[o0[q]:o0[q + 1]], i1[o1[q]:o1[q + 1]], i3[o3[q]:o3[q + 1]]
        assert np.all(np.diff(a) > 0) and np.all(np.diff(b) > 0) and np.all(np.diff(c) > 0)
        assert np.isin(a, b).all() and np.isin(b, c).all()      # more steps only add sub-indexes
    ids, sc = gpu.query_topk(pick, Q, steps=1, k=10)
    for q in range(0, 512, 17):
        cand = i1[o1[q]:o1[q + 1]]
        d2 = ((X[cand].astype(np.float64) - Q[q].astype(np.float64)) ** 2).sum(axis=1)
        order = np.lexsort((cand, d2))[:10]
        assert np.array_equal(ids[q, :len(order)], cand[order])
        assert np.allclose(sc[q, :len(order)], d2[order], rtol=1e-9)
    # the oracle on a 1/5 sample of the same data agrees too
    orc = O.OracleIndex(params)
    orc.fit(X[:200_000].astype(np.float64))
    g2 = DpfIndex(params)
    g2.fit(X[:200_000])
    assert_same_sets(g2.query_batch(pick[:64], Q[:64], steps=1),
                     orc.query_batch(pick[:64], Q[:64].astype(np.float64), steps=1))
    assert g2.stats()["misflag_events"] == orc.misflag_events


def test_large_forest_batches_in_a_row():
    # the benchmark's forest shape (30 trees, 1-step search) at 300k rows: batches of changing size through
    # one handle (sizes carried from batch to batch, regrown once) give the gather path's answers
    params, X, Q, pick = make_problem(n=300_000, nq=6000, table_num=10, permutation_num=3, n_clusters=300)
    fast = DpfIndex(params, metric="l2")
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    fast.fit(X)
    slow.fit(X)
    seen = []
    for lo, hi in ((0, 4000), (4000, 4500), (0, 4000), (0, 6000), (3000, 6000)):
        fi, fs = fast.query_topk(pick[lo:hi], Q[lo:hi], steps=1, k=10)
        st = fast.stats()
        assert st["last_path"] == 1
        seen.append(st["last_sizes_carried"])
        si, ss = slow.query_topk(pick[lo:hi], Q[lo:hi], steps=1, k=10)
        assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True), (lo, hi, seen)
    assert seen[0] == 0 and seen[1] == 1 and seen[2] == 1 and seen[4] == 1, seen


@pytest.mark.parametrize("metric", ["l2", "dot"])
@pytest.mark.parametrize("dim", [100, 50, 128, 24, 200, 256, 300, 512, 768, 1000])
def test_bucket_major_rerank_equals_gather_rerank(metric, dim):
    # f32 handles of up to 1024 dimensions rerank bucket-major (fp16 filter with measured rounding
    # bounds + exact fp64 rerank of the survivors); the per-query gather kernel scores every
    # candidate in fp64.  Same ids, same scores, bit for bit.
    params, X, Q, pick = make_problem(n=20000, nq=300, dim=dim, bucket_overflow=80, table_num=3, permutation_num=2,
                                      n_clusters=12)
    fast = DpfIndex(params, metric=metric)
    slow = DpfIndex(dict(params, flags=2), metric=metric)          # DPF_FLAG_GATHER_ONLY
    fast.fit(X)
    slow.fit(X)
    qids = pick.astype(np.int32)
    qids[:5] = [0, 5, 127, 128, -1]
    for steps in (0, 1, 3):
        fi, fs = fast.query_topk(qids, Q, steps=steps, k=10)
        assert fast.stats()["last_path"] == 1
        si, ss = slow.query_topk(qids, Q, steps=steps, k=10)
        assert slow.stats()["last_path"] == 0
        assert np.array_equal(fi, si)
        assert np.array_equal(fs, ss, equal_nan=True)
    fi, fs = fast.query_topk(qids, Q, steps=1, k=64)
    si, ss = slow.query_topk(qids, Q, steps=1, k=64)
    assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True)
    st = fast.stats()
    assert st["last_pairs"] > 0 and 0 < st["last_forwarded"] <= st["last_raw_candidates"]


def test_bucket_major_overflow_falls_back_to_gather():
    params, X, Q, pick = make_problem(n=8000, nq=40, bucket_overflow=60, n_clusters=6)
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    slow.fit(X)
    want = slow.query_topk(pick, Q, steps=1, k=10)
    fast = DpfIndex(params, metric="l2")
    fast.set_option("emit_cap", 4)               # any query forwarding more than 4 candidates overflows
    fast.fit(X)
    got = fast.query_topk(pick, Q, steps=1, k=10)
    assert fast.stats()["last_path"] == 0
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True)


def test_bucket_major_with_row_shards():
    params, X, Q, pick = make_problem(n=9000, nq=50, bucket_overflow=50)
    full = DpfIndex(dict(params, flags=2), metric="l2")
    full.fit(X)
    fo, fi = full.query_batch(pick, Q, steps=1)
    for lo, hi in ((0, 3000), (3000, 9000)):
        g = DpfIndex(params, metric="l2", row_begin=lo, row_end=hi)
        g.fit(X)
        ids, sc = g.query_topk(pick, Q, steps=1, k=10)
        assert g.stats()["last_path"] == 1
        for q in range(50):
            c = fi[fo[q]:fo[q + 1]]
            c = c[(c >= lo) & (c < hi)]
            d2 = ((X[c].astype(np.float64) - Q[q].astype(np.float64)) ** 2).sum(axis=1)
            order = np.lexsort((c, d2))[:10]
            assert np.array_equal(ids[q, :len(order)], c[order])


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["l2", "dot"])
@pytest.mark.parametrize("kind", ["huge", "tiny", "mixed", "near_duplicates", "zeros"])
@pytest.mark.parametrize("dim", [100, 700])
def test_fp16_filter_is_exact_on_hostile_values(metric, kind, dim):
    # The bucket-major filter works on fp16 copies whose rounding distance is measured per row, so
    # values outside the fp16 range (clamped), below its normal range (flushed) or differing only
    # in bits fp16 drops must still give the gather path's answer bit for bit.
    # (700 dimensions: the 16-row-tile form of the scan)
    params, X, Q, pick = make_problem(n=6000, nq=120, dim=dim, bucket_overflow=40, table_num=2, permutation_num=2,
                                      n_clusters=5)
    rng = np.random.default_rng(7)
    if kind == "huge":
        X = X * np.float32(3.0e4); Q = Q * np.float32(3.0e4)            # many elements beyond 65504
    elif kind == "tiny":
        X = X * np.float32(1.0e-6); Q = Q * np.float32(1.0e-6)          # everything below 2^-14
    elif kind == "mixed":
        s = (10.0 ** rng.integers(-7, 6, size=(1, X.shape[1]))).astype(np.float32)
        X = X * s; Q = Q * s
    elif kind == "near_duplicates":
        base = X[:60]
        X = (base[rng.integers(0, 60, size=X.shape[0])] * (1 + 1e-4 * rng.standard_normal(X.shape))).astype(np.float32)
        Q = (X[pick] * (1 + 1e-5 * rng.standard_normal(Q.shape))).astype(np.float32)
    elif kind == "zeros":
        X[::7] = 0; Q[::5] = 0
    X = np.ascontiguousarray(X, dtype=np.float32); Q = np.ascontiguousarray(Q, dtype=np.float32)
    fast = DpfIndex(params, metric=metric)
    slow = DpfIndex(dict(params, flags=2), metric=metric)
    fast.fit(X)
    slow.fit(X)
    for k in (1, 10, 64):
        fi, fs = fast.query_topk(pick, Q, steps=2, k=k)
        si, ss = slow.query_topk(pick, Q, steps=2, k=k)
        assert np.array_equal(fi, si)
        assert np.array_equal(fs, ss, equal_nan=True)


@pytest.mark.gpu
def test_bucket_major_many_queries_per_bucket_and_repeated_batches():
    # a few buckets probed by hundreds of queries (several work units per tile), batches of
    # changing size through one handle (workspaces are reused and regrown)
    params, X, Q, pick = make_problem(n=5000, nq=1500, dim=64, bucket_overflow=100, table_num=2, permutation_num=1,
                                      n_clusters=3)
    fast = DpfIndex(params, metric="l2")
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    fast.fit(X)
    slow.fit(X)
    for nq in (1500, 7, 300, 1500):
        fi, fs = fast.query_topk(pick[:nq], Q[:nq], steps=1, k=10)
        assert fast.stats()["last_path"] == 1
        si, ss = slow.query_topk(pick[:nq], Q[:nq], steps=1, k=10)
        assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_list_sizes_carried_from_batch_to_batch(metric):
    # After a handle's first bucket-major batch the host no longer waits for the batch's list sizes: the
    # buffers keep the room earlier batches left, and a batch that outgrows them is abandoned on the
    # device and run again.  Same answers in all three modes, through growing and shrinking batches,
    # a change of search steps and a refit.
    params, X, Q, pick = make_problem(n=6000, nq=2000, dim=64, bucket_overflow=80, table_num=3, permutation_num=1,
                                      n_clusters=5)
    fast = DpfIndex(params, metric=metric)
    slow = DpfIndex(dict(params, flags=2), metric=metric)
    fast.fit(X[:4000])
    slow.fit(X[:4000])
    seen = []
    for nq, steps in ((5, 0), (5, 0), (2000, 1), (2000, 1), (300, 2), (40, 0), (2000, 3)):
        fi, fs = fast.query_topk(pick[:nq], Q[:nq], steps=steps, k=10)
        st = fast.stats()
        assert st["last_path"] == 1
        seen.append(st["last_sizes_carried"])
        si, ss = slow.query_topk(pick[:nq], Q[:nq], steps=steps, k=10)
        assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True), (nq, steps, seen)
    assert seen[0] == 0 and seen[1] == 1 and seen[2] == 2 and seen[3] == 1 and seen[4] == 1 and seen[5] == 1, seen
    fast.fit(X[4000:])
    slow.fit(X[4000:])
    for nq, steps in ((2000, 1), (100, 1)):
        fi, fs = fast.query_topk(pick[:nq], Q[:nq], steps=steps, k=10)
        assert fast.stats()["last_path"] == 1
        si, ss = slow.query_topk(pick[:nq], Q[:nq], steps=steps, k=10)
        assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("bins", [0, 1, 2])
def test_carried_sizes_with_a_truncated_bucket_list(bins):
    # a batch that did not wait for its sizes and then truncates a query's bucket list is run again
    # with the second walk - when the lists are filled from the remembered buckets (list_bins = 0); built from the binned
    # (bucket, query) pairs they do not depend on the remembered buckets, which then only feed the seed sample
    params, X, Q, pick = make_problem(n=8000, nq=60, bucket_overflow=60, n_clusters=6, table_num=2, permutation_num=1)
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    slow.fit(X)
    fast = DpfIndex(params, metric="l2")
    fast.set_option("list_bins", bins)
    fast.fit(X)
    seen = []
    for i, steps in enumerate((1, 1, 1, 1, 0)):
        if i == 2:
            fast.set_option("bucket_list_cap", 3)
        got = fast.query_topk(pick, Q, steps=steps, k=10)
        st = fast.stats()
        assert st["last_path"] == 1
        seen.append(st["last_sizes_carried"])
        want = slow.query_topk(pick, Q, steps=steps, k=10)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True), (steps, seen)
    assert seen[:4] == ([0, 1, 1, 1] if bins else [0, 1, 2, 0]), seen


@pytest.mark.gpu
@pytest.mark.parametrize("bins", [1, 2])
def test_binned_lists_when_the_queries_move(bins):
    # The bins of buckets and their regions are cut from the batch that was sized on the host.  A later batch of many
    # more, and other, queries outgrows the regions: it is run again with its sizes read back, which cuts the bins
    # anew, and the batches after it are carried again.  Same answers as the query-major path throughout.
    params, X, Q, pick = make_problem(n=20000, nq=1500, bucket_overflow=40, n_clusters=40, table_num=3, permutation_num=2)
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    slow.fit(X)
    fast = DpfIndex(params, metric="l2")
    fast.set_option("list_bins", bins)
    fast.fit(X)
    seen = []
    for lo, hi in ((0, 20), (0, 20), (20, 1500), (20, 1500), (700, 760), (0, 1500)):
        got = fast.query_topk(pick[lo:hi], Q[lo:hi], steps=1, k=10)
        st = fast.stats()
        assert st["last_path"] == 1
        seen.append(st["last_sizes_carried"])
        want = slow.query_topk(pick[lo:hi], Q[lo:hi], steps=1, k=10)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True), ((lo, hi), seen)
    assert seen == [0, 1, 2, 1, 1, 1], seen


@pytest.mark.gpu
def test_binned_lists_with_query_numbers_past_16_bits():
    # the list entries are staged in LDS as 16-bit query numbers; a batch of more than 65536 queries is not staged
    params, X, Q, pick = make_problem(n=6000, nq=200, dim=32, bucket_overflow=40, n_clusters=20, table_num=2, permutation_num=1)
    reps = 66000 // 200 + 1
    Qb = np.ascontiguousarray(np.tile(Q, (reps, 1)))
    pb = np.ascontiguousarray(np.tile(pick, reps))
    fast = DpfIndex(params, metric="l2")
    fast.fit(X)
    first = fast.query_topk(pb, Qb, steps=1, k=5)
    again = fast.query_topk(pb, Qb, steps=1, k=5)
    assert fast.stats()["last_path"] == 1 and fast.stats()["last_sizes_carried"] == 1
    assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1], equal_nan=True)
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    slow.fit(X)
    want = slow.query_topk(pick, Q, steps=1, k=5)
    for r in (0, reps - 1):
        assert np.array_equal(again[0][r * 200:(r + 1) * 200], want[0])
        assert np.array_equal(again[1][r * 200:(r + 1) * 200], want[1], equal_nan=True)


@pytest.mark.gpu
def test_bucket_major_truncated_bucket_lists_walk_again():
    # the first probe pass remembers each query's buckets up to a cap; past it the lists are
    # filled by a second walk.  Same answer either way.
    params, X, Q, pick = make_problem(n=8000, nq=60, bucket_overflow=60, n_clusters=6)
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    slow.fit(X)
    want = slow.query_topk(pick, Q, steps=1, k=10)
    fast = DpfIndex(params, metric="l2")
    fast.set_option("bucket_list_cap", 3)
    fast.fit(X)
    got = fast.query_topk(pick, Q, steps=1, k=10)
    assert fast.stats()["last_path"] == 1
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["l2", "dot"])
def test_table_shards_and_merge(metric):
    # every handle holds everything but probes only its tables; per-handle candidate sets are the
    # union over those tables, and the merged top-k (an id found by several handles taken once)
    # is the unsharded answer
    import torch
    params, X, Q, pick = make_problem(n=6000, nq=40, bucket_overflow=50, table_num=3, permutation_num=2)
    orc = O.OracleIndex(params)
    orc.fit(X.astype(np.float64))
    Q64 = Q.astype(np.float64)
    off, ids = orc.query_batch(pick, Q64, steps=1)
    want = orc.topk(Q64, off, ids, k=10, metric=metric)
    bounds = [(0, 2), (2, 3), (3, 6)]
    parts_ids, parts_sc = [], []
    for lo, hi in bounds:
        g = DpfIndex(params, metric=metric, table_begin=lo, table_end=hi)
        g.fit(X)
        so, si = g.query_batch(pick, Q, steps=1)
        for q in range(40):
            w = np.unique(np.concatenate([orc.get_similar(t, int(pick[q]), Q64[q], 1) for t in range(lo, hi)]))
            assert np.array_equal(si[so[q]:so[q + 1]], w)
        pi, ps = g.query_topk(pick, Q, steps=1, k=10)
        parts_ids.append(pi)
        parts_sc.append(ps)
    dev = torch.device("cuda:0")
    t_ids = torch.from_numpy(np.stack(parts_ids)).to(dev)
    t_sc = torch.from_numpy(np.stack(parts_sc)).to(dev)
    o_ids = torch.empty((40, 10), dtype=torch.int32, device=dev)
    o_sc = torch.empty((40, 10), dtype=torch.float64, device=dev)
    g.merge_topk_device(t_ids.data_ptr(), t_sc.data_ptr(), 3, 40, 10, o_ids.data_ptr(), o_sc.data_ptr())
    torch.cuda.synchronize()
    assert_topk_equal(o_ids.cpu().numpy(), o_sc.cpu().numpy(), want[0], want[1])
    with pytest.raises(Exception):
        DpfIndex(params, metric=metric, table_begin=4, table_end=9)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,family", [(np.float32, "angle"), (np.float64, "pStable")])
def test_save_and_load_round_trip(tmp_path, dtype, family):
    # a loaded handle answers exactly as the saved one (sets, top-k, stats), accepts further fits
    # like the original, and damaged files are refused
    kw = dict(table_num=2, permutation_num=2) if family == "angle" else dict(table_num=3, permutation_num=1)
    params, X, Q, pick = make_problem(n=5000, nq=40, bucket_overflow=40, family=family, **kw)
    a = DpfIndex(params, metric="l2", dtype=dtype)
    a.fit(X[:3000])
    path = str(tmp_path / "index.dpf")
    a.save(path)
    b = DpfIndex.load(path)
    assert (b.dim, b.L, b.pb) == (a.dim, a.L, a.pb) and b.np_dtype == a.np_dtype
    for ix in (a, b):
        ix.fit(X[3000:])                                   # incremental fit on both
    ao, ai = a.query_batch(pick, Q, steps=1)
    bo, bi = b.query_batch(pick, Q, steps=1)
    assert np.array_equal(ao, bo) and np.array_equal(ai, bi)
    at, as_ = a.query_topk(pick, Q, steps=1, k=10)
    bt, bs = b.query_topk(pick, Q, steps=1, k=10)
    assert np.array_equal(at, bt) and np.array_equal(as_, bs, equal_nan=True)
    sa, sb = a.stats(), b.stats()
    for key in ("num_vectors", "num_dnodes", "num_leaves", "max_leaf", "misflag_events"):
        assert sa[key] == sb[key]
    blob = open(path, "rb").read()
    bad = str(tmp_path / "bad.dpf")
    open(bad, "wb").write(blob[:len(blob) // 2])
    with pytest.raises(Exception):
        DpfIndex.load(bad)
    open(bad, "wb").write(b"garbage!" + blob[8:])
    with pytest.raises(Exception):
        DpfIndex.load(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("family,toi,pb,dns,bb", [("pStable", "original", 3, 32, 28), ("angle", "sampling", 2, 128, 28),
                                                 ("angle", "continueBitsCount", 3, 64, 30), ("pStable", "original", 0, 32, 28)])
def test_bucket_major_across_families_and_geometries(family, toi, pb, dns, bb):
    # the bucket-major path sits behind every hash family / index transform / trie geometry the
    # probe supports: same answer as the gather path, and as the oracle
    kw = dict(table_num=3, permutation_num=1) if family == "pStable" else dict(table_num=2, permutation_num=2)
    params, X, Q, pick = make_problem(n=7000, nq=80, bucket_overflow=40, family=family, type_of_index=toi,
                                      partition_bits=pb, dir_node_size=dns, bucket_bits=bb, **kw)
    fast = DpfIndex(params, metric="l2")
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    fast.fit(X)
    slow.fit(X)
    orc = O.OracleIndex(params)
    orc.fit(X.astype(np.float64))
    Q64 = Q.astype(np.float64)
    for steps in range(0, pb + 1):
        fi, fs = fast.query_topk(pick, Q, steps=steps, k=10)
        assert fast.stats()["last_path"] == 1
        si, ss = slow.query_topk(pick, Q, steps=steps, k=10)
        assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True)
        off, ids = orc.query_batch(pick, Q64, steps=steps)
        want = orc.topk(Q64, off, ids, k=10, metric="l2")
        assert_topk_equal(fi, fs, want[0], want[1])


def _same_snapshot(a, b):
    for x, y, name in zip(a.snapshot(), b.snapshot(), ("roots", "dnodes", "leaf_off", "ids")):
        assert x.shape == y.shape, name
        assert np.array_equal(x, y), name
    sa, sb = a.stats(), b.stats()
    for key in ("num_dnodes", "num_leaves", "max_leaf", "misflag_events", "snapshot_bytes"):
        assert sa[key] == sb[key], key
    assert np.array_equal(a.subindex_distribution(), b.subindex_distribution())


@pytest.mark.gpu
@pytest.mark.parametrize("n,B,dns,bb,pb,family", [
    (2000, 500, 32, 28, 3, "angle"),
    (3000, 1, 32, 28, 3, "angle"),
    (3000, 2, 32, 28, 1, "angle"),
    (3000, 4, 32, 28, 3, "angle"),
    (6000, 7, 128, 28, 2, "angle"),
    (6000, 5, 64, 30, 3, "angle"),
    (6000, 3, 32, 30, 0, "angle"),
    (5000, 4, 32, 28, 3, "pStable"),
    (20000, 40, 32, 28, 3, "angle"),
])
def test_device_forest_equals_host_forest(n, B, dns, bb, pb, family):
    # The forest derived on the device (radix-sort rounds over all tables) against the host builder,
    # which tests/test_forest_builder.py pins to the oracle's one-at-a-time insertion: identical
    # arrays, so identical walks.  Small overflow thresholds exercise the "born full" corner of
    # RDT:1719-1768 (a child bucket created with more than BUCKET_OVERFLOW ids).
    kw = dict(table_num=3, permutation_num=1, pstable_w=2) if family == "pStable" else dict(table_num=2, permutation_num=2)
    params, X, Q, pick = make_problem(n=n, nq=32, family=family, bucket_overflow=B, dir_node_size=dns, bucket_bits=bb,
                                      partition_bits=pb, n_clusters=20, **kw)
    dev = DpfIndex(params, metric="l2")
    host = DpfIndex(dict(params, flags=4), metric="l2")       # DPF_FLAG_HOST_FOREST
    dev.fit(X)
    host.fit(X)
    _same_snapshot(dev, host)
    orc = O.OracleIndex(params)
    orc.fit(X.astype(np.float64))
    assert dev.stats()["misflag_events"] == orc.misflag_events
    for steps in (0, pb):
        assert_same_sets(dev.query_batch(pick, Q, steps=steps), orc.query_batch(pick, Q.astype(np.float64), steps=steps))


@pytest.mark.gpu
def test_device_forest_incremental_fit_and_row_shard():
    params, X, Q, pick = make_problem(n=9000, nq=32, bucket_overflow=6, n_clusters=10)
    whole = DpfIndex(dict(params, flags=4), metric="l2")
    whole.fit(X)
    parts = DpfIndex(params, metric="l2")
    parts.fit(X[:2500])
    parts.fit(X[2500:2501])
    parts.fit(X[2501:])
    _same_snapshot(parts, whole)
    a = DpfIndex(params, metric="l2", row_begin=3000, row_end=7000)
    b = DpfIndex(dict(params, flags=4), metric="l2", row_begin=3000, row_end=7000)
    a.fit(X)
    b.fit(X)
    _same_snapshot(a, b)
    ai, asc = a.query_topk(pick, Q, steps=1, k=10)
    bi, bsc = b.query_topk(pick, Q, steps=1, k=10)
    assert np.array_equal(ai, bi) and np.array_equal(asc, bsc, equal_nan=True)


@pytest.mark.gpu
def test_device_forest_degenerate_inputs():
    # all rows equal: one hash per table, every id lands in one level-0 chain of n ids
    params, X, Q, pick = make_problem(n=1500, nq=8, bucket_overflow=4)
    X[:] = X[0]
    dev = DpfIndex(params, metric="l2")
    host = DpfIndex(dict(params, flags=4), metric="l2")
    dev.fit(X)
    host.fit(X)
    _same_snapshot(dev, host)
    assert dev.stats()["max_leaf"] == 1500
    # a single row
    one = DpfIndex(params, metric="l2")
    ref = DpfIndex(dict(params, flags=4), metric="l2")
    one.fit(X[:1])
    ref.fit(X[:1])
    _same_snapshot(one, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("metric,with_qids", [("l2", True), ("l2", False), ("dot", True)])
def test_overflowed_queries_take_the_gather_path_alone(metric, with_qids):
    # a forward-list cap that only a few queries exceed: those are answered by the gather path as a
    # batch of their own, the others by the scan, and the batch equals the gather-only handle's
    params, X, Q, pick = make_problem(n=12000, nq=400, bucket_overflow=60, n_clusters=8)
    slow = DpfIndex(dict(params, flags=2), metric=metric)
    slow.fit(X)
    qids = pick if with_qids else None
    want = slow.query_topk(qids, Q, steps=1, k=10)
    seen_partial = False
    for cap in (512, 256, 128, 96, 64, 48, 32, 24, 16):
        fast = DpfIndex(params, metric=metric)
        fast.set_option("emit_cap", cap)
        fast.fit(X)
        got = fast.query_topk(qids, Q, steps=1, k=10)
        st = fast.stats()
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True), cap
        if st["last_path"] == 1 and st["last_overflowed"] > 0:
            seen_partial = True
            # and once more on the same handle: the per-batch buffers are back in place
            again = fast.query_topk(qids, Q, steps=1, k=10)
            assert np.array_equal(again[0], want[0]) and np.array_equal(again[1], want[1], equal_nan=True)
            assert fast.stats()["last_raw_candidates"] == st["last_raw_candidates"]
        fast.close()
    assert seen_partial


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(48))
def test_random_configurations(seed, dims=(8, 33, 64, 100, 130, 256)):
    # a seeded draw over the configuration space: hash family, index transform, trie geometry, overflow
    # threshold, dimension, metric, k, search steps; scan path == gather path (bit for bit) == oracle
    rng = np.random.RandomState(1000 + seed)
    family = ["angle", "pStable"][rng.randint(2)]
    toi = ["original", "sampling", "continueBitsCount"][rng.randint(3)] if family == "angle" else "original"
    dns, bb = [(32, 28), (128, 28), (64, 30), (32, 30)][rng.randint(4)]
    pb = int(rng.randint(0, 4))
    B = int([3, 17, 60, 500][rng.randint(4)])
    dim = int(dims[rng.randint(6)])
    metric = ["l2", "dot"][rng.randint(2)]
    k = int([1, 5, 10, 33][rng.randint(4)])
    n, nq = int(rng.randint(1500, 6000)), int(rng.randint(5, 120))
    kw = dict(table_num=int(rng.randint(1, 4)), permutation_num=1, pstable_w=3) if family == "pStable" else \
        dict(table_num=int(rng.randint(1, 3)), permutation_num=int(rng.randint(1, 4)))
    params, X, Q, pick = make_problem(n=n, nq=nq, dim=dim, family=family, type_of_index=toi, partition_bits=pb,
                                      dir_node_size=dns, bucket_bits=bb, bucket_overflow=B, n_clusters=int(rng.randint(2, 40)),
                                      seed=50 + seed, **kw)
    # (seeds from 24 on) the values themselves: as drawn; integers inside one window of 256 values (the int8 scan, where
    # the metric is l2 and the rows are short enough); floats with a common offset (fp16 copies around the mean row)
    kind = ["plain", "integers", "offset"][seed % 3] if seed >= 24 else "plain"
    if kind == "integers":
        lo = float([0, -128, 40, -700][rng.randint(4)])
        scale = 255.0 / max(float(X.max() - X.min()), 1e-6)
        X = np.clip(np.rint((X - X.min()) * scale * 0.9 + lo + 10.0), lo, lo + 255.0).astype(np.float32)
        Q = np.clip(np.rint(X[pick] + rng.randint(-3, 4, size=Q.shape)), lo, lo + 255.0).astype(np.float32)
    elif kind == "offset":
        shift = np.float32([30.0, -250.0, 2000.0][rng.randint(3)])
        X = np.ascontiguousarray(X + shift, dtype=np.float32)
        Q = np.ascontiguousarray(Q + shift, dtype=np.float32)
    fast = DpfIndex(params, metric=metric, top_k=k)
    slow = DpfIndex(dict(params, flags=2), metric=metric, top_k=k)
    fast.fit(X)
    slow.fit(X)
    orc = O.OracleIndex(params)
    orc.fit(X.astype(np.float64))
    Q64 = Q.astype(np.float64)
    steps = int(rng.randint(0, pb + 1))
    fi, fs = fast.query_topk(pick, Q, steps=steps, k=k)
    if kind == "plain":
        assert fast.stats()["last_path"] == 1
    if kind == "integers" and metric == "l2" and dim <= 256 and fast.stats()["last_path"] == 1:
        assert fast.stats()["last_scan_int8"] == 1
    si, ss = slow.query_topk(pick, Q, steps=steps, k=k)
    assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True)
    off, ids = orc.query_batch(pick, Q64, steps=steps)
    assert_same_sets(fast.query_batch(pick, Q, steps=steps), (off, ids))
    want = orc.topk(Q64, off, ids, k=k, metric=metric)
    assert_topk_equal(fi, fs, want[0], want[1])
    # the handle's later batches do not wait for their list sizes and build the bucket lists from binned pairs: other
    # queries, other step counts, same answers as the gather path
    for rep in range(3):
        steps2 = int(rng.randint(0, pb + 1))
        sel = rng.permutation(len(pick))[:max(1, len(pick) - 7 * rep)]
        qi, qq = np.ascontiguousarray(pick[sel]), np.ascontiguousarray(Q[sel])
        fi2, fs2 = fast.query_topk(qi, qq, steps=steps2, k=k)
        si2, ss2 = slow.query_topk(qi, qq, steps=steps2, k=k)
        assert np.array_equal(fi2, si2) and np.array_equal(fs2, ss2, equal_nan=True), (rep, steps2, fast.stats()["last_sizes_carried"])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(100, 118))
def test_random_configurations_long_rows(seed):
    # the same draw over rows of 300 to 1000 dimensions: the scan's forms of 16, 24 and 32 steps (one wave per SIMD with
    # the chunk's query rows staged in LDS; 16-row tiles beyond 768 dimensions)
    test_random_configurations(seed, dims=(300, 400, 512, 640, 768, 1000))


@pytest.mark.parametrize("groups", [2, 3, 8])
def test_subindex_shards_and_merge(groups):
    # every handle holds everything but probes, in every table, only the sub-indexes of its group (the reference's own
    # partitioning unit, Partitioner.scala:27-64); per handle the sets equal the oracle restricted the same way, their
    # union is the whole set, and the merged top-k is the unsharded answer
    import torch
    params, X, Q, pick = make_problem(n=7000, nq=40, bucket_overflow=40, table_num=2, permutation_num=2)
    orc = O.OracleIndex(params)
    orc.fit(X.astype(np.float64))
    Q64 = Q.astype(np.float64)
    off, ids = orc.query_batch(pick, Q64, steps=1)
    want = orc.topk(Q64, off, ids, k=10, metric="l2")
    parts_ids, parts_sc, union = [], [], [set() for _ in range(40)]
    for g in range(groups):
        h = DpfIndex(params, metric="l2", subindex_group=g, subindex_groups=groups)
        h.fit(X)
        orc.set_subindex_shard(g, groups)
        for steps in (0, 1, 3):
            assert_same_sets(h.query_batch(pick, Q, steps=steps), orc.query_batch(pick, Q64, steps=steps))
        so, si = h.query_batch(pick, Q, steps=1)
        for q in range(40):
            union[q] |= set(si[so[q]:so[q + 1]].tolist())
        pi, ps = h.query_topk(pick, Q, steps=1, k=10)
        assert h.stats()["last_path"] == 1
        parts_ids.append(pi)
        parts_sc.append(ps)
    orc.set_subindex_shard(0, 1)
    for q in range(40):
        assert sorted(union[q]) == ids[off[q]:off[q + 1]].tolist()
    dev = torch.device("cuda:0")
    t_ids = torch.from_numpy(np.stack(parts_ids)).to(dev)
    t_sc = torch.from_numpy(np.stack(parts_sc)).to(dev)
    o_ids = torch.empty((40, 10), dtype=torch.int32, device=dev)
    o_sc = torch.empty((40, 10), dtype=torch.float64, device=dev)
    h.merge_topk_device(t_ids.data_ptr(), t_sc.data_ptr(), groups, 40, 10, o_ids.data_ptr(), o_sc.data_ptr())
    torch.cuda.synchronize()
    assert_topk_equal(o_ids.cpu().numpy(), o_sc.cpu().numpy(), want[0], want[1])
    with pytest.raises(Exception):
        DpfIndex(params, subindex_group=9, subindex_groups=9)      # more groups than the 8 sub-indexes


@pytest.mark.parametrize("family,pb", [("angle", 3), ("pStable", 2), ("angle", 0)])
def test_keyed_walk_probes_the_hash_itself(family, pb):
    # RandomDrawTreeMap.getSimilarWithStepWise(key, steps) (RandomDrawTreeMap.java:630-675), the walk
    # SparsevectorRDFInit's keyed batch query uses (SparsevectorRDFInit.scala:407): the query's own bucket in every
    # sub-index within `steps`, no bit flips, no probe-count quirk
    kw = dict(table_num=3, permutation_num=1, pstable_w=2) if family == "pStable" else dict(table_num=2, permutation_num=2)
    params, X, Q, pick = make_problem(n=6000, nq=48, family=family, bucket_overflow=30, partition_bits=pb, n_clusters=10, **kw)
    orc, gpu = both(params, X, metric="l2")
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    slow.fit(X)
    orc.set_probe_own(True)
    for h in (gpu, slow):
        h.set_probe_mode("own")
    qids = pick.astype(np.int32)
    qids[:4] = [0, 5, 127, 128]
    Qm = Q.copy()
    Qm[:4] = X[[0, 5, 127, 128]]
    Q64 = Qm.astype(np.float64)
    for steps in range(0, pb + 1):
        want = orc.query_batch(qids, Q64, steps=steps)
        assert_same_sets(gpu.query_batch(qids, Qm, steps=steps), want)
        wi, ws = orc.topk(Q64, want[0], want[1], k=10, metric="l2")
        fi, fs = gpu.query_topk(qids, Qm, steps=steps, k=10)
        si, ss = slow.query_topk(qids, Qm, steps=steps, k=10)
        assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True)
        assert_topk_equal(fi, fs, wi, ws)
    assert want[0][-1] > 0
    # and back: the default walk again
    orc.set_probe_own(False)
    gpu.set_probe_mode("flips")
    assert_same_sets(gpu.query_batch(qids, Qm, steps=1), orc.query_batch(qids, Q64, steps=1))


def test_load_shards_of_a_saved_index(tmp_path):
    # one file saved by an unsharded handle; row, table and sub-index shards brought up from it (memory-mapped, only
    # the shard's rows copied) answer as handles fitted with the same shard do
    params, X, Q, pick = make_problem(n=9000, nq=40, bucket_overflow=40, table_num=3, permutation_num=2)
    whole = DpfIndex(params, metric="l2")
    whole.fit(X)
    path = str(tmp_path / "whole.dpf")
    whole.save(path)
    shards = [dict(row_begin=0, row_end=3000), dict(row_begin=3000, row_end=-1), dict(table_begin=2, table_end=5),
              dict(subindex_group=1, subindex_groups=3), dict(row_begin=4000, row_end=8000, table_begin=0, table_end=3)]
    for sh in shards:
        a = DpfIndex.load(path, **sh)
        b = DpfIndex(params, metric="l2", **sh)
        b.fit(X)
        assert a.stats()["base_bytes"] == b.stats()["base_bytes"]
        for x, y in zip(a.snapshot(), b.snapshot()):
            assert np.array_equal(x, y)
        ao, ai = a.query_batch(pick, Q, steps=1)
        bo, bi = b.query_batch(pick, Q, steps=1)
        assert np.array_equal(ao, bo) and np.array_equal(ai, bi)
        at, as_ = a.query_topk(pick, Q, steps=1, k=10)
        bt, bs = b.query_topk(pick, Q, steps=1, k=10)
        assert np.array_equal(at, bt) and np.array_equal(as_, bs, equal_nan=True)
    # a shard file cannot serve rows it does not hold
    part = DpfIndex.load(path, row_begin=0, row_end=3000)
    ppath = str(tmp_path / "part.dpf")
    part.save(ppath)
    again = DpfIndex.load(ppath)
    assert again.stats()["base_bytes"] == part.stats()["base_bytes"]
    with pytest.raises(Exception):
        DpfIndex.load(ppath, row_begin=2000, row_end=5000)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["clustered", "tail_heavy", "tail_only", "huge_tail"])
@pytest.mark.parametrize("dim,row_bytes", [(33, 64), (34, 64), (66, 128), (68, 128), (97, 192), (100, 192), (102, 192),
                                          (130, 256), (136, 256), (260, 512), (264, 512), (520, 1024), (770, 1536),
                                          (776, 1536), (104, 256), (40, 128)])
def test_cut_rows_with_tails_filter_exactly(kind, dim, row_bytes):
    # l2 handles whose dimension sticks out over a row length of the scan by at most 8 (and a sixteenth) scan rows
    # cut at that length and add the tail's share for the scores that pass: the same answers as the whole-row filter
    # (DPF_FLAG_FULL_FILTER) and as the gather path, bit for bit - also when the tail carries most of the distance,
    # all of it, or values beyond the fp16 range.  (104 and 40 dimensions: too much sticks out, whole rows.)
    n, nq = (6000, 100) if dim <= 300 else (3000, 60)
    params, X, Q, pick = make_problem(n=n, nq=nq, dim=dim, bucket_overflow=40, table_num=2, permutation_num=2, n_clusters=6)
    cut = dim // 32 * 32
    rng = np.random.default_rng(dim)
    if kind == "tail_heavy":
        X[:, cut:] *= 40.0; Q[:, cut:] *= 40.0
    elif kind == "tail_only":                       # rows agree with one of six prototypes on the leading dimensions: only the tail tells them apart
        proto = rng.integers(0, 6, size=n)
        X[:, :cut] = X[:6][proto][:, :cut]
        Q[:, :cut] = X[pick][:, :cut]
    elif kind == "huge_tail":
        X[:, cut:] *= 3.0e4; Q[:, cut:] *= 3.0e4    # beyond 65504: clamped in the fp16 tail, covered by the measured rounding distance
    X = np.ascontiguousarray(X, dtype=np.float32); Q = np.ascontiguousarray(Q, dtype=np.float32)
    fast = DpfIndex(params, metric="l2")
    full = DpfIndex(dict(params, flags=8), metric="l2")            # DPF_FLAG_FULL_FILTER
    slow = DpfIndex(dict(params, flags=2), metric="l2")            # DPF_FLAG_GATHER_ONLY
    for ix in (fast, full, slow):
        ix.fit(X)
    assert fast.stats()["base16_bytes"] == n * row_bytes
    whole = next(v for v in (1, 2, 3, 4, 8, 16, 24, 32) if (dim + 31) // 32 <= v)      # steps of 32 dimensions a whole row takes
    assert full.stats()["base16_bytes"] == n * whole * 64
    qids = pick.astype(np.int32)
    for steps, k in ((1, 10), (3, 1), (0, 64)):
        fi, fs = fast.query_topk(qids, Q, steps=steps, k=k)
        # (clamped values make the measured rounding distances huge: the filter forwards everything and the batch may
        # end on the gather path - still the same answer)
        assert fast.stats()["last_path"] == 1 or kind == "huge_tail"
        ui, us = full.query_topk(qids, Q, steps=steps, k=k)
        si, ss = slow.query_topk(qids, Q, steps=steps, k=k)
        assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True)
        assert np.array_equal(ui, si) and np.array_equal(us, ss, equal_nan=True)
    # the two filters use the same fp16 information (their rounding allowances differ a little): about the same number
    # of rows is forwarded - not the many more a filter blind to the tail would let through
    f_cut, f_whole = fast.stats()["last_forwarded"], full.stats()["last_forwarded"]
    assert f_cut <= 1.3 * f_whole + 2 * nq, (f_cut, f_whole)
    for ix in (fast, full, slow):
        ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dim,passes,early,first", [(100, 3, 40, 100), (100, 3, 300, 600), (100, 2, 0, 0), (100, 2, 0, 1000),
                                                    (768, 2, 0, 100), (768, 3, 5, 10), (768, 3, 500, 999), (768, 0, 40, 100)])
def test_scan_in_two_or_three_passes(dim, passes, early, first):
    # the work units of the scan run in slices with the thresholds refined in between; however they are cut (no
    # early slice, an empty first one, nearly everything before the last refinement), the answer is the gather path's
    params, X, Q, pick = make_problem(n=12000, nq=150, dim=dim, bucket_overflow=60, table_num=3, permutation_num=2, n_clusters=8)
    fast = DpfIndex(params, metric="l2")
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    fast.fit(X)
    slow.fit(X)
    fast.set_option("scan_passes", passes)
    fast.set_option("early_pass_permille", early)
    fast.set_option("first_pass_permille", first)
    si, ss = slow.query_topk(pick, Q, steps=1, k=10)
    for rep in range(3):                                   # sized on the host, then twice with carried sizes
        fi, fs = fast.query_topk(pick, Q, steps=1, k=10)
        st = fast.stats()
        assert st["last_path"] == 1 and st["last_sizes_carried"] == (0 if rep == 0 else 1)
        assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True)
    want = 1 + (1 if 0 < first < 1000 else 0) + (1 if (passes == 3 or (passes == 0 and dim >= 512)) and 0 < early < first else 0)
    assert fast.stats()["last_scan_launches"] <= want
    with pytest.raises(Exception):
        fast.set_option("scan_passes", 4)


def _integer_problem(n, nq, dim, lo, seed, spread=12.0, n_clusters=12, **kw):
    """Integer-valued rows and queries inside the window [lo, lo + 255] (SIFT-shaped when lo = 0)."""
    params, _, _, _ = make_problem(n=16, nq=2, dim=dim, **kw)
    rng = np.random.default_rng(seed)
    centres = rng.integers(lo + 20, lo + 236, size=(n_clusters, dim)).astype(np.float32)
    lab = rng.integers(0, n_clusters, size=n)
    X = np.clip(np.rint(centres[lab] + spread * rng.standard_normal((n, dim), dtype=np.float32)), lo, lo + 255).astype(np.float32)
    pick = rng.integers(0, n, size=nq)
    Q = np.clip(np.rint(X[pick] + 4.0 * rng.standard_normal((nq, dim), dtype=np.float32)), lo, lo + 255).astype(np.float32)
    return params, np.ascontiguousarray(X), np.ascontiguousarray(Q), pick.astype(np.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,lo", [(128, 0), (100, 0), (24, -128), (64, 1000), (65, 0), (200, -300), (256, 0), (192, 7)])
def test_int8_scan_equals_fp16_scan_and_gather(dim, lo):
    # l2 handles whose rows are all integers inside one window of 256 values keep an int8 copy; a batch of integer
    # queries inside the window is scanned with exact integer products (v_mfma_i32_16x16x64_i8): the answers are those of
    # the fp16 scan (DPF_FLAG_NO_INT8) and of the gather path, bit for bit; a batch with a query that does not fit the
    # window takes the fp16 scan
    n, nq = 9000, 160
    params, X, Q, pick = _integer_problem(n, nq, dim, lo, seed=dim + 7, bucket_overflow=50, table_num=2, permutation_num=2)
    X[5] = X[4]; X[6] = X[4]                                       # equal rows: ties by id
    X[0, 0] = lo; X[1, 0] = lo + 255                                # the window's ends are there
    i8 = DpfIndex(params, metric="l2")
    f16 = DpfIndex(dict(params, flags=16), metric="l2")            # DPF_FLAG_NO_INT8
    slow = DpfIndex(dict(params, flags=2), metric="l2")            # DPF_FLAG_GATHER_ONLY
    for ix in (i8, f16, slow):
        ix.fit(X)
    assert i8.stats()["base8_bytes"] == n * ((dim + 63) // 64) * 64
    assert f16.stats()["base8_bytes"] == 0 and slow.stats()["base8_bytes"] == 0
    for steps, k in ((1, 10), (3, 1), (0, 64), (1, 10)):
        ii, is_ = i8.query_topk(pick, Q, steps=steps, k=k)
        st = i8.stats()
        # (windows far from zero make the sign hashes nearly constant: a handful of huge buckets, and a 3-step batch
        # may overflow its forward lists and end on the gather path - still the same answer)
        assert (st["last_path"] == 1 and st["last_scan_int8"] == 1) or (abs(lo) > 255 and steps != 1)
        fi, fs = f16.query_topk(pick, Q, steps=steps, k=k)
        assert f16.stats()["last_scan_int8"] == 0
        si, ss = slow.query_topk(pick, Q, steps=steps, k=k)
        assert np.array_equal(ii, si) and np.array_equal(is_, ss, equal_nan=True)
        assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True)
    # the exact filter forwards no more than the fp16 one (whose thresholds carry rounding allowances), up to the half
    # unit the integer test gives away
    assert i8.stats()["last_forwarded"] <= 1.05 * f16.stats()["last_forwarded"] + nq
    # queries without ids (nothing excluded), and the no-threshold start (fewer candidates than k)
    ii, is_ = i8.query_topk(None, Q, steps=2, k=10)
    si, ss = slow.query_topk(None, Q, steps=2, k=10)
    assert i8.stats()["last_scan_int8"] == 1 and np.array_equal(ii, si) and np.array_equal(is_, ss, equal_nan=True)
    # one query off the integers or outside the window: those batches are scanned in fp16, same answers
    for bad in (lo + 0.5, lo + 256.0, lo - 1.0):
        Q2 = Q.copy()
        Q2[nq // 2, dim - 1] = bad
        ii, is_ = i8.query_topk(pick, Q2, steps=1, k=10)
        st = i8.stats()
        assert st["last_path"] == 1 and st["last_scan_int8"] == 0
        si, ss = slow.query_topk(pick, Q2, steps=1, k=10)
        assert np.array_equal(ii, si) and np.array_equal(is_, ss, equal_nan=True)
    ii, is_ = i8.query_topk(pick, Q, steps=1, k=10)               # and back
    assert i8.stats()["last_scan_int8"] == 1
    si, ss = slow.query_topk(pick, Q, steps=1, k=10)
    assert np.array_equal(ii, si) and np.array_equal(is_, ss, equal_nan=True)
    for ix in (i8, f16, slow):
        ix.close()


@pytest.mark.gpu
def test_int8_copy_only_where_the_rows_allow_it():
    n, nq, dim = 3000, 40, 96
    params, X, Q, pick = _integer_problem(n, nq, dim, 0, seed=3, bucket_overflow=50, table_num=2, permutation_num=1)
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    slow.fit(X)
    si, ss = slow.query_topk(pick, Q, steps=1, k=10)

    def check(Xv, want_i8, metric="l2"):
        ix = DpfIndex(params, metric=metric)
        ix.fit(Xv)
        assert (ix.stats()["base8_bytes"] > 0) == want_i8
        got = ix.query_topk(pick, Q, steps=1, k=10)
        assert ix.stats()["last_scan_int8"] == (1 if want_i8 else 0)
        ix.close()
        return got

    ii, is_ = check(X, True)
    assert np.array_equal(ii, si) and np.array_equal(is_, ss, equal_nan=True)
    check(X, False, metric="dot")                                  # the int8 form is an l2 filter
    Xh = X.copy(); Xh[17, 3] += 0.25                               # one value off the integers
    check(Xh, False)
    Xw = X.copy(); Xw[17, 3] = 256.0                               # 257 values wide
    check(Xw, False)
    # a second fit that breaks the window drops the copy, one that keeps it keeps it
    ix = DpfIndex(params, metric="l2")
    ix.fit(X[:2000])
    assert ix.stats()["base8_bytes"] == 2000 * 128
    ix.fit(X[2000:])
    assert ix.stats()["base8_bytes"] == n * 128
    ii, is_ = ix.query_topk(pick, Q, steps=1, k=10)
    assert ix.stats()["last_scan_int8"] == 1 and np.array_equal(ii, si) and np.array_equal(is_, ss, equal_nan=True)
    ix.fit(Xh[:50])
    assert ix.stats()["base8_bytes"] == 0
    ix.query_topk(pick, Q, steps=1, k=10)
    assert ix.stats()["last_scan_int8"] == 0
    ix.close()
    Xf = X + 3000.0                                                # beyond the integers fp16 holds exactly
    check(Xf, False)
    # more than 256 dimensions: no int8 rows
    p2, X2, Q2, pick2 = _integer_problem(2000, 20, 300, 0, seed=4, bucket_overflow=50, table_num=2, permutation_num=1)
    ix = DpfIndex(p2, metric="l2")
    ix.fit(X2)
    assert ix.stats()["base8_bytes"] == 0
    ix.close()
    slow.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dim", [33, 100, 128, 200, 300, 768, 1000])
def test_first_radii_from_fp16_copies_or_exact_seeds(dim):
    # l2 handles take each query's first radius from the fp16 copies of its seed rows (an upper bound of the k-th best
    # distance); DPF_OPT_EXACT_SEEDS scores the seeds exactly from the fp32 rows instead: same answers either way, and
    # about the same number of rows forwarded (the bound is loose only by the measured rounding distances)
    n, nq = (8000, 120) if dim <= 300 else (3000, 60)
    params, X, Q, pick = make_problem(n=n, nq=nq, dim=dim, bucket_overflow=50, table_num=2, permutation_num=2, n_clusters=8)
    half = DpfIndex(params, metric="l2")
    exact = DpfIndex(params, metric="l2")
    slow = DpfIndex(dict(params, flags=2), metric="l2")
    for ix in (half, exact, slow):
        ix.fit(X)
    exact.set_option("exact_seeds", 1)
    for steps, k in ((1, 10), (0, 1), (3, 64)):
        hi_, hs = half.query_topk(pick, Q, steps=steps, k=k)
        ei, es = exact.query_topk(pick, Q, steps=steps, k=k)
        si, ss = slow.query_topk(pick, Q, steps=steps, k=k)
        assert half.stats()["last_path"] == 1 and exact.stats()["last_path"] == 1
        assert np.array_equal(hi_, si) and np.array_equal(hs, ss, equal_nan=True)
        assert np.array_equal(ei, si) and np.array_equal(es, ss, equal_nan=True)
        f_half, f_exact = half.stats()["last_forwarded"], exact.stats()["last_forwarded"]
        assert f_half <= 1.2 * f_exact + 4 * nq, (f_half, f_exact)
    for ix in (half, exact, slow):
        ix.close()


@pytest.mark.gpu
def test_three_ways_to_take_the_sets():
    # NewMultiThreadQueryBatch's Sets (DensevectorRDFInit.scala:335-360): dpf_query_batch returns a block the caller owns;
    # dpf_query_batch_view the handle's own pinned buffer, reused by the next call (a copy taken before stays what it
    # was); dpf_query_batch_resident leaves them on the device for dpf_sets_fetch (any query range) and dpf_sets_topk
    # (the rerank of DensevectorRDFInit.scala:477-505), until another query or fit call drops them
    from similaritysearchbyrdf_amd._lib import DpfError, DPF_EINVAL
    params, X, Q, pick = make_problem(n=5000, nq=64, dim=48, bucket_overflow=40, table_num=2, permutation_num=2)
    orc, gpu = both(params, X)
    Q64 = Q.astype(np.float64)
    want0 = orc.query_batch(pick, Q64, steps=0, threads=2, mode=1)
    want1 = orc.query_batch(pick, Q64, steps=1, threads=2, mode=1)
    o1, v1 = gpu.query_batch(pick, Q, steps=0, copy=False)
    keep = v1.copy()
    at1 = v1.ctypes.data
    o2, v2 = gpu.query_batch(pick, Q, steps=1, copy=False)          # the next call's lists take the buffer over (or a grown one)
    assert o2[-1] >= o1[-1]
    assert_same_sets((o2, np.asarray(v2).copy()), want1)
    o3, v3 = gpu.query_batch(pick[:8], Q[:8], steps=0, copy=False)   # smaller lists fit the same buffer
    assert v3.ctypes.data == v2.ctypes.data and at1 != 0
    assert_same_sets((o3, np.asarray(v3).copy()), orc.query_batch(pick[:8], Q64[:8], steps=0, threads=2, mode=1))
    assert_same_sets((o1, keep), want0)
    o4, v4 = gpu.query_batch(pick, Q, steps=1)                       # the default: a block of the caller's own
    assert_same_sets((o4, v4), want1)
    assert v4.ctypes.data != v3.ctypes.data
    # resident Sets: sizes first, lists on demand, top-k without ever moving the lists
    off = gpu.query_batch_resident(pick, Q, steps=1)
    assert np.array_equal(off, want1[0])
    assert gpu.L_.dpf_sets_queries(gpu.h) == 64 and gpu.L_.dpf_sets_count(gpu.h, 3, 9) == off[9] - off[3]
    for q0, q1 in ((0, 64), (5, 6), (17, 40), (63, 64), (10, 10)):
        assert np.array_equal(gpu.sets_fetch(q0, q1), want1[1][off[q0]:off[q1]])
    wi, ws = orc.topk(Q64, want1[0], want1[1], k=10, metric="dot")
    ti, ts = gpu.sets_topk(10)
    assert_topk_equal(ti, ts, wi, ws)
    di, ds = gpu.query_topk(pick, Q, steps=1, k=10)                  # another query call: same answer, and the Sets are gone
    assert np.array_equal(di, ti) and np.array_equal(ds, ts, equal_nan=True)
    with pytest.raises(DpfError) as ei:
        gpu.sets_fetch(0, 1)
    assert ei.value.code == DPF_EINVAL and gpu.L_.dpf_sets_queries(gpu.h) == -1
    # a batch large enough for several staging pieces (dpf_query_batch moves 8 MiB at a time)
    params, X, Q, pick = make_problem(n=60000, nq=2000, dim=32, bucket_overflow=500, table_num=4, permutation_num=3, n_clusters=3)
    big = DpfIndex(params, metric="l2")
    big.fit(X)
    ob, vb = big.query_batch(pick, Q, steps=3)
    assert ob[-1] * 4 > 3 * (8 << 20), ob[-1]
    ow, vw = big.query_batch(pick, Q, steps=3, copy=False)
    assert np.array_equal(ob, ow) and np.array_equal(vb, np.asarray(vw))
    big.close()
    gpu.close()


@pytest.mark.gpu
def test_int8_scan_on_row_table_and_subindex_shards():
    # the int8 copy follows a handle's shard: row shards hold int8 rows of their own rows only (each with its own
    # window), table and sub-index shards scan their part of the forest; every shard's answer is the gather path's
    n, nq, dim = 9000, 80, 128
    params, X, Q, pick = _integer_problem(n, nq, dim, 0, seed=11, bucket_overflow=50, table_num=2, permutation_num=2)
    X[:4000] = np.minimum(X[:4000], 200.0)                         # the first row shard's window ends below the second's
    shards = [dict(row_begin=0, row_end=4000), dict(row_begin=4000, row_end=-1), dict(table_begin=1, table_end=3),
              dict(subindex_group=1, subindex_groups=3)]
    for kw in shards:
        fast = DpfIndex(params, metric="l2", **kw)
        slow = DpfIndex(dict(params, flags=2), metric="l2", **kw)
        fast.fit(X)
        slow.fit(X)
        rows = (4000 if kw.get("row_end") == 4000 else 5000) if "row_begin" in kw else n
        assert fast.stats()["base8_bytes"] == rows * 128
        for steps in (1, 2):
            fi, fs = fast.query_topk(pick, Q, steps=steps, k=10)
            st = fast.stats()
            assert st["last_path"] == 1 and st["last_scan_int8"] == 1, kw
            si, ss = slow.query_topk(pick, Q, steps=steps, k=10)
            assert np.array_equal(fi, si) and np.array_equal(fs, ss, equal_nan=True), kw
        fast.close()
        slow.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dim,offset", [(100, 300.0), (128, -1000.0), (40, 5.0e4), (768, 100.0)])
def test_fp16_copies_are_taken_around_the_mean_row(dim, offset):
    # l2 handles convert rows and queries to fp16 around the mean of the resident rows: the same answers as with
    # DPF_FLAG_NO_CENTRE and as the gather path, and on rows with a large common offset far fewer rows forwarded
    # (uncentred, 300 +- 1 rounds to steps of 0.25 and the filter has to allow for it)
    n, nq = (6000, 80) if dim <= 128 else (2500, 40)
    params, X, Q, pick = make_problem(n=n, nq=nq, dim=dim, bucket_overflow=50, table_num=2, permutation_num=2, n_clusters=6)
    X = np.ascontiguousarray(X + np.float32(offset), dtype=np.float32)
    Q = np.ascontiguousarray(Q + np.float32(offset), dtype=np.float32)
    X[7] = X[3]                                                     # equal rows: ties by id
    centred = DpfIndex(params, metric="l2")
    plain = DpfIndex(dict(params, flags=32), metric="l2")           # DPF_FLAG_NO_CENTRE
    slow = DpfIndex(dict(params, flags=2), metric="l2")             # DPF_FLAG_GATHER_ONLY
    for ix in (centred, plain, slow):
        ix.fit(X)
    fwd = {}
    for steps, k in ((1, 10), (0, 3)):
        si, ss = slow.query_topk(pick, Q, steps=steps, k=k)
        for name, ix in (("centred", centred), ("plain", plain)):
            ii, is_ = ix.query_topk(pick, Q, steps=steps, k=k)
            assert np.array_equal(ii, si) and np.array_equal(is_, ss, equal_nan=True), name
            st = ix.stats()
            fwd[name] = st["last_forwarded"] if st["last_path"] == 1 else 10 ** 9      # (the whole batch went the gather way)
        assert centred.stats()["last_path"] == 1
        assert fwd["centred"] <= fwd["plain"]
    if abs(offset) >= 300.0 and dim <= 128:
        assert 3 * fwd["centred"] < fwd["plain"], fwd
    # a second fit moves the mean: the copies are taken again around the new one
    centred.fit(X[:500] + np.float32(7.0))
    slow.fit(X[:500] + np.float32(7.0))
    ii, is_ = centred.query_topk(pick, Q, steps=1, k=10)
    si, ss = slow.query_topk(pick, Q, steps=1, k=10)
    assert np.array_equal(ii, si) and np.array_equal(is_, ss, equal_nan=True)
    for ix in (centred, plain, slow):
        ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("metric", ["l2", "dot"])
@pytest.mark.parametrize("dim,kind", [(100, "plain"), (128, "integers"), (128, "offset"), (768, "plain"), (36, "near")])
def test_rows_the_final_threshold_passed_by_are_dropped_safely(metric, dim, kind):
    # before the exact rerank a forward list loses the entries the query's final threshold would no longer forward
    # (recovered from the scan's one stored bound per entry with handle-wide maxima on the safe side);
    # DPF_OPT_KEEP_FORWARDED scores every forwarded row: the same answers, and the gather path's
    n, nq = (7000, 100) if dim <= 128 else (2500, 40)
    if kind == "integers":
        params, X, Q, pick = _integer_problem(n, nq, dim, 0, seed=5, bucket_overflow=50, table_num=2, permutation_num=2)
    else:
        params, X, Q, pick = make_problem(n=n, nq=nq, dim=dim, bucket_overflow=50, table_num=2, permutation_num=2, n_clusters=5)
        if kind == "offset":
            X = np.ascontiguousarray(X + np.float32(40.0)); Q = np.ascontiguousarray(Q + np.float32(40.0))
        if kind == "near":                                          # many rows at (nearly) the k-th best distance
            X[: n // 2] = X[n // 2: n // 2 * 2] + np.float32(1e-3) * np.random.default_rng(1).standard_normal((n // 2, dim), dtype=np.float32)
    pick = pick.astype(np.int32)
    pruned = DpfIndex(params, metric=metric)
    kept = DpfIndex(params, metric=metric)
    slow = DpfIndex(dict(params, flags=2), metric=metric)
    for ix in (pruned, kept, slow):
        ix.fit(X)
    kept.set_option("keep_forwarded", 1)
    for steps, k in ((1, 10), (2, 1), (0, 40)):
        si, ss = slow.query_topk(pick, Q, steps=steps, k=k)
        for ix in (pruned, kept):
            for rep in range(2):                                    # sized on the host, then with carried sizes
                ii, is_ = ix.query_topk(pick, Q, steps=steps, k=k)
                assert ix.stats()["last_path"] == 1
                assert np.array_equal(ii, si) and np.array_equal(is_, ss, equal_nan=True)
    for ix in (pruned, kept, slow):
        ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["plain", "integers"])
def test_list_chain_on_the_side_stream_changes_nothing(kind):
    # from a handle's second batch on the kernels that build the bucket lists run on a stream of the handle's own, beside
    # those that find the first radii (DPF_OPT_SIDE_STREAM, default 1): the same answers as with one stream, on torch's
    # default stream and on another one, over batches of changing size
    import torch
    if kind == "integers":
        params, X, Q, pick = _integer_problem(9000, 300, 128, 0, seed=11, bucket_overflow=50, table_num=3, permutation_num=2)
    else:
        params, X, Q, pick = make_problem(n=9000, nq=300, dim=100, bucket_overflow=50, table_num=3, permutation_num=2, n_clusters=6)
    pick = pick.astype(np.int32)
    two = DpfIndex(params, metric="l2")
    one = DpfIndex(params, metric="l2")
    for ix in (two, one):
        ix.fit(X)
    one.set_option("side_stream", 0)
    dev = torch.device("cuda", 0)
    dQ, dp = torch.from_numpy(Q).to(dev), torch.from_numpy(pick).to(dev)
    other = torch.cuda.Stream(device=dev)
    k = 10
    for rep, nq in enumerate((300, 300, 120, 300, 7, 300)):
        want_i, want_s = one.query_topk(pick[:nq], Q[:nq], steps=1, k=k)
        got_i, got_s = two.query_topk(pick[:nq], Q[:nq], steps=1, k=k)
        assert two.stats()["last_path"] == 1
        assert np.array_equal(got_i, want_i) and np.array_equal(got_s, want_s, equal_nan=True)
        # the device entry point on a stream that is not the default one
        ids = torch.full((nq, k), -7, dtype=torch.int32, device=dev)
        sc = torch.zeros((nq, k), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        with torch.cuda.stream(other):
            two.query_topk_device(dp.data_ptr(), dQ.data_ptr(), nq, 1, k, ids.data_ptr(), sc.data_ptr(), other.cuda_stream)
        other.synchronize()
        assert np.array_equal(ids.cpu().numpy(), want_i) and np.array_equal(sc.cpu().numpy(), want_s, equal_nan=True)
    for ix in (two, one):
        ix.close()


@pytest.mark.gpu
def test_streamed_fit_equals_whole_fit():
    # a base that arrives chunk by chunk (DPF_OPT_DEFER_BUILD / DPF_OPT_EXPECTED_ROWS): rows and hashes are appended as
    # the chunks pass, the forest (put / putInner in id order, RandomDrawTreeMap.java:1558-1790) is built once at the
    # end and equals the forest of one whole fit; a row-sharded handle keeps only its rows of every chunk
    from similaritysearchbyrdf_amd._lib import DpfError, DPF_ENOTFIT
    params, X, Q, pick = make_problem(n=30000, nq=200, bucket_overflow=60, table_num=3, permutation_num=2, n_clusters=12)
    qids = pick.astype(np.int32)
    whole = DpfIndex(params, metric="l2")
    whole.fit(X)
    want = whole.query_topk(qids, Q, steps=1, k=10)
    cuts = [0, 1, 4097, 4097, 11000, 25000, 30000]                    # uneven chunks, an empty one among them
    for kw in (dict(), dict(row_begin=7000, row_end=19000)):
        ix = DpfIndex(params, metric="l2", **kw)
        ix.set_option("expected_rows", 30000)
        ix.set_option("defer_build", 1)
        for a, b in zip(cuts[:-1], cuts[1:]):
            ix.fit(X[a:b])
        with pytest.raises(DpfError) as ei:                           # no forest yet
            ix.query_topk(qids, Q, steps=1, k=10)
        assert ei.value.code == DPF_ENOTFIT
        ix.set_option("defer_build", 0)
        ix.fit(X[:0])
        st = ix.stats()
        assert st["num_vectors"] == 30000 and st["misflag_events"] == whole.stats()["misflag_events"]
        if not kw:
            for a, b in zip(ix.snapshot(), whole.snapshot()):
                assert np.array_equal(a, b)
            got = ix.query_topk(qids, Q, steps=1, k=10)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True)
        else:
            ref = DpfIndex(params, metric="l2", **kw)
            ref.fit(X)
            for a, b in zip(ix.snapshot(), ref.snapshot()):
                assert np.array_equal(a, b)
            got, exp = ix.query_topk(qids, Q, steps=1, k=10), ref.query_topk(qids, Q, steps=1, k=10)
            assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1], equal_nan=True)
            assert (got[0][got[0] >= 0] >= 7000).all() and (got[0][got[0] >= 0] < 19000).all()
            ref.close()
        # fit_stream: the same through the binding's helper, and more rows afterwards still append
        ix2 = DpfIndex(params, metric="l2", **kw)
        ix2.fit_stream((X[a:b] for a, b in zip(cuts[:-1], cuts[1:])), total_rows=30000)
        for a, b in zip(ix2.snapshot(), ix.snapshot()):
            assert np.array_equal(a, b)
        ix2.close()
        ix.close()


@pytest.mark.gpu
@pytest.mark.parametrize("metric,dim,kw", [("l2", 100, {}), ("dot", 100, {}), ("l2", 128, {}), ("l2", 33, {}), ("l2", 256, {}),
                                           ("l2", 100, dict(row_begin=5000, row_end=14000)), ("l2", 100, dict(table_begin=1, table_end=4))])
def test_bucket_order_rows_scan_equals_gather(metric, dim, kw):
    # DPF_FLAG_BUCKET_ORDER_ROWS: the scan reads its fp16 rows from a second copy laid out in bucket order; same
    # ids and scores as the per-query gather rerank, bit for bit (whole tiles, ragged last tiles, row and table shards,
    # a second fit that rebuilds the copy)
    params, X, Q, pick = make_problem(n=20000, nq=300, dim=dim, bucket_overflow=80, table_num=3, permutation_num=2, n_clusters=12)
    fast = DpfIndex(dict(params, flags=64), metric=metric, **kw)
    slow = DpfIndex(dict(params, flags=2), metric=metric, **kw)
    qids = pick.astype(np.int32)
    qids[:5] = [0, 5, 127, 128, -1]
    for part in (X[:12000], X[12000:]):
        fast.fit(part)
        slow.fit(part)
        for steps in (0, 1, 3):
            fi, fs = fast.query_topk(qids, Q, steps=steps, k=10)
            assert fast.stats()["last_path"] == 1
            si, ss = slow.query_topk(qids, Q, steps=steps, k=10)
            assert np.array_equal(fi, si)
            assert np.array_equal(fs, ss, equal_nan=True)
