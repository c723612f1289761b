"""World-size-2 rehearsal of the sharded query on CPU (gloo), row shards and table
shards: shard bounds, the all-gather exchange and the merge rule, with the oracle
standing in for each rank's device stage.  The merged lists must equal the
unsharded oracle's."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from similaritysearchbyrdf_amd.dist import shard_bounds, table_bounds, num_tables_of, exchange, merge_topk_host
        from tests.common import make_problem
        params, X, Q, pick = make_problem(n=3001, nq=21, bucket_overflow=30, table_num=2, permutation_num=2)
        n = X.shape[0]
        lo, hi = shard_bounds(n, world, rank)
        orc = O.OracleIndex(params)
        orc.fit(X.astype(np.float64), threads=2)
        Q64 = Q.astype(np.float64)
        off, ids = orc.query_batch(pick, Q64, steps=1, threads=2)
        for metric in ("l2", "dot"):
            # this rank's stage: candidates restricted to its row shard, then its partial top-k
            loc_off = [0]
            loc = []
            for q in range(len(pick)):
                c = ids[off[q]:off[q + 1]]
                c = c[(c >= lo) & (c < hi)]
                loc.append(c)
                loc_off.append(loc_off[-1] + len(c))
            loc_ids = np.concatenate(loc) if loc else np.zeros(0, dtype=np.int32)
            pi, ps = orc.topk(Q64, np.asarray(loc_off), loc_ids, k=10, metric=metric, threads=2)
            g_ids, g_sc = exchange(torch.from_numpy(pi), torch.from_numpy(ps))
            assert g_ids.shape == (world, len(pick), 10)
            m_ids, m_sc = merge_topk_host(g_ids.numpy(), g_sc.numpy(), metric)
            want_ids, want_sc = orc.topk(Q64, off, ids, k=10, metric=metric, threads=2)
            assert np.array_equal(m_ids, want_ids), metric
            assert np.array_equal(np.nan_to_num(m_sc), np.nan_to_num(want_sc)), metric
        # table shards: this rank's stage is the union of the candidate lists of ITS tables
        # (RandomDrawTreeMap.java:742-797 per table, Set union of DensevectorRDFInit.scala:420,428)
        L = num_tables_of(params)
        t_lo, t_hi = table_bounds(L, world, rank)
        loc, loc_off = [], [0]
        for q in range(len(pick)):
            per_table = [orc.get_similar(t, int(pick[q]), Q64[q], 1) for t in range(t_lo, t_hi)]
            c = np.unique(np.concatenate(per_table)) if per_table else np.zeros(0, dtype=np.int32)
            loc.append(c.astype(np.int32))
            loc_off.append(loc_off[-1] + len(c))
        loc_ids = np.concatenate(loc) if loc else np.zeros(0, dtype=np.int32)
        for metric in ("l2", "dot"):
            pi, ps = orc.topk(Q64, np.asarray(loc_off), loc_ids, k=10, metric=metric, threads=2)
            g_ids, g_sc = exchange(torch.from_numpy(pi), torch.from_numpy(ps))
            m_ids, m_sc = merge_topk_host(g_ids.numpy(), g_sc.numpy(), metric)
            want_ids, want_sc = orc.topk(Q64, off, ids, k=10, metric=metric, threads=2)
            assert np.array_equal(m_ids, want_ids), ("tables", metric)
            assert np.array_equal(np.nan_to_num(m_sc), np.nan_to_num(want_sc)), ("tables", metric)
        tb = [table_bounds(L, world, r) for r in range(world)]
        assert tb[0][0] == 0 and tb[-1][1] == L and all(tb[i][1] == tb[i + 1][0] for i in range(world - 1))
        # shard bounds tile the id space exactly
        b = [shard_bounds(n, world, r) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def _grid_worker(rank, world, port, ret):
    """Tables mode on a grid of query groups x table groups: each rank answers its query group from
    its table group; one all-gather, then a merge per query group."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from similaritysearchbyrdf_amd.dist import (shard_bounds, table_bounds, num_tables_of, exchange, merge_grid_host,
                                                    grid_of)
        from tests.common import make_problem
        params, X, Q, pick = make_problem(n=2501, nq=23, bucket_overflow=30, table_num=2, permutation_num=2)
        L = num_tables_of(params)
        qg, tg = 2, 2
        assert grid_of(world, L)[0] * grid_of(world, L)[1] == world
        nq = len(pick)
        orc = O.OracleIndex(params)
        orc.fit(X.astype(np.float64), threads=2)
        Q64 = Q.astype(np.float64)
        off, ids = orc.query_batch(pick, Q64, steps=1, threads=2)
        q_lo, q_hi = shard_bounds(nq, qg, rank // tg)
        t_lo, t_hi = table_bounds(L, tg, rank % tg)
        m = -(-nq // qg)
        for metric in ("l2", "dot"):
            loc, loc_off = [], [0]
            for q in range(q_lo, q_hi):
                per_table = [orc.get_similar(t, int(pick[q]), Q64[q], 1) for t in range(t_lo, t_hi)]
                c = np.unique(np.concatenate(per_table)) if per_table else np.zeros(0, dtype=np.int32)
                loc.append(c.astype(np.int32))
                loc_off.append(loc_off[-1] + len(c))
            loc_ids = np.concatenate(loc) if loc else np.zeros(0, dtype=np.int32)
            pi, ps = orc.topk(Q64[q_lo:q_hi], np.asarray(loc_off), loc_ids, k=10, metric=metric, threads=2)
            l_ids = np.full((m, 10), -1, dtype=np.int32)
            l_sc = np.zeros((m, 10))
            l_ids[:q_hi - q_lo], l_sc[:q_hi - q_lo] = pi, ps
            g_ids, g_sc = exchange(torch.from_numpy(l_ids), torch.from_numpy(l_sc))
            m_ids, m_sc = merge_grid_host(g_ids.numpy(), g_sc.numpy(), qg, tg, nq, metric)
            want_ids, want_sc = orc.topk(Q64, off, ids, k=10, metric=metric, threads=2)
            assert np.array_equal(m_ids, want_ids), metric
            assert np.array_equal(np.nan_to_num(m_sc), np.nan_to_num(want_sc)), metric
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_query_by_table_grid_gloo():
    world = 4
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_grid_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == ["ok"] * world


def test_grid_of():
    from similaritysearchbyrdf_amd.dist import grid_of
    assert grid_of(1, 30) == (1, 1) and grid_of(2, 30) == (1, 2) and grid_of(4, 30) == (1, 4) and grid_of(8, 30) == (2, 4)
    assert grid_of(8, 1) == (8, 1) and grid_of(8, 3) == (4, 2) and grid_of(3, 30) == (1, 3)
    for w in range(1, 33):
        for L in (1, 2, 30):
            q, t = grid_of(w, L)
            assert q * t == w and t <= L


@pytest.mark.parametrize("world", [2])
def test_sharded_exchange_and_merge_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret.get(r) for r in range(world)] == ["ok"] * world


def test_shard_bounds_edge_cases():
    from similaritysearchbyrdf_amd.dist import shard_bounds, merge_topk_host
    for n in (0, 1, 7, 8, 1_000_000):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
    # merge: ties by id, padding ignored, fewer than k survivors padded
    ids = np.array([[[5, 9, -1]], [[2, 7, -1]]], dtype=np.int32)
    sc = np.array([[[1.0, 3.0, np.nan]], [[1.0, 2.0, np.nan]]])
    mi, ms = merge_topk_host(ids, sc, "l2")
    assert mi.tolist() == [[2, 5, 7]] and ms.tolist() == [[1.0, 1.0, 2.0]]
    mi, ms = merge_topk_host(ids, sc, "dot")
    assert mi.tolist() == [[9, 7, 2]]
    # an id present in two lists (table shards) is taken once
    ids = np.array([[[5, 9, 11]], [[5, 7, 9]]], dtype=np.int32)
    sc = np.array([[[1.0, 3.0, 4.0]], [[1.0, 2.0, 3.0]]])
    mi, ms = merge_topk_host(ids, sc, "l2")
    assert mi.tolist() == [[5, 7, 9]] and ms.tolist() == [[1.0, 2.0, 3.0]]


def test_row_addressable_generators():
    # any rank can produce any row range of the chunk-seeded base without the rows before it
    from similaritysearchbyrdf_amd import synthetic as S
    n, dim = 3 * S.ROW_CHUNK + 1234, 8
    whole = S.unit_gaussian_rows(0, n, dim)
    assert whole.shape == (n, dim) and np.allclose(np.linalg.norm(whole, axis=1), 1.0, atol=1e-5)
    for lo, hi in ((0, 10), (S.ROW_CHUNK - 3, S.ROW_CHUNK + 5), (2 * S.ROW_CHUNK, n), (n - 1, n), (500, 500)):
        assert np.array_equal(S.unit_gaussian_rows(lo, hi, dim), whole[lo:hi])
    assert np.array_equal(np.concatenate(list(S.unit_gaussian_chunks(n, dim))), whole)
    Q, pick = S.unit_gaussian_queries(n, dim, 40)
    assert ((Q * whole[pick]).sum(axis=1) > 0.99).all()              # noisy copies of the picked rows
    batches = S.unit_gaussian_queries(n, dim, 10, batches=4)
    assert len(batches) == 4 and np.array_equal(batches[0][0], Q[:10]) and np.array_equal(batches[3][1], pick[30:40])
