"""Sharded multi-GPU batch query: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI; "gloo" for CPU rehearsal).

Two ways to split one batch, both ending in the same exchange (ONE all-gather
of the per-rank top-k lists, ids and scores packed into one block per rank,
then a merge by score, then id, with repeated ids taken once):

* "tables" (used when the forest has at least as many tables as there are
  ranks): everything is replicated; the G ranks form a grid of QG query groups
  x TG table groups (grid_of), rank r = (r // TG, r % TG) answers the queries
  of its query group from the tables [t*L/TG, (t+1)*L/TG) of its table group.
  A query's candidate set is the union over tables
  (DensevectorRDFInit.scala:420,428), so the union of the TG per-rank sets of
  a query group is the whole set.  Table groups divide the scan (every rank
  reads its tables' rows once, however few queries it serves); query groups
  divide what is paid per query (hashing, seed thresholds, exact rerank).
* "rows": the forest and the hash family are replicated; the G ranks form a
  grid of QG query groups x RG row groups, rank r = (r // RG, r % RG) holds the
  base rows [j*N/RG, (j+1)*N/RG) of its row group j and answers the queries of
  its query group: it hashes and probes them over all tables but scores only
  the candidates inside its row shard.  Used for single-table forests, when the
  base rows do not fit one GPU, and on request (mode="rows").

* "subindexes": everything replicated; QG query groups x SG sub-index groups,
  rank r = (r // SG, r % SG) probes, in every table, only the sub-indexes p with
  p % SG == its group - the reference's own partitioning unit
  (Partitioner.scala:27-64: a second-level LSH assigns every hash to one of
  2^partitionBits sub-indexes).  A steps-s query touches the sub-indexes within
  s bit flips of its own, so the load per rank is even only on average.

The reference is a single JVM process (its "distribution" is the in-process
sub-index partitioning of Partitioner.scala:27-64), so there is no reference
call pattern to mirror here.
"""
import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous row shard of rank `rank`: [lo, hi)."""
    return n * rank // world, n * (rank + 1) // world


def table_bounds(num_tables, world, rank):
    """Contiguous table shard of rank `rank`: [lo, hi)."""
    return num_tables * rank // world, num_tables * (rank + 1) // world


def grid_of(world, num_tables):
    """(query groups, table groups) for `world` ranks in "tables" mode: the largest divisor TG of world
    with TG <= 2 * sqrt(world) and TG <= num_tables (2 ranks: 1 x 2, 4: 1 x 4, 8: 2 x 4, measured best on
    an MI355X), the other factor as query groups."""
    tg = 1
    for d in range(1, world + 1):
        if world % d == 0 and d * d <= 4 * world and d <= num_tables:
            tg = d
    return world // tg, tg


def rows_grid_of(world):
    """(query groups, row groups) for `world` ranks in "rows" mode: two row groups, the other factor as query
    groups (odd world sizes: one query group, `world` row groups).  Row groups divide the rows a rank's scan
    reads but every rank still hashes and probes all of its queries over all tables; query groups divide that
    per-query work.  Measured per-rank batch time on one MI355X, c3 shape (tools/shard_timing.py), query groups x
    row groups: 1x2 2.03 ms, 2x1 2.07; 2x2 1.44, 1x4 1.67; 4x2 1.10, 2x4 1.15, 8x1 1.34, 1x8 1.43.  A base that
    does not fit two GPUs needs more row groups: pass grid=(qg, rg)."""
    if world >= 2 and world % 2 == 0:
        return world // 2, 2
    return 1, world


def merge_grid_host(ids, scores, qg, tg, nq, metric):
    """Host-side merge of the gathered lists [G, ceil(nq / qg), k] of a QG x TG grid into [nq, k]:
    used to rehearse the exchange on CPU tensors (gloo)."""
    out_i, out_s = [], []
    for j in range(qg):
        lo, hi = shard_bounds(nq, qg, j)
        mi, ms = merge_topk_host(ids[j * tg:(j + 1) * tg, :hi - lo], scores[j * tg:(j + 1) * tg, :hi - lo], metric)
        out_i.append(mi)
        out_s.append(ms)
    return np.concatenate(out_i), np.concatenate(out_s)


def merge_topk_host(ids, scores, metric):
    """Host-side merge of G partial lists [G, nq, k] (ids < 0 = padding; an id present in
    several lists is taken once): used to rehearse the exchange on CPU tensors (gloo); on
    GPUs the merge is k_merge_topk."""
    G, nq, k = ids.shape
    out_ids = np.full((nq, k), -1, dtype=np.int32)
    out_sc = np.full((nq, k), np.nan, dtype=np.float64)
    for q in range(nq):
        i = ids[:, q, :].reshape(-1)
        s = scores[:, q, :].reshape(-1)
        keep = i >= 0
        i, s = i[keep], s[keep]
        i, first = np.unique(i, return_index=True)
        s = s[first]
        key = -s if metric == "dot" else s
        order = np.lexsort((i, key))[:k]
        out_ids[q, :len(order)] = i[order]
        out_sc[q, :len(order)] = s[order]
    return out_ids, out_sc


def packed_bytes(m, k):
    """Bytes of one rank's packed block: scores [m, k] float64, ids [m, k] int32, padded to 8."""
    return (m * k * 12 + 7) // 8 * 8


def packed_views(block, m, k):
    """(scores [.., m, k] float64, ids [.., m, k] int32) views into packed blocks [.., packed_bytes(m, k)] (uint8)."""
    import torch
    n = m * k
    lead = tuple(block.shape[:-1])
    return (block[..., :n * 8].view(torch.float64).view(lead + (m, k)),
            block[..., n * 8:n * 12].view(torch.int32).view(lead + (m, k)))


def exchange_packed(block, out=None, group=None):
    """ONE all-gather of every rank's packed block [B] (uint8) -> [G, B]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo" and block.is_cuda:
        # CPU rehearsal of the exchange with device-resident lists (several ranks sharing one GPU)
        g = exchange_packed(block.cpu(), None, group).to(block.device)
        if out is not None:
            out.copy_(g)
            return out
        return g
    g = out if out is not None else torch.empty((world, block.numel()), dtype=torch.uint8, device=block.device)
    dist.all_gather_into_tensor(g.view(-1), block, group=group)
    return g


def exchange(local_ids, local_scores, group=None):
    """all-gather the [nq, k] partial lists of every rank -> ([G, nq, k], [G, nq, k]), packed into one collective."""
    import torch
    m, k = local_ids.shape
    block = torch.zeros(packed_bytes(m, k), dtype=torch.uint8, device=local_ids.device)
    sc, ids = packed_views(block, m, k)
    sc.copy_(local_scores)
    ids.copy_(local_ids)
    g_sc, g_ids = packed_views(exchange_packed(block, None, group), m, k)
    return g_ids, g_sc


def num_tables_of(params):
    """L = rows of the chain table (tableNum * permutationNum of the reference's settings)."""
    return int(np.asarray(params["chain_fn"]).shape[0])


class ShardedDpf:
    """One rank's share of a sharded index on its GPU (see the module docstring for the two modes)."""

    def __init__(self, params, n_total, metric="l2", top_k=10, dtype=np.float32, device=0, rank=0, world=1,
                 group=None, mode=None, grid=None, load_from=None, num_tables=None, partition_bits=None):
        """params: the hash family etc. (as DpfIndex takes); or, with load_from = the path of a file written by
        DpfIndex.save from an unsharded handle, params may be None: every rank then brings up only its shard of
        that file (dpf_load_shard; num_tables / partition_bits of the saved index must be given)."""
        from ._lib import DpfIndex
        self.rank, self.world, self.group, self.metric = rank, world, group, metric
        L = num_tables_of(params) if params is not None else int(num_tables)
        pb = int(params.get("partition_bits", 3)) if params is not None else int(partition_bits)
        self.mode = mode or ("tables" if world > 1 and L >= world else "rows")
        if self.mode not in ("tables", "rows", "subindexes"):
            raise ValueError("mode must be 'tables', 'rows' or 'subindexes'")
        # qg query groups x sg shard groups (table, row or sub-index groups); rank r = (r // sg, r % sg)
        self.qg, self.sg = grid or (rows_grid_of(world) if self.mode == "rows" else grid_of(world, L if self.mode == "tables" else 1 << pb))
        limit = {"tables": L, "rows": max(n_total, 1), "subindexes": 1 << pb}[self.mode]
        if self.qg * self.sg != world or self.sg > limit:
            raise ValueError("grid %r does not cover %d ranks in mode %s" % ((self.qg, self.sg), world, self.mode))
        self.lo, self.hi, self.t_lo, self.t_hi = 0, n_total, 0, L
        shard = {}
        if self.mode == "tables":
            self.t_lo, self.t_hi = table_bounds(L, self.sg, rank % self.sg)
            shard = dict(table_begin=self.t_lo, table_end=self.t_hi)
        elif self.mode == "rows":
            self.lo, self.hi = shard_bounds(n_total, self.sg, rank % self.sg)
            shard = dict(row_begin=self.lo, row_end=self.hi)
        elif self.sg > 1:
            shard = dict(subindex_group=rank % self.sg, subindex_groups=self.sg)
        if load_from is not None:
            self.index = DpfIndex.load(load_from, device=device, **(shard or dict(row_begin=0, row_end=-1)))
        else:
            self.index = DpfIndex(params, metric=metric, top_k=top_k, dtype=dtype, device=device, **shard)
        self.tg = self.sg            # (older name)
        self._bufs = None
        self.always_exchange = False  # True: a single rank also goes through the all-gather and the merge (rehearsals)

    def fit(self, X):
        """Every rank sees every row (hashes build the replicated forest); in "rows" mode only the
        shard stays in HBM."""
        self.index.fit(X)

    def fit_stream(self, chunks, total_rows=0):
        """fit() for a base that no rank holds on its host at once: an iterable of [rows, dim] chunks in id order (every
        rank iterates over the same chunks, e.g. from a row-addressable generator or a memory-mapped file).  Each chunk
        is hashed on the device and the rows of this rank's shard are kept; the forest is built once at the end."""
        self.index.fit_stream(chunks, total_rows)

    def query_topk(self, d_qids, d_Q, steps, k, out_ids, out_scores, stream=None):
        """Device tensors in, merged device tensors out.  `stream` must be torch's current stream (the
        collective and the copies below are ordered on that one); the device stage blocks the host once
        per batch (dpf_query_topk_device), the results are complete once `stream` has drained."""
        import torch
        nq = d_Q.shape[0]
        if d_Q.is_cuda:
            cur = torch.cuda.current_stream(d_Q.device).cuda_stream
            if (stream or 0) != cur:
                raise ValueError("ShardedDpf.query_topk: `stream` must be torch's current stream on the index's device "
                                 "(the all-gather and the copies are ordered on that one)")
        if self.world == 1 and not self.always_exchange:
            self.index.query_topk_device(d_qids.data_ptr(), d_Q.data_ptr(), nq, steps, k, out_ids.data_ptr(),
                                         out_scores.data_ptr(), stream)
            return
        qg, tg = self.qg, self.sg
        q_lo, q_hi = shard_bounds(nq, qg, self.rank // tg)        # this rank's query group
        m = -(-nq // qg)                                           # rows every rank contributes to the exchange
        if self._bufs is None or self._bufs[0] != (m, k):
            dev = d_Q.device
            block = torch.zeros(packed_bytes(m, k), dtype=torch.uint8, device=dev)
            l_sc, l_ids = packed_views(block, m, k)
            l_ids.fill_(-1)                                        # rows past this rank's query group stay empty lists
            self._bufs = ((m, k), block, l_ids, l_sc,
                          torch.empty((self.world, block.numel()), dtype=torch.uint8, device=dev),
                          torch.empty((m, k), dtype=torch.int32, device=dev),
                          torch.empty((m, k), dtype=torch.float64, device=dev))
        _, block, l_ids, l_sc, gathered, t_ids, t_sc = self._bufs
        if q_hi > q_lo:
            qid_ptr = d_qids[q_lo:q_hi].data_ptr() if d_qids is not None else None
            self.index.query_topk_device(qid_ptr, d_Q[q_lo:q_hi].data_ptr(), q_hi - q_lo, steps, k, l_ids.data_ptr(),
                                         l_sc.data_ptr(), stream)
        exchange_packed(block, gathered, self.group)               # [G, packed block]: one collective per batch
        for j in range(qg):
            lo, hi = shard_bounds(nq, qg, j)
            if hi == lo:
                continue
            src = gathered[j * tg].data_ptr()                      # the tg blocks of query group j sit back to back
            if hi - lo == m:
                self.index.merge_topk_packed_device(src, block.numel(), tg, m, k, out_ids[lo:hi].data_ptr(),
                                                    out_scores[lo:hi].data_ptr(), stream)
            else:                                                  # a short last group: merge all m rows aside
                self.index.merge_topk_packed_device(src, block.numel(), tg, m, k, t_ids.data_ptr(), t_sc.data_ptr(), stream)
                out_ids[lo:hi].copy_(t_ids[:hi - lo])
                out_scores[lo:hi].copy_(t_sc[:hi - lo])
