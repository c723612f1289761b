"""Sharded multi-GPU batch query: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI; "gloo" for CPU rehearsal).

Two ways to split one batch, both ending in the same exchange (one all-gather
per array of the per-rank top-k lists, then a merge by score, then id, with
repeated ids taken once):

* "tables" (used when the forest has at least as many tables as there are
  ranks): everything is replicated, rank r probes and scans only tables
  [r*L/G, (r+1)*L/G).  A query's candidate set is the union over tables
  (DensevectorRDFInit.scala:420,428), so the union of the per-rank sets is the
  whole set.  Probe, scan and rerank work all divide by G.
* "rows": the forest and the hash family are replicated, base rows are sharded
  contiguously, rank r owning ids [r*N/G, (r+1)*N/G).  Every rank hashes and
  probes the whole query batch but scores only the candidates inside its shard.
  Used for single-table forests, and when the base rows do not fit one GPU.

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


def exchange(local_ids, local_scores, group=None):
    """all-gather the [nq, k] partial lists of every rank -> ([G, nq, k], [G, nq, k])."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo" and local_ids.is_cuda:
        # CPU rehearsal of the exchange with device-resident lists (several ranks sharing one GPU)
        c_ids, c_sc = exchange(local_ids.cpu(), local_scores.cpu(), group)
        return c_ids.to(local_ids.device), c_sc.to(local_scores.device)
    g_ids = torch.empty((world,) + tuple(local_ids.shape), dtype=local_ids.dtype, device=local_ids.device)
    g_sc = torch.empty((world,) + tuple(local_scores.shape), dtype=local_scores.dtype, device=local_scores.device)
    # concatenated-along-dim-0 output form: accepted by both RCCL and gloo
    dist.all_gather_into_tensor(g_ids.view((-1,) + tuple(local_ids.shape[1:])), local_ids.contiguous(), group=group)
    dist.all_gather_into_tensor(g_sc.view((-1,) + tuple(local_scores.shape[1:])), local_scores.contiguous(), group=group)
    return g_ids, g_sc


def num_tables_of(params):
    """L = rows of the chain table (tableNum * permutationNum of the reference's settings)."""
    return int(np.asarray(params["chain_fn"]).shape[0])


class ShardedDpf:
    """One rank's share of a sharded index on its GPU (see the module docstring for the two modes)."""

    def __init__(self, params, n_total, metric="l2", top_k=10, dtype=np.float32, device=0, rank=0, world=1,
                 group=None, mode=None):
        from ._lib import DpfIndex
        self.rank, self.world, self.group, self.metric = rank, world, group, metric
        L = num_tables_of(params)
        self.mode = mode or ("tables" if world > 1 and L >= world else "rows")
        if self.mode == "tables":
            self.lo, self.hi = 0, n_total
            self.t_lo, self.t_hi = table_bounds(L, world, rank)
            self.index = DpfIndex(params, metric=metric, top_k=top_k, dtype=dtype, device=device,
                                  table_begin=self.t_lo, table_end=self.t_hi)
        else:
            self.lo, self.hi = shard_bounds(n_total, world, rank)
            self.t_lo, self.t_hi = 0, L
            self.index = DpfIndex(params, metric=metric, top_k=top_k, dtype=dtype, device=device,
                                  row_begin=self.lo, row_end=self.hi)
        self._bufs = None

    def fit(self, X):
        """Every rank sees every row (hashes build the replicated forest); in "rows" mode only the
        shard stays in HBM."""
        self.index.fit(X)

    def query_topk(self, d_qids, d_Q, steps, k, out_ids, out_scores, stream=None):
        """Device tensors in, merged device tensors out (asynchronous on `stream`)."""
        import torch
        nq = d_Q.shape[0]
        if self.world == 1:
            self.index.query_topk_device(d_qids.data_ptr(), d_Q.data_ptr(), nq, steps, k, out_ids.data_ptr(),
                                         out_scores.data_ptr(), stream)
            return
        if self._bufs is None or self._bufs[0].shape != (nq, k):
            dev = d_Q.device
            self._bufs = (torch.empty((nq, k), dtype=torch.int32, device=dev),
                          torch.empty((nq, k), dtype=torch.float64, device=dev))
        l_ids, l_sc = self._bufs
        self.index.query_topk_device(d_qids.data_ptr(), d_Q.data_ptr(), nq, steps, k, l_ids.data_ptr(),
                                     l_sc.data_ptr(), stream)
        g_ids, g_sc = exchange(l_ids, l_sc, self.group)
        self.index.merge_topk_device(g_ids.data_ptr(), g_sc.data_ptr(), self.world, nq, k, out_ids.data_ptr(),
                                     out_scores.data_ptr(), stream)
