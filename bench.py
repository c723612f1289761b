#!/usr/bin/env python3
"""Batch-query benchmark of the MI355X Dynamic Partition Forest engine.

One "step" = one pass of the hot path (LSH projection -> forest probe ->
candidate rerank/top-k) over one batch of `nq` synthetic queries that is
already resident in HBM.  Prints ONE JSON line on rank 0.

  python bench.py                       # 1 GPU, workload c3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json `configs`):
  c3  100-d, 1M base / 10k queries, full forest (tableNum 10 x permutationNum 3 = 30
      trees, 3 partition bits), 1-step search -- the configuration the metric
      "queries/sec @ recall@10, GloVe-100d 1M base / 10k query batch" is quoted on.
      GloVe itself is not available offline: clustered synthetic vectors of the same shape.
  c2  same data, a single tree, 0-step search (configs[1]).
  c4  128-d SIFT-shaped integer-valued vectors, 1M / 10k, forest as c3.
  c5  768-d unit-normalised Gaussian vectors, forest as c3 over a family of 768 functions
      (configs[4] is 10M / 100k on 8 GPUs: pass --base-rows / --queries for the share to run).

With N > 1 everything is replicated and the ranks form a grid of query groups x table
groups: a rank answers its query group from its range of the trees, the per-rank top-k
lists are all-gathered (RCCL, one collective) and merged: total work is fixed, so
scaling is "strong".  Single-tree forests shard the base rows instead.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak


def workload(name, n, nq, dim=None):
    from similaritysearchbyrdf_amd import synthetic
    if name == "c2":
        params = synthetic.make_gaussian_family(100, 1, 32, 3, seed=88387, part_seed=88388)
        X, Q, pick = synthetic.make_clustered(n, 100, nq)
        steps = 0
    elif name == "c3":
        params = synthetic.make_family(100, table_num=10, permutation_num=3, chain_length=32, family="angle",
                                       family_size=100, partition_bits=3, orthogonal=True)
        X, Q, pick = synthetic.make_clustered(n, 100, nq)
        steps = 1
    elif name == "c4":
        params = synthetic.make_family(128, table_num=10, permutation_num=3, chain_length=32, family="angle",
                                       family_size=128, partition_bits=3, orthogonal=True)
        X, Q, pick = synthetic.make_sift_like(n, 128, nq)
        steps = 1
    elif name == "c5":
        d = dim or 768
        params = synthetic.make_family(d, table_num=10, permutation_num=3, chain_length=32, family="angle",
                                       family_size=d, partition_bits=3, orthogonal=True)
        X, Q, pick = synthetic.make_unit_gaussian(n, d, nq)
        steps = 1
    else:
        raise SystemExit("unknown workload " + name)
    params.update(bucket_overflow=500, dir_node_size=32, bucket_bits=28, exclude_self="compat")
    return params, X, Q, pick, steps


def pmc_traffic(workload_name, n, nq, world, kernel, steps):
    """HBM-side bytes per rerank launch from the committed rocprofv3 PMC passes
    (FETCH_SIZE doubled as gfx950 needs, plus WRITE_SIZE); None when no pass
    matches this configuration."""
    import glob
    if world != 1 or n != 1_000_000 or nq != 10_000:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_%s_pmc_traffic.json" % (workload_name, kernel))))
    if not files:
        return None
    with open(files[-1]) as f:
        rec = json.load(f)
    return rec.get("traffic_bytes_per_launch") if rec.get("search_steps", 1) == steps else None


def ground_truth_l2(torch, dX, dQ, k, chunk=500):
    """Exact top-k by squared L2 distance (fp32 GEMM form); only used to report recall."""
    xn = (dX * dX).sum(dim=1)
    out = []
    for s in range(0, dQ.shape[0], chunk):
        q = dQ[s:s + chunk]
        d = xn[None, :] - 2.0 * (q @ dX.T)
        out.append(torch.topk(d, k, dim=1, largest=False).indices)
    return torch.cat(out).cpu().numpy()


def cpu_baseline(params, X, Q, pick, steps, k, metric, target_s=12.0):
    """The CPU restatement of the reference path, reference threading (5 threads, each
    owning a table range, DensevectorRDFInit.scala:344-350; single-threaded rerank,
    DensevectorRDFInit.scala:477-505), on a bounded sample of the same batch."""
    from oracle import oracle as O
    cores = os.cpu_count() or 8
    fit_threads = max(1, min(16, cores))
    t0 = time.time()
    orc = O.OracleIndex(params)
    orc.fit(X.astype(np.float64), threads=fit_threads)
    fit_s = time.time() - t0
    Q64 = Q.astype(np.float64)

    def run(m, threads, rerank_threads):
        t = time.time()
        off, ids = orc.query_batch(pick[:m], Q64[:m], steps=steps, threads=threads, mode=0 if threads == 5 else 1)
        orc.topk(Q64[:m], off, ids, k=k, metric=metric, threads=rerank_threads)
        return time.time() - t

    probe = min(64, len(Q64))
    dt = run(probe, 5, 1)
    m = int(min(len(Q64), max(probe, probe * target_s / max(dt, 1e-6))))
    dt = run(m, 5, 1)
    all_threads = max(1, min(16, cores))
    dt_all = run(m, all_threads, all_threads)
    res = {"value": m / dt, "unit": "queries/s", "cores": 5, "kind": "port",
           "sample": "%d of the batch's queries against the full index, %d-step search + top-%d rerank; "
                     "C restatement of the reference path (oracle/), reference threading: 5 query threads over "
                     "table ranges, single-threaded rerank" % (m, steps, k),
           "all_cores_value": m / dt_all, "all_cores": all_threads, "index_build_s": round(fit_s, 2),
           "misflag_events": orc.misflag_events}
    orc.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--base-rows", dest="n", type=int, default=1_000_000)
    ap.add_argument("--queries", dest="nq", type=int, default=10_000)
    ap.add_argument("--dim", type=int, default=None, help="c5 only: dimensions of the vectors (default 768)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="l2", choices=["l2", "dot"])
    ap.add_argument("--search-steps", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--rerank", default="auto", choices=["auto", "gather"],
                    help="auto: bucket-major bf16 scan + exact rerank where it applies; gather: per-query fp64 gather kernel")
    ap.add_argument("--verify", action="store_true", help="rank 0 also runs an unsharded index and compares the merged ids")
    ap.add_argument("--grid", default=None, help="QxT: query groups x table groups of the tables mode (default: chosen from --gpus)")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL, one GPU per rank) or gloo (rehearsal: ranks may share a GPU)")
    args = ap.parse_args()

    import torch
    from similaritysearchbyrdf_amd.dist import ShardedDpf

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    if args.backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    params, X, Q, pick, steps = workload(args.workload, args.n, args.nq, args.dim)
    if args.rerank == "gather":
        params["flags"] = 2          # DPF_FLAG_GATHER_ONLY
    if args.search_steps is not None:
        steps = args.search_steps
    n, nq, k = X.shape[0], Q.shape[0], args.k
    grid = tuple(int(v) for v in args.grid.lower().split("x")) if args.grid else None
    sharded = ShardedDpf(params, n, metric=args.metric, top_k=k, dtype=np.float32, device=local, rank=rank, world=world,
                         grid=grid)
    ix = sharded.index
    t0 = time.time()
    sharded.fit(X)
    fit_s = time.time() - t0

    dQ = torch.from_numpy(Q).to(dev)
    dqid = torch.from_numpy(pick.astype(np.int32)).to(dev)
    d_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    d_sc = torch.empty((nq, k), dtype=torch.float64, device=dev)

    def step():
        # hash -> probe -> scan -> rerank on this rank's shard (of the tables, or of the rows for a
        # single-table forest); with N > 1 one all-gather of the packed
        # partial top-k lists and the merge kernel
        sharded.query_topk(dqid, dQ, steps, k, d_ids, d_sc, torch.cuda.current_stream().cuda_stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    rr_ms, rr_bytes = [], 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # dominant kernel: the rerank launch, timed with HIP events on its own stream
    ix.set_profiling(True)
    for _ in range(max(3, min(args.steps, 10))):
        step()
        ms, by = ix.last_rerank_profile()
        rr_ms.append(ms)
        rr_bytes = by
    fence()
    st = ix.stats()
    final_ids = d_ids.cpu().numpy()

    if args.verify and rank == 0:
        from similaritysearchbyrdf_amd._lib import DpfIndex
        full = DpfIndex(params, metric=args.metric, top_k=k, dtype=np.float32, device=local)
        full.fit(X)
        want_ids, _ = full.query_topk(pick.astype(np.int32), Q, steps=steps, k=k)
        if not np.array_equal(want_ids, final_ids):
            raise SystemExit("verify failed: merged sharded result differs from the unsharded index")
        print("verify ok: %d-way sharded result equals the unsharded index" % world, file=sys.stderr)
        full.close()
    recall = None
    if not args.no_recall and rank == 0:
        dX = torch.from_numpy(X).to(dev)
        gt = ground_truth_l2(torch, dX, dQ, k)
        del dX
        hit = sum(len(np.intersect1d(final_ids[q][final_ids[q] >= 0], gt[q])) for q in range(nq))
        recall = hit / float(nq * k)
    if world > 1:
        dist.barrier()

    gather = None
    if rank == 0 and world == 1 and st["last_path"] == 1:
        # the per-query gather rerank on the same batch (untimed): its bytes are the classic rerank roofline
        # (every de-duplicated candidate row once per query), and it yields the de-duplicated candidate count
        from similaritysearchbyrdf_amd._lib import DpfIndex
        gi = DpfIndex(dict(params, flags=2), metric=args.metric, top_k=k, dtype=np.float32, device=local)
        gi.fit(X)
        gi.set_profiling(True)
        g_ms = []
        for _ in range(3):
            gi.query_topk_device(dqid.data_ptr(), dQ.data_ptr(), nq, steps, k, d_ids.data_ptr(), d_sc.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
            ms, by = gi.last_rerank_profile()
            g_ms.append(ms)
        gst = gi.stats()
        same = bool(np.array_equal(d_ids.cpu().numpy(), final_ids))
        gather = {"kernel": "k_rerank", "bound": "hbm", "achieved": by / (float(np.mean(g_ms[1:])) * 1e-3) / 1e9,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / (float(np.mean(g_ms[1:])) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "algorithmic_bytes_per_launch": by, "ms_per_launch": float(np.mean(g_ms[1:])),
                  "unique_candidates_per_query": gst["last_unique_candidates"] / nq, "same_ids_as_timed_path": same}
        gi.close()
    if rank == 0:
        # on the bucket-major path the profile call returns sums over the scan kernel's launches of
        # one batch; per launch both are divided by the launch count (the ratio is the same)
        launches = max(1, st["last_scan_launches"]) if st["last_path"] == 1 else 1
        kernel = "k_leafscan_f16" if st["last_path"] == 1 else "k_rerank"
        rr = float(np.mean(rr_ms)) / launches
        rr_bytes = rr_bytes / launches
        achieved = rr_bytes / (rr * 1e-3) / 1e9
        out = {
            "metric": "queries/sec @ recall@10, GloVe-100d 1M base / 10k query batch",
            "value": nq * args.steps / dt, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %d-d float32, %d base / %d queries, %d trees, %d partition bits, "
                                   "bucket overflow %d, %d-step search, top-%d by %s" % (
                                       args.workload, X.shape[1], n, nq, ix.L, ix.pb, params["bucket_overflow"],
                                       steps, k, args.metric),
                       "global_batch": nq, "parallelism": ("tables: %d query groups x %d table groups" % (sharded.qg, sharded.tg)) if sharded.mode == "tables"
                       else "%s%d" % (sharded.mode, world),
                       "recall_at_10": recall, "fit_seconds": round(fit_s, 3),
                       "fit_ms_on_rank0": {"copy_in_and_hash": round(st["fit_ms_hash"], 1),
                                           "forest_build_on_device": round(st["fit_ms_forest"], 1)},
                       "rerank_path": "bucket-major fp16 scan + exact fp64 rerank" if st["last_path"] == 1 else "gather",
                       "ids_visited_per_query_on_rank0": st["last_raw_candidates"] / nq,
                       "unique_candidates_per_query_on_rank0": (st["last_unique_candidates"] / nq) if st["last_path"] == 0 else None,
                       "bucket_incidences_per_query_on_rank0": st["last_pairs"] / nq,
                       "forwarded_to_exact_rerank_per_query_on_rank0": st["last_forwarded"] / nq,
                       "misflag_events": st["misflag_events"], "short_hash_events": st["short_hash_events"]},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.workload, n, nq, world, kernel, steps),
                         "algorithmic_bytes_per_launch": rr_bytes, "ms_per_launch": rr,
                         "launches_per_batch": launches},
        }
        if gather is not None:
            out["roofline_gather_rerank"] = gather
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, X, Q, pick, steps, k, args.metric)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
