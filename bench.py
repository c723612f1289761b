#!/usr/bin/env python3
"""Batch-query benchmark of the MI355X Dynamic Partition Forest engine.

One "step" = one pass of the hot path (LSH projection -> forest probe ->
candidate rerank/top-k) over one batch of `nq` synthetic queries that is
already resident in HBM; the timed loop rotates through --batches distinct
batches (default 4), so no step repeats the one before it.  The same batches
are then timed once more through the host-pointer entry point (host query
matrix in, host ids + scores out: `value_host_to_host`).  Prints ONE JSON
line on rank 0.

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

With N > 1 (--mode): "rows" = the forest replicated, base rows sharded over row groups and the
query batch over query groups (a rank scores the candidates of its query group that fall in its
row shard); "tables" = everything replicated, query groups x table groups (a rank answers its
query group from its range of the trees).  Either way the per-rank top-k lists are all-gathered
(RCCL, one collective) and merged.  Total work is fixed, so scaling is "strong".
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


QUERY_RECIPE = {"c2": "clustered", "c3": "clustered", "c4": "sift", "c5": "unit"}


def pmc_record(workload_name, n, nq, world, kernel, steps):
    """The committed rocprofv3 PMC record (tools/pmc_traffic.py) of this kernel on this configuration: HBM-side
    bytes per launch (FETCH_SIZE doubled as gfx950 needs, plus WRITE_SIZE) and, when the pass was made, the
    matrix-core counters; None when no committed pass matches."""
    import glob
    if world != 1:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_%s_pmc_traffic.json" % (workload_name, kernel))), reverse=True):
        with open(path) as f:
            rec = json.load(f)
        if (rec.get("search_steps", 1) == steps and rec.get("base_rows", 1_000_000) == n and rec.get("queries", 10_000) == nq):
            rec["file"] = os.path.relpath(path, ROOT)
            return rec
    return None


def ground_truth_l2(torch, dX, dQ, k, chunk=500):
    """Exact top-k by squared L2 distance (fp32 GEMM form); only used to report recall."""
    xn = (dX * dX).sum(dim=1)
    out = []
    for s in range(0, dQ.shape[0], chunk):
        q = dQ[s:s + chunk]
        d = xn[None, :] - 2.0 * (q @ dX.T)
        out.append(torch.topk(d, k, dim=1, largest=False).indices)
    return torch.cat(out).cpu().numpy()


def peak_rss_mb():
    import resource
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(params, X, Q, pick, steps, k, metric, target_s=12.0):
    """The CPU restatement of the reference path, reference threading (5 threads, each
    owning a table range, DensevectorRDFInit.scala:344-350; single-threaded rerank,
    DensevectorRDFInit.scala:477-505), on a bounded sample of the same batch.  Also returns the
    oracle's top-k ids of the sampled queries, for the parity line of the benchmark."""
    from oracle import oracle as O
    cores = os.cpu_count() or 8
    fit_threads = max(1, min(16, cores))
    t0 = time.time()
    orc = O.OracleIndex(params)
    orc.fit(X.astype(np.float64), threads=fit_threads)
    fit_s = time.time() - t0
    Q64 = Q.astype(np.float64)
    last = {}

    def run(m, threads, rerank_threads):
        t = time.time()
        off, ids = orc.query_batch(pick[:m], Q64[:m], steps=steps, threads=threads, mode=0 if threads == 5 else 1)
        last["ids"], last["scores"] = orc.topk(Q64[:m], off, ids, k=k, metric=metric, threads=rerank_threads)
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
           "host_cores": cores, "cpu_model": cpu_model(),
           "threads_note": "value: 5 threads (the reference's queryThreadNum); all_cores_value: min(16, host_cores) threads - "
                           "a one-GPU box is given 16 of the host's cores",
           "misflag_events": orc.misflag_events}
    orc.close()
    return res, last["ids"], last["scores"]


def roofline_of(ix, run_batch, nbatches, repeats, n, nq, world, workload_name, steps):
    """Per-launch roofline record of the dominant kernel: HIP-event time and the kernel's own algorithmic
    bytes (dpf_last_rerank_profile), summed over `repeats` rounds of the rotating batches."""
    ix.set_profiling(True)
    tot_ms, tot_bytes, launches_total = 0.0, 0, 0
    for i in range(repeats * nbatches):
        run_batch(i % nbatches)
        ms, by = ix.last_rerank_profile()
        st = ix.stats()
        launches = max(1, st["last_scan_launches"]) if st["last_path"] == 1 else 1
        tot_ms += ms
        tot_bytes += by
        launches_total += launches
    ix.set_profiling(False)
    kernel = "k_rerank" if st["last_path"] != 1 else "k_leafscan_i8" if st["last_scan_int8"] == 1 else "k_leafscan_f16"
    rr = tot_ms / launches_total
    rr_bytes = tot_bytes / launches_total
    achieved = rr_bytes / (rr * 1e-3) / 1e9
    rec = pmc_record(workload_name, n, nq, world, kernel, steps)
    out = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": rec.get("traffic_bytes_per_launch") if rec else None,
           "algorithmic_bytes_per_launch": rr_bytes, "ms_per_launch": rr,
           "launches_per_batch": launches_total / float(repeats * nbatches)}
    if st["last_path"] == 1:
        int8 = st["last_scan_int8"] == 1
        table_mb = (st["base8_bytes"] if int8 else st["base16_bytes"]) / 1e6
        out["bound_note"] = ("bytes cross the L2 <-> Infinity Fabric boundary; the %s table the scan re-reads is %.0f MB "
                             "against 256 MiB of Infinity Cache, so %s of these bytes can be served from that cache rather "
                             "than from HBM" % ("int8" if int8 else "fp16", table_mb, "most" if table_mb <= 270 else "only a fraction"))
        out["scan_arithmetic"] = "int8 products, int32 sums (v_mfma_i32_16x16x64_i8)" if int8 else "fp16 products, fp32 sums (v_mfma_f32_16x16x32_f16)"
        out["int8_table_mb" if int8 else "fp16_table_mb"] = table_mb
        out["tile_rows"] = st["tile_rows"]
        out["rows_read"] = "from a bucket-order copy of the fp16 rows (%.0f MB)" % (st["bucket_order_bytes"] / 1e6) if st["bucket_order_bytes"] else "gathered by id"
        out["mean_tile_occupancy"] = st["resident_ids"] / float(max(1, st["num_tiles"]) * max(1, st["tile_rows"]))
    if rec:
        out["traffic_source"] = rec["file"]
        for key in ("mfma_flops_per_launch", "mfma_busy_cycles_per_launch", "mfma_tflops", "mfma_util"):
            if key in rec:
                out[key] = rec[key]
    return out, st


OOS_RECIPE = {"c2": "clustered", "c3": "clustered", "c4": "sift", "c5": "unit"}


def out_of_sample_record(torch, ix, args, X, nq, k, steps, dev, d_ids, d_sc):
    """The timed index queried with batches that are NOT noisy copies of base rows: fresh draws from the distribution
    the base was drawn from.  The first radii of such queries are wider, so more rows pass the filter."""
    from similaritysearchbyrdf_amd import synthetic
    nb = max(1, args.batches)
    batches = []
    for b in range(nb):
        qb, ib = synthetic.fresh_queries(OOS_RECIPE[args.workload], X, nq, seed=7000 + b)
        batches.append((torch.from_numpy(qb).to(dev), torch.from_numpy(ib).to(dev)))
    stream = torch.cuda.current_stream().cuda_stream

    def run(b):
        ix.query_topk_device(batches[b][1].data_ptr(), batches[b][0].data_ptr(), nq, steps, k, d_ids.data_ptr(), d_sc.data_ptr(), stream)
    for b in range(nb + 1):
        run(b % nb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = max(nb, min(args.steps, 12))
    for i in range(reps):
        run(i % nb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    run(0)
    torch.cuda.synchronize()
    st = ix.stats()
    rec = {"queries": "fresh draws from the base's distribution (%s), ids -1: no query is a stored row" % OOS_RECIPE[args.workload],
           "ms_per_step": dt * 1e3, "queries_per_s": nq / dt,
           "forwarded_to_exact_rerank_per_query": st["last_forwarded"] / nq,
           "bucket_incidences_per_query": st["last_pairs"] / nq, "ids_visited_per_query": st["last_raw_candidates"] / nq,
           "overflowed_queries_last_batch": st["last_overflowed"], "rerank_path": "bucket-major" if st["last_path"] == 1 else "gather"}
    if not args.no_recall and args.metric == "l2":
        ids = d_ids.cpu().numpy()
        dX = torch.from_numpy(X).to(dev)
        gt = ground_truth_l2(torch, dX, batches[0][0], k)
        del dX
        rec["recall_at_%d" % k] = sum(len(np.intersect1d(ids[q][ids[q] >= 0], gt[q])) for q in range(nq)) / float(nq * k)
    return rec


def c1_record(args, k):
    """BASELINE configs[0], the reference's README "simple test" (README.md:32-43) at its stated size: 20 000 100-d
    vectors (GloVe is not available offline: the c2 / c3 generator), the default forest (10 x 3 trees, 3 partition bits,
    bucket overflow 500), two queries with components uniform in [0, 1), ids 0 and 1, NewMultiThreadQueryBatch with
    0 steps.  Candidate Sets and top-k of both queries against the CPU oracle."""
    from oracle import oracle as O
    from similaritysearchbyrdf_amd import synthetic
    from similaritysearchbyrdf_amd._lib import DpfIndex
    params = synthetic.make_family(100, table_num=10, permutation_num=3, chain_length=32, family="angle",
                                   family_size=100, partition_bits=3, orthogonal=True)
    params.update(bucket_overflow=500, dir_node_size=32, bucket_bits=28, exclude_self="compat")
    X, _, _ = synthetic.make_clustered(20_000, 100, 2)
    Q, _ = synthetic.fresh_queries("uniform", X, 2, seed=5)
    qids = np.array([0, 1], dtype=np.int32)
    gi = DpfIndex(params, metric="dot", top_k=k, dtype=np.float32, device=0)
    t0 = time.perf_counter()
    gi.fit(X)
    fit_s = time.perf_counter() - t0
    orc = O.OracleIndex(params)
    orc.fit(X.astype(np.float64))
    rec = {"workload": "c1: 20000 x 100-d, 30 trees, 2 uniform-[0,1) queries (README simple test)", "fit_seconds": round(fit_s, 4), "steps": {}}
    same_all = True
    for st_ in (0, 1, 2, 3):
        off, ids = gi.query_batch(qids, Q, steps=st_)
        t0 = time.perf_counter()
        for _ in range(5):
            gi.query_batch(qids, Q, steps=st_)
        ms = (time.perf_counter() - t0) / 5 * 1e3
        woff, wids = orc.query_batch(qids, Q.astype(np.float64), steps=st_)
        same_sets = bool(np.array_equal(off, woff) and np.array_equal(ids, wids))
        ti, ts = gi.query_topk(qids, Q, steps=st_, k=k)
        wi, ws = orc.topk(Q.astype(np.float64), woff, wids, k=k, metric="dot")
        same_topk = bool(np.array_equal(ti, wi))
        same_all = same_all and same_sets and same_topk
        rec["steps"][str(st_)] = {"set_sizes": [int(off[1] - off[0]), int(off[2] - off[1])], "same_sets_as_oracle": same_sets,
                                  "same_top%d_as_oracle" % k: same_topk, "query_batch_ms": ms}
    rec["oracle_check"] = same_all
    orc.close()
    gi.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--base-rows", dest="n", type=int, default=1_000_000)
    ap.add_argument("--queries", dest="nq", type=int, default=10_000)
    ap.add_argument("--dim", type=int, default=None, help="c5 only: dimensions of the vectors (default 768)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--metric", default="l2", choices=["l2", "dot"])
    ap.add_argument("--search-steps", type=int, default=None)
    ap.add_argument("--batches", type=int, default=4, help="distinct query batches the timed loop rotates through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-host-to-host", action="store_true")
    ap.add_argument("--no-out-of-sample", action="store_true",
                    help="skip the record of the same index queried with fresh draws from the base's distribution")
    ap.add_argument("--no-gather-check", action="store_true",
                    help="skip the untimed run of the same batch through the per-query gather rerank (its candidate lists need "
                         "4 bytes per visited id: too much at 10M base / 100k queries)")
    ap.add_argument("--in-flight", type=int, default=1, choices=[1, 2],
                    help="2: also measure the throughput with two batches in flight on one GPU (reported as value_two_in_flight)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="dpf_set_option on the handle before the batches, e.g. --option scan_passes=3 (repeatable)")
    ap.add_argument("--full-filter", action="store_true",
                    help="fp16 copies of every dimension (DPF_FLAG_FULL_FILTER) instead of the leading ones: A/B of the truncated filter")
    ap.add_argument("--bucket-order-rows", default="auto", choices=["auto", "on", "off"],
                    help="whether the scan reads a second fp16 copy of the rows laid out in bucket order (DPF_FLAG_[NO_]BUCKET_ORDER_ROWS); "
                         "auto: the library's rule (line-straddling rows, a table beyond the Infinity Cache, room for the copy)")
    ap.add_argument("--no-int8", action="store_true",
                    help="never scan in int8 (DPF_FLAG_NO_INT8): A/B of the exact integer scan on integer-valued bases (c4)")
    ap.add_argument("--large-table-rows", type=int, default=4_000_000,
                    help="c3 on one GPU: base rows of the extra roofline run whose fp16 table (256 B per row) is several "
                         "times the Infinity Cache; 0 = skip")
    ap.add_argument("--rerank", default="auto", choices=["auto", "gather"],
                    help="auto: bucket-major fp16 scan + exact rerank where it applies; gather: per-query fp64 gather kernel")
    ap.add_argument("--stream-fit", action="store_true",
                    help="c5 only: the base is never held on the host - every rank generates it chunk by chunk (chunk-seeded "
                         "recipe: any rank can produce any row range), hashes each chunk on the device and keeps the rows of its "
                         "shard; the forest is built once after the last chunk.  Skips what needs the whole base on the host "
                         "(recall, CPU baseline, gather check)")
    ap.add_argument("--verify", action="store_true", help="rank 0 also runs an unsharded index and compares the merged ids")
    ap.add_argument("--mode", default="auto", choices=["auto", "rows", "tables"],
                    help="N > 1: rows = base rows x query batch sharded (forest replicated); tables = trees x query batch "
                         "sharded (everything replicated); auto = tables when the forest has at least N trees, else rows")
    ap.add_argument("--grid", default=None, help="QxS: query groups x shard groups (row or table groups; default: chosen from --gpus)")
    ap.add_argument("--backend", default="nccl", help="nccl (RCCL, one GPU per rank) or gloo (rehearsal: ranks may share a GPU)")
    args = ap.parse_args()

    import torch
    from similaritysearchbyrdf_amd import synthetic
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

    stream_fit = args.stream_fit
    if stream_fit:
        if args.workload != "c5":
            raise SystemExit("--stream-fit is the bring-up of c5-sized bases (the other workloads fit the host easily)")
        d_ = args.dim or 768
        params = synthetic.make_family(d_, table_num=10, permutation_num=3, chain_length=32, family="angle",
                                       family_size=d_, partition_bits=3, orthogonal=True)
        params.update(bucket_overflow=500, dir_node_size=32, bucket_bits=28, exclude_self="compat")
        X, steps = None, 1
        q_batches = synthetic.unit_gaussian_queries(args.n, d_, args.nq, batches=max(1, args.batches))
        if max(1, args.batches) == 1:
            q_batches = [q_batches]
        Q, pick = q_batches[0]
        args.no_recall = args.no_cpu_baseline = args.no_gather_check = args.no_out_of_sample = True
    else:
        params, X, Q, pick, steps = workload(args.workload, args.n, args.nq, args.dim)
    if args.rerank == "gather":
        params["flags"] = 2          # DPF_FLAG_GATHER_ONLY
    if args.full_filter:
        params["flags"] = int(params.get("flags", 0)) | 8          # DPF_FLAG_FULL_FILTER
    if args.no_int8:
        params["flags"] = int(params.get("flags", 0)) | 16         # DPF_FLAG_NO_INT8
    if args.bucket_order_rows != "auto":
        params["flags"] = int(params.get("flags", 0)) | (64 if args.bucket_order_rows == "on" else 128)
    if args.search_steps is not None:
        steps = args.search_steps
    n, nq, k = (args.n if stream_fit else X.shape[0]), Q.shape[0], args.k
    dim_ = Q.shape[1]
    nb = max(1, args.batches)
    host_batches = [(Q, pick.astype(np.int32))]
    for b in range(1, nb):
        qb, pb_ = q_batches[b] if stream_fit else synthetic.more_queries(QUERY_RECIPE[args.workload], X, nq, seed=9000 + b)
        host_batches.append((qb, pb_.astype(np.int32)))
    grid = tuple(int(v) for v in args.grid.lower().split("x")) if args.grid else None
    sharded = ShardedDpf(params, n, metric=args.metric, top_k=k, dtype=np.float32, device=local, rank=rank, world=world,
                         mode=None if args.mode == "auto" else args.mode, grid=grid)
    ix = sharded.index
    for ov in args.option:
        name, _, value = ov.partition("=")
        ix.set_option(name, int(value))
    t0 = time.time()
    if stream_fit:
        sharded.fit_stream(synthetic.unit_gaussian_chunks(n, dim_), total_rows=n)
    else:
        sharded.fit(X)
    fit_s = time.time() - t0

    dev_batches = [(torch.from_numpy(q).to(dev), torch.from_numpy(p).to(dev)) for q, p in host_batches]
    d_ids = torch.empty((nq, k), dtype=torch.int32, device=dev)
    d_sc = torch.empty((nq, k), dtype=torch.float64, device=dev)

    def run_batch(b):
        # hash -> probe -> scan -> rerank on this rank's shard; with N > 1 one all-gather of the packed partial
        # top-k lists and the merge kernel
        dQ, dqid = dev_batches[b]
        sharded.query_topk(dqid, dQ, steps, k, d_ids, d_sc, torch.cuda.current_stream().cuda_stream)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        run_batch(i % nb)
    st_before = ix.stats()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_batch((args.warmup + i) % nb)
    fence()
    dt = time.perf_counter() - t0
    st_after = ix.stats()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    sizing = {key: st_after["total_" + key] - st_before["total_" + key]
              for key in ("topk_batches", "sized_on_host", "carried", "rerun", "overflowed", "gather_batches")}

    # the same batches from host memory to host memory (the host-pointer entry point: H2D of the query matrix and
    # the query ids, the batch, D2H of ids and scores)
    h2h = None
    if world == 1 and not args.no_host_to_host:
        from similaritysearchbyrdf_amd._lib import pinned_array

        def timed_h2h(batches, out):
            for b in range(min(2, nb)):
                ix.query_topk(batches[b][1], batches[b][0], steps=steps, k=k, out=out)
            t0_ = time.perf_counter()
            for i in range(args.steps):
                qb, pb_ = batches[i % nb]
                ix.query_topk(pb_, qb, steps=steps, k=k, out=out)
            return time.perf_counter() - t0_

        dth = timed_h2h(host_batches, None)
        # the same from / into pinned host memory (dpf_host_alloc): the copy engines move it from where it lies
        pin_batches = []
        for qb, pb_ in host_batches:
            pq, pp = pinned_array(qb.shape, qb.dtype), pinned_array(pb_.shape, np.int32)
            pq[...] = qb
            pp[...] = pb_
            pin_batches.append((pq, pp))
        pin_out = (pinned_array((nq, k), np.int32), pinned_array((nq, k), np.float64))
        dtp = timed_h2h(pin_batches, pin_out)
        del pin_batches, pin_out
        h2h = {"value": nq * args.steps / dth, "ms_per_step": dth / args.steps * 1e3,
               "h2d_d2h_and_sync_ms_per_step": (dth - dt) / args.steps * 1e3,
               "bytes_per_step": {"h2d": int(nq * dim_ * 4 + nq * 4), "d2h": int(nq * k * 12)},
               "host_memory": "pageable (numpy arrays handed to dpf_query_topk; staged through the handle's pinned area in 1 MiB pieces)",
               "pinned": {"value": nq * args.steps / dtp, "ms_per_step": dtp / args.steps * 1e3,
                          "h2d_d2h_and_sync_ms_per_step": (dtp - dt) / args.steps * 1e3,
                          "host_memory": "pinned (dpf_host_alloc) query matrix, ids and result arrays: asynchronous copies, no staging"}}

    # Two batches in flight: a second handle over the same base, each handle driven by its own host thread on its own
    # stream, the same rotation of batches split between them.  What a serving loop gains when the probe / threshold /
    # exact-rerank stages of one batch overlap the fabric-bound scan of the other.  Reported beside `value`, never as it.
    two = None
    if world == 1 and args.in_flight == 2 and not args.no_host_to_host:
        import threading
        from similaritysearchbyrdf_amd._lib import DpfIndex
        other = DpfIndex(params, metric=args.metric, top_k=k, dtype=np.float32, device=local)
        other.fit(X)
        handles = [ix, other]
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        outs = [(torch.empty_like(d_ids), torch.empty_like(d_sc)) for _ in range(2)]
        per = max(1, args.steps // 2)

        def worker(w, count):
            torch.cuda.set_device(local)
            with torch.cuda.stream(streams[w]):
                for i in range(count):
                    dQ_, dqid_ = dev_batches[(2 * i + w) % nb]
                    handles[w].query_topk_device(dqid_.data_ptr(), dQ_.data_ptr(), nq, steps, k, outs[w][0].data_ptr(),
                                                 outs[w][1].data_ptr(), streams[w].cuda_stream)
            streams[w].synchronize()

        for count in (2, per):                     # a short warm-up round, then the timed one
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(w, count)) for w in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
        two = {"value": 2 * per * nq / dt2, "ms_per_batch": dt2 / (2 * per) * 1e3, "batches": 2 * per,
               "what": "two handles over the same base, one host thread and stream each, batches alternating between them"}
        other.close()

    # dominant kernel: the rerank launch, timed with HIP events on its own stream
    roofline, st = roofline_of(ix, run_batch, nb, max(1, min(args.steps, 12) // nb), n, nq, world, args.workload, steps)
    run_batch(0)
    fence()
    st = ix.stats()
    final_ids = d_ids.cpu().numpy()
    final_sc = d_sc.cpu().numpy()
    dQ, dqid = dev_batches[0]

    if args.verify and rank == 0:
        from similaritysearchbyrdf_amd._lib import DpfIndex
        full = DpfIndex(params, metric=args.metric, top_k=k, dtype=np.float32, device=local)
        if stream_fit:
            full.fit_stream(synthetic.unit_gaussian_chunks(n, dim_), total_rows=n)
        else:
            full.fit(X)
        want_ids, _ = full.query_topk(host_batches[0][1], Q, steps=steps, k=k)
        if not np.array_equal(want_ids, final_ids):
            raise SystemExit("verify failed: merged sharded result differs from the unsharded index")
        print("verify ok: %d-way sharded result (%s) equals the unsharded index" % (world, sharded.mode), file=sys.stderr)
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
    if rank == 0 and world == 1 and st["last_path"] == 1 and not args.no_gather_check:
        # the per-query gather rerank on the same batch (untimed): its bytes are the classic rerank roofline
        # (every de-duplicated candidate row once per query), and it yields the de-duplicated candidate count
        from similaritysearchbyrdf_amd._lib import DpfIndex
        gi = DpfIndex(dict(params, flags=2), metric=args.metric, top_k=k, dtype=np.float32, device=local)
        gi.fit(X)
        gi.set_profiling(True)
        g_ms = []
        g_ids = torch.empty_like(d_ids)
        g_sc = torch.empty_like(d_sc)
        for _ in range(3):
            gi.query_topk_device(dqid.data_ptr(), dQ.data_ptr(), nq, steps, k, g_ids.data_ptr(), g_sc.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
            ms, by = gi.last_rerank_profile()
            g_ms.append(ms)
        gst = gi.stats()
        same = bool(np.array_equal(g_ids.cpu().numpy(), final_ids))
        grec = pmc_record(args.workload, n, nq, world, "k_rerank", steps)
        gather = {"kernel": "k_rerank", "bound": "hbm", "achieved": by / (float(np.mean(g_ms[1:])) * 1e-3) / 1e9,
                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": by / (float(np.mean(g_ms[1:])) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "traffic": grec.get("traffic_bytes_per_launch") if grec else None,
                  "algorithmic_bytes_per_launch": by, "ms_per_launch": float(np.mean(g_ms[1:])),
                  "unique_candidates_per_query": gst["last_unique_candidates"] / nq, "same_ids_as_timed_path": same}
        gi.close()

    sampled = None
    if rank == 0 and world == 1 and st["last_path"] == 1 and args.no_gather_check:
        # instead: a few queries of batch 0 through the gather path of the same handle (the entry point that takes that
        # path within fixed buffers), compared with what the timed path returned for them
        m = int(min(nq, 1024, max(1, 5_000_000 // max(1, int(st["last_raw_candidates"] / nq)))))
        s_ids = torch.empty((m, k), dtype=torch.int32, device=dev)
        s_sc = torch.empty((m, k), dtype=torch.float64, device=dev)
        ix.query_topk_device_static(dqid.data_ptr(), dQ.data_ptr(), m, steps, k, s_ids.data_ptr(), s_sc.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        sampled = {"queries": m, "same_ids": bool(np.array_equal(s_ids.cpu().numpy(), final_ids[:m])),
                   "same_scores": bool(np.array_equal(s_sc.cpu().numpy(), final_sc[:m], equal_nan=True)),
                   "what": "the first queries of batch 0 through the per-query gather path (exact fp64 scores of the whole "
                           "candidate Set) against the timed path's answers"}
        run_batch(0)                      # the statistics below describe the timed path again
        fence()
        st = ix.stats()

    large = None
    if (rank == 0 and world == 1 and args.workload == "c3" and n == 1_000_000 and args.large_table_rows > n
            and args.rerank == "auto" and st["last_path"] == 1):
        # the same forest shape over a base whose fp16 table is several times the Infinity Cache: what the scan
        # reaches when its row reads really come from HBM
        from similaritysearchbyrdf_amd._lib import DpfIndex
        nl_ = args.large_table_rows
        Xl, Ql, pl = synthetic.make_clustered(nl_, X.shape[1], nq)
        li = DpfIndex(params, metric=args.metric, top_k=k, dtype=np.float32, device=local)
        li.fit(Xl)
        lb = [(torch.from_numpy(Ql).to(dev), torch.from_numpy(pl.astype(np.int32)).to(dev))]
        for b in range(1, nb):
            qb, pb_ = synthetic.more_queries("clustered", Xl, nq, seed=9100 + b)
            lb.append((torch.from_numpy(qb).to(dev), torch.from_numpy(pb_.astype(np.int32)).to(dev)))
        l_ids = torch.empty_like(d_ids)
        l_sc = torch.empty_like(d_sc)

        def run_large(b):
            li.query_topk_device(lb[b][1].data_ptr(), lb[b][0].data_ptr(), nq, steps, k, l_ids.data_ptr(), l_sc.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
        for b in range(nb):
            run_large(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(2 * nb):
            run_large(i % nb)
        torch.cuda.synchronize()
        l_dt = (time.perf_counter() - t0) / (2 * nb)
        large, lst = roofline_of(li, run_large, nb, 2, nl_, nq, world, args.workload, steps)
        large["workload"] = "c3 forest over %d base rows (same generator)" % nl_
        large["queries_per_s"] = nq / l_dt
        large["ms_per_step"] = l_dt * 1e3
        large["forwarded_to_exact_rerank_per_query"] = lst["last_forwarded"] / nq
        li.close()
        del Xl, lb

    oos = None
    if rank == 0 and world == 1 and not args.no_out_of_sample:
        oos = out_of_sample_record(torch, ix, args, X, nq, k, steps, dev, d_ids, d_sc)
        run_batch(0)
        fence()
    c1 = None
    if rank == 0 and world == 1 and args.workload == "c3" and not args.no_cpu_baseline:
        c1 = c1_record(args, k)

    if rank == 0:
        out = {
            "metric": "queries/sec @ recall@10, GloVe-100d 1M base / 10k query batch",
            "value": nq * args.steps / dt, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d-d float32, %d base / %d queries, %d trees, %d partition bits, "
                                   "bucket overflow %d, %d-step search, top-%d by %s" % (
                                       args.workload, dim_, n, nq, ix.L, ix.pb, params["bucket_overflow"],
                                       steps, k, args.metric),
                       "arithmetic": "rows and queries float32 in HBM; hashing and the exact rerank in float64; "
                                     "the bucket-major filter in " + (
                                         "int8 products with int32 accumulation (matrix cores; rows and queries are integers "
                                         "inside one window of 256 values)" if st["last_scan_int8"] == 1 else
                                         "fp16 products with fp32 accumulation (matrix cores)"),
                       "global_batch": nq, "distinct_batches_in_timed_loop": nb,
                       "parallelism": "one GPU" if world == 1 else ("%s: %d query groups x %d %s groups" % (
                           sharded.mode, sharded.qg, sharded.sg, "table" if sharded.mode == "tables" else "row")),
                       "batches_in_timed_loop_by_list_sizing_on_rank0": sizing,
                       "recall_at_10": recall, "recall_measured_on": "batch 0 of the rotation",
                       "fit_seconds": round(fit_s, 3),
                       "fit": ("streamed: %d-row chunks generated, hashed and sharded one at a time, forest built once "
                               "(includes the host's generation of every chunk)" % synthetic.ROW_CHUNK) if stream_fit else "whole base from host memory",
                       "host_peak_rss_mb_on_rank0": round(peak_rss_mb(), 1),
                       "resident_base_mb_on_rank0": round(st["base_bytes"] / 1e6, 1),
                       "fit_ms_on_rank0": {"copy_in_and_hash": round(st["fit_ms_hash"], 1),
                                           "forest_build_on_device": round(st["fit_ms_forest"], 1)},
                       "rerank_path": ("bucket-major %s scan + exact fp64 rerank" % ("int8" if st["last_scan_int8"] == 1 else "fp16"))
                                      if st["last_path"] == 1 else "gather",
                       "ids_visited_per_query_on_rank0": st["last_raw_candidates"] / nq,
                       "unique_candidates_per_query_on_rank0": (st["last_unique_candidates"] / nq) if st["last_path"] == 0 else None,
                       "bucket_incidences_per_query_on_rank0": st["last_pairs"] / nq,
                       "forwarded_to_exact_rerank_per_query_on_rank0": st["last_forwarded"] / nq,
                       "longest_forward_list_on_rank0": st["last_longest_list"],
                       "misflag_events": st["misflag_events"], "short_hash_events": st["short_hash_events"]},
            "roofline": roofline,
        }
        out["value_device_resident"] = out["value"]
        out["value_note"] = ("value = query matrix and result arrays resident in HBM when the timed region starts; "
                             "value_host_to_host = host query matrix in, host ids + scores out (pageable memory), "
                             "value_host_to_host_pinned = the same through pinned host memory")
        if h2h is not None:
            out["value_host_to_host"] = h2h["value"]
            out["value_host_to_host_pinned"] = h2h["pinned"]["value"]
            out["host_to_host"] = h2h
        if two is not None:
            out["value_two_in_flight"] = two["value"]
            out["two_in_flight"] = two
        if gather is not None:
            out["roofline_gather_rerank"] = gather
        if sampled is not None:
            out["sampled_gather_check"] = sampled
        if large is not None:
            out["roofline_large_table"] = large
        if oos is not None:
            out["out_of_sample"] = oos
        if c1 is not None:
            out["c1"] = c1
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], o_ids, o_sc = cpu_baseline(params, X, Q, pick, steps, k, args.metric)
            m = o_ids.shape[0]
            ok_ids = bool(np.array_equal(o_ids, final_ids[:m]))
            valid = o_ids >= 0
            rel = np.abs(final_sc[:m][valid] - o_sc[valid]) / np.maximum(np.abs(o_sc[valid]), 1e-300) if ok_ids else None
            out["oracle_check"] = {"queries": int(m), "same_top%d_ids" % k: ok_ids,
                                   "max_rel_score_diff": (float(rel.max()) if rel is not None and rel.size else None),
                                   "what": "top-%d of the first %d queries of batch 0: the timed HIP path against the CPU oracle "
                                           "fitted on the same full base" % (k, m)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
