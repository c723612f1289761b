"""One rank of the multi-process check of the sharded device stage (started by tests/test_a_dist_gpu.py through
torch.distributed.run, backend gloo, every rank on GPU 0): ShardedDpf.query_topk in "rows" and "tables" mode, on
several grids and with a batch size the query groups do not divide, against the unsharded CPU oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import oracle as O
    from similaritysearchbyrdf_amd.dist import ShardedDpf
    from tests.common import make_problem, assert_topk_equal
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    params, X, Q, pick = make_problem(n=9000, nq=67, bucket_overflow=50, table_num=3, permutation_num=2, n_clusters=15)
    n, nq, k, steps = X.shape[0], Q.shape[0], 10, 1
    pick = pick.astype(np.int32)
    pick[:4] = [0, 5, 127, 128]
    want = {}
    if rank == 0:
        orc = O.OracleIndex(params)
        orc.fit(X.astype(np.float64), threads=4)
        Q64 = Q.astype(np.float64)
        off, ids = orc.query_batch(pick, Q64, steps=steps, threads=4)
        for metric in ("l2", "dot"):
            want[metric] = orc.topk(Q64, off, ids, k=k, metric=metric, threads=4)
    dQ = torch.from_numpy(Q).to(dev)
    dqid = torch.from_numpy(pick).to(dev)
    grids = {2: [(1, 2), (2, 1)], 4: [(1, 4), (2, 2), (4, 1)]}[world]
    checked = 0
    for mode in ("rows", "tables", "subindexes"):
        for grid in grids:
            for metric in ("l2", "dot"):
                sh = ShardedDpf(params, n, metric=metric, top_k=k, dtype=np.float32, device=0, rank=rank, world=world,
                                mode=mode, grid=grid)
                sh.fit(X)
                for m in (nq, 33):                     # 67 and 33 queries: a short last query group
                    out_i = torch.full((m, k), -7, dtype=torch.int32, device=dev)
                    out_s = torch.full((m, k), -7.0, dtype=torch.float64, device=dev)
                    sh.query_topk(dqid[:m], dQ[:m], steps, k, out_i, out_s, torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    gi, gs = out_i.cpu().numpy(), out_s.cpu().numpy()
                    # every rank holds the merged answer; rank 0 owns the oracle's
                    both = [None] * world
                    dist.all_gather_object(both, (gi.tobytes(), gs.tobytes()))
                    assert all(b == both[0] for b in both), "ranks disagree on the merged answer"
                    if rank == 0:
                        wi, ws = want[metric]
                        assert_topk_equal(gi, gs, wi[:m], ws[:m])
                        checked += 1
                sh.index.close()
    if rank == 0:
        print("dist check ok: world %d, %d sharded batches equal the unsharded oracle" % (world, checked), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
