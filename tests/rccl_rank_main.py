"""A single rank on the RCCL backend (started by tests/test_a_dist_gpu.py through torch.distributed.run with one
process): the collectives the sharded query and the benchmark use - all_gather_into_tensor of the packed uint8
blocks, the float64 MAX all-reduce of the timing, barrier - and ShardedDpf.query_topk through the exchange and the
merge kernel, against the same handle's direct answer.  More than one RCCL rank needs more than one GPU."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from similaritysearchbyrdf_amd.dist import ShardedDpf, exchange_packed, packed_bytes
    from tests.common import make_problem
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == 1
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    block = torch.randint(0, 256, (packed_bytes(67, 10),), dtype=torch.uint8, device=dev)
    out = torch.zeros((1, block.numel()), dtype=torch.uint8, device=dev)
    got = exchange_packed(block, out)
    dist.barrier()
    torch.cuda.synchronize()
    assert got.data_ptr() == out.data_ptr() and torch.equal(out[0], block)
    tt = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert float(tt.item()) == 1.25

    params, X, Q, pick = make_problem(n=9000, nq=67, bucket_overflow=50, table_num=3, permutation_num=2, n_clusters=15)
    k, steps = 10, 1
    sh = ShardedDpf(params, X.shape[0], metric="l2", top_k=k, dtype=np.float32, device=0, rank=0, world=1)
    sh.fit(X)
    dQ = torch.from_numpy(Q).to(dev)
    dqid = torch.from_numpy(pick.astype(np.int32)).to(dev)
    answers = []
    for through_exchange in (False, True):
        sh.always_exchange = through_exchange
        out_i = torch.full((67, k), -7, dtype=torch.int32, device=dev)
        out_s = torch.full((67, k), -7.0, dtype=torch.float64, device=dev)
        for _ in range(3):
            sh.query_topk(dqid, dQ, steps, k, out_i, out_s, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        answers.append((out_i.cpu().numpy(), out_s.cpu().numpy()))
    assert np.array_equal(answers[0][0], answers[1][0]) and np.array_equal(answers[0][1], answers[1][1], equal_nan=True)
    assert (answers[0][0] >= 0).any()
    dist.barrier()
    dist.destroy_process_group()
    print("rccl single-rank check ok")


if __name__ == "__main__":
    main()
