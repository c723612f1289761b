"""The sharded query with the real device stage: two ranks, each a fresh process on GPU 0, exchanging over gloo
(torch.distributed.run), ShardedDpf.query_topk in both modes against the unsharded oracle (tests/dist_rank_main.py).

This file sorts first on purpose: the ranks must be started by a parent that has not initialised the GPU itself
(a process that has may not start GPU programs on the test machines), and the other GPU tests do initialise it.
RCCL with more than one rank is not exercised here (one GPU): the exchange is the same all-gather on gloo."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_sharded_device_stage_two_ranks_on_one_gpu():
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: run this file first or on its own")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    port = 23000 + os.getpid() % 4000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_rank_main.py")]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = res.stdout.decode(errors="replace")
    assert res.returncode == 0, text[-4000:]
    assert "dist check ok: world 2, 24 sharded batches" in text, text[-2000:]


def test_rccl_collectives_single_rank():
    # the RCCL backend with the one rank a one-GPU machine allows: the exchange's all-gather on uint8 blocks, the
    # timing all-reduce, barriers, and the sharded query through exchange + merge kernel (tests/rccl_rank_main.py)
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("the GPU is already initialised in this process: run this file first or on its own")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    port = 27000 + os.getpid() % 4000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "rccl_rank_main.py")]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = res.stdout.decode(errors="replace")
    assert res.returncode == 0, text[-4000:]
    assert "rccl single-rank check ok" in text, text[-2000:]
