"""The entry point that allocates nothing (dpf_query_topk_device_static: every per-batch buffer is a slice
of one array in the library's device image) against dpf_query_topk_device, through the C ABI: same ids and
scores bit for bit, on two calls in a row, the second with a smaller input; limits refused with
DPF_ECAPACITY."""
import numpy as np
import pytest

from similaritysearchbyrdf_amd._lib import DpfIndex, DpfError, DPF_ECAPACITY
from tests.common import make_problem

pytestmark = pytest.mark.gpu


def _device_call(fn, dev, pick, Q, steps, k, with_qids=True):
    import torch
    nq = Q.shape[0]
    dq = torch.from_numpy(np.ascontiguousarray(Q)).to(dev)
    dqid = torch.from_numpy(np.ascontiguousarray(pick).astype(np.int32)).to(dev)
    di = torch.full((nq, k), -7, dtype=torch.int32, device=dev)
    ds = torch.full((nq, k), -7.0, dtype=torch.float64, device=dev)
    fn(dqid.data_ptr() if with_qids else None, dq.data_ptr(), nq, steps, k, di.data_ptr(), ds.data_ptr(),
       torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return di.cpu().numpy(), ds.cpu().numpy()


@pytest.mark.parametrize("dim,dtype,metric", [(100, np.float32, "l2"), (50, np.float32, "dot"), (33, np.float64, "l2")])
def test_static_scratch_entry_point_equals_the_allocating_one(dim, dtype, metric):
    import torch
    dev = torch.device("cuda:0")
    # rows of 50 floats / 33 doubles are no multiple of 16 bytes: the padded query copies live in the scratch too
    params, X, Q, pick = make_problem(n=20000, nq=300, dim=dim, dtype=dtype, bucket_overflow=80, table_num=3,
                                      permutation_num=2, n_clusters=12)
    ix = DpfIndex(params, metric=metric, dtype=dtype)
    ix.fit(X)
    pick = pick.astype(np.int32)
    pick[:4] = [0, 5, 127, 128]                      # self-exclusion on both sides of the boxed-Integer cache
    for nq, steps, k in ((300, 1, 10), (37, 2, 10), (300, 0, 64), (1, 3, 1)):    # the second call is smaller than the first
        want = _device_call(ix.query_topk_device, dev, pick[:nq], Q[:nq], steps, k)
        got = _device_call(ix.query_topk_device_static, dev, pick[:nq], Q[:nq], steps, k)
        assert np.array_equal(got[0], want[0]), (nq, steps, k)
        assert np.array_equal(got[1], want[1], equal_nan=True), (nq, steps, k)
    # without query ids (no self-exclusion)
    want = _device_call(ix.query_topk_device, dev, pick[:50], Q[:50], 1, 10, with_qids=False)
    got = _device_call(ix.query_topk_device_static, dev, pick[:50], Q[:50], 1, 10, with_qids=False)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True)
    # a second handle on the same GPU shares the array; calls one after the other do not disturb each other
    other = DpfIndex(params, metric=metric, dtype=dtype)
    other.fit(X[:5000])
    o_want = _device_call(other.query_topk_device, dev, pick[:64], Q[:64], 1, 10)
    o_got = _device_call(other.query_topk_device_static, dev, pick[:64], Q[:64], 1, 10)
    assert np.array_equal(o_got[0], o_want[0]) and np.array_equal(o_got[1], o_want[1], equal_nan=True)
    got = _device_call(ix.query_topk_device_static, dev, pick[:37], Q[:37], 2, 10)
    want = _device_call(ix.query_topk_device, dev, pick[:37], Q[:37], 2, 10)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True)


def test_static_scratch_limits_are_refused():
    import torch
    dev = torch.device("cuda:0")
    params, X, Q, pick = make_problem(n=30000, nq=1100, bucket_overflow=500, table_num=4, permutation_num=3, n_clusters=3)
    ix = DpfIndex(params, metric="l2")
    ix.fit(X)
    with pytest.raises(DpfError) as ei:              # more than 1024 queries
        _device_call(ix.query_topk_device_static, dev, pick, Q, 1, 10)
    assert ei.value.code == DPF_ECAPACITY
    # 1000 queries whose candidate lists (up to min(ids visited, 30000) each) exceed the 6M-id arena
    want = _device_call(ix.query_topk_device, dev, pick[:1000], Q[:1000], 3, 10)
    raw = ix.stats()["last_raw_candidates"]
    try:
        got = _device_call(ix.query_topk_device_static, dev, pick[:1000], Q[:1000], 3, 10)
        assert raw <= 40_000_000                     # it can only have fitted if the lists were short
        assert np.array_equal(got[0], want[0])
    except DpfError as e:
        assert e.code == DPF_ECAPACITY and "arena" in str(e)
    # and the handle still works afterwards
    got = _device_call(ix.query_topk_device_static, dev, pick[:20], Q[:20], 1, 10)
    want = _device_call(ix.query_topk_device, dev, pick[:20], Q[:20], 1, 10)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1], equal_nan=True)


def test_static_call_leaves_the_gpus_free_memory_unchanged():
    """hipMemGetInfo before and after the FIRST call of a fresh handle through the entry point that allocates nothing
    (the kernels it launches were loaded by another handle's gather path before the first reading)."""
    import torch
    dev = torch.device("cuda:0")
    params, X, Q, pick = make_problem(n=20000, nq=256, dim=100, bucket_overflow=80, table_num=3, permutation_num=2, n_clusters=12)
    pick = pick.astype(np.int32)
    warm = DpfIndex(dict(params, flags=2), metric="l2")                # gather-only handle: same kernels as the static path
    warm.fit(X)
    want = _device_call(warm.query_topk_device, dev, pick, Q, 1, 10)
    _device_call(warm.query_topk_device_static, dev, pick[:8], Q[:8], 1, 10)
    ix = DpfIndex(params, metric="l2")
    ix.fit(X)
    nq, k = 256, 10
    dq = torch.from_numpy(np.ascontiguousarray(Q)).to(dev)
    dqid = torch.from_numpy(pick).to(dev)
    di = torch.full((nq, k), -7, dtype=torch.int32, device=dev)
    ds = torch.full((nq, k), -7.0, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info(dev)
    ix.query_topk_device_static(dqid.data_ptr(), dq.data_ptr(), nq, 1, k, di.data_ptr(), ds.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info(dev)
    assert free_after == free_before, (free_before, free_after)
    assert np.array_equal(di.cpu().numpy(), want[0]) and np.array_equal(ds.cpu().numpy(), want[1], equal_nan=True)
    # the allocating entry point, for contrast, does grow the handle's workspaces on its first call
    ix.query_topk_device(dqid.data_ptr(), dq.data_ptr(), nq, 1, k, di.data_ptr(), ds.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(dev)[0] < free_before


def test_static_calls_back_to_back_on_two_streams_and_handles():
    """Two handles share the GPU's scratch array.  Calls issued one after the other on DIFFERENT streams with no
    synchronisation in between: the library orders the second call's kernels behind the first call's last kernel."""
    import torch
    dev = torch.device("cuda:0")
    params, X, Q, pick = make_problem(n=30000, nq=600, dim=100, bucket_overflow=80, table_num=3, permutation_num=2, n_clusters=12)
    pick = pick.astype(np.int32)
    a = DpfIndex(params, metric="l2")
    a.fit(X)
    b = DpfIndex(params, metric="dot")
    b.fit(X[:12000])
    want_a = _device_call(a.query_topk_device, dev, pick[:600], Q[:600], 2, 10)
    want_b = _device_call(b.query_topk_device, dev, pick[:450], Q[:450], 3, 10)
    dq = torch.from_numpy(np.ascontiguousarray(Q)).to(dev)
    dqid = torch.from_numpy(pick).to(dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    outs = []
    for r in range(6):
        outs.append((torch.full((600, 10), -7, dtype=torch.int32, device=dev), torch.full((600, 10), -7.0, dtype=torch.float64, device=dev),
                     torch.full((450, 10), -7, dtype=torch.int32, device=dev), torch.full((450, 10), -7.0, dtype=torch.float64, device=dev)))
    torch.cuda.synchronize()
    for r in range(6):
        ai, as_, bi, bs = outs[r]
        a.query_topk_device_static(dqid.data_ptr(), dq.data_ptr(), 600, 2, 10, ai.data_ptr(), as_.data_ptr(), s1.cuda_stream)
        b.query_topk_device_static(dqid.data_ptr(), dq.data_ptr(), 450, 3, 10, bi.data_ptr(), bs.data_ptr(), s2.cuda_stream)
    torch.cuda.synchronize()
    for ai, as_, bi, bs in outs:
        assert np.array_equal(ai.cpu().numpy(), want_a[0]) and np.array_equal(as_.cpu().numpy(), want_a[1], equal_nan=True)
        assert np.array_equal(bi.cpu().numpy(), want_b[0]) and np.array_equal(bs.cpu().numpy(), want_b[1], equal_nan=True)


def test_static_calls_from_two_threads_are_refused_or_right():
    """Two host threads, a handle and a stream each, calling at the same time: a call either gets the GPU's scratch
    array to itself and answers exactly, or is refused with DPF_EBUSY.  Never a wrong answer."""
    import threading
    import torch
    from similaritysearchbyrdf_amd._lib import DPF_EBUSY
    dev = torch.device("cuda:0")
    params, X, Q, pick = make_problem(n=30000, nq=500, dim=100, bucket_overflow=80, table_num=3, permutation_num=2, n_clusters=12)
    pick = pick.astype(np.int32)
    handles = [DpfIndex(params, metric="l2"), DpfIndex(params, metric="l2")]
    for h in handles:
        h.fit(X)
    want = _device_call(handles[0].query_topk_device, dev, pick, Q, 2, 10)
    dq = torch.from_numpy(np.ascontiguousarray(Q)).to(dev)
    dqid = torch.from_numpy(pick).to(dev)
    torch.cuda.synchronize()
    tally = [{"ok": 0, "busy": 0, "bad": 0, "err": []} for _ in range(2)]

    def worker(w):
        torch.cuda.set_device(0)
        st = torch.cuda.Stream(device=dev)
        di = torch.empty((500, 10), dtype=torch.int32, device=dev)
        ds = torch.empty((500, 10), dtype=torch.float64, device=dev)
        for _ in range(40):
            with torch.cuda.stream(st):
                di.fill_(-7)
            try:
                handles[w].query_topk_device_static(dqid.data_ptr(), dq.data_ptr(), 500, 2, 10, di.data_ptr(), ds.data_ptr(), st.cuda_stream)
            except DpfError as e:
                if e.code == DPF_EBUSY:
                    tally[w]["busy"] += 1
                else:
                    tally[w]["err"].append(str(e))
                continue
            st.synchronize()
            good = np.array_equal(di.cpu().numpy(), want[0]) and np.array_equal(ds.cpu().numpy(), want[1], equal_nan=True)
            tally[w]["ok" if good else "bad"] += 1

    th = [threading.Thread(target=worker, args=(w,)) for w in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not tally[0]["err"] and not tally[1]["err"], tally
    assert tally[0]["bad"] == 0 and tally[1]["bad"] == 0, tally
    assert tally[0]["ok"] + tally[1]["ok"] > 0, tally
