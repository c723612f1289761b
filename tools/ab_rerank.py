import sys, time, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bench import workload
from similaritysearchbyrdf_amd._lib import DpfIndex
params, X, Q, pick, steps = workload(sys.argv[1] if len(sys.argv) > 1 else "c3", 1_000_000, 10_000)
for flags in ((0, 2) if len(sys.argv) < 3 else (int(sys.argv[2]),)):
    ix = DpfIndex(dict(params, flags=flags), metric="l2")
    ix.fit(X)
    dev = torch.device("cuda:0")
    dQ = torch.from_numpy(Q).to(dev); dq = torch.from_numpy(pick.astype(np.int32)).to(dev)
    di = torch.empty((10000, 10), dtype=torch.int32, device=dev); ds = torch.empty((10000, 10), dtype=torch.float64, device=dev)
    ix.set_profiling(True)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ix.query_topk_device(dq.data_ptr(), dQ.data_ptr(), 10000, steps, 10, di.data_ptr(), ds.data_ptr(), None)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = ix.stats()
    print("flags", flags, "ms/batch %.2f" % (dt * 1e3), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if k.startswith("last")})
    if flags == 0: ref = di.cpu().numpy().copy()
    else: print("same ids:", np.array_equal(ref, di.cpu().numpy()))
    ix.close()
