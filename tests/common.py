"""Shared builders for the parity tests: one seeded problem, two engines."""
import numpy as np

from similaritysearchbyrdf_amd import synthetic


def make_problem(n=2000, nq=16, dim=100, family="angle", table_num=2, permutation_num=2,
                 partition_bits=3, bucket_overflow=500, dir_node_size=32, bucket_bits=28,
                 type_of_index="original", exclude_self="compat", dtype=np.float32, seed=1,
                 family_size=None, pstable_w=4, n_clusters=50, sigma=0.3, gaussian=False):
    if gaussian:
        params = synthetic.make_gaussian_family(dim, table_num * permutation_num, 32, partition_bits,
                                                seed=seed + 100, part_seed=seed + 200)
    else:
        params = synthetic.make_family(dim, table_num, permutation_num, 32, family,
                                       family_size or max(dim, 32), partition_bits,
                                       pstable_w=pstable_w, seed=seed + 100, part_seed=seed + 200)
    params.update(bucket_overflow=bucket_overflow, dir_node_size=dir_node_size, bucket_bits=bucket_bits,
                  type_of_index=type_of_index, exclude_self=exclude_self)
    if type_of_index == "sampling":
        from oracle import oracle as O
        import ctypes as C
        si = np.empty(32, dtype=np.int32)
        O.lib().orc_scala_shuffle_0_31(88387, si.ctypes.data_as(C.POINTER(C.c_int32)))
        params["sampling_index"] = si
    X, Q, pick = synthetic.make_clustered(n, dim, nq, n_clusters=n_clusters, sigma=sigma, seed=seed + 300,
                                          qseed=seed + 400, dtype=dtype)
    return params, X, Q, pick


def sets_of(offsets, ids):
    return [ids[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]


def assert_same_sets(got, want):
    go, gi = got
    wo, wi = want
    assert go.tolist() == wo.tolist(), "set sizes differ: %s vs %s" % (np.diff(go)[:8], np.diff(wo)[:8])
    assert np.array_equal(gi, wi)


def assert_topk_equal(got_ids, got_sc, want_ids, want_sc, rtol=1e-5):
    """ids exact; scores within rtol.  Where ids differ the two must be a tie of
    (near-)equal scores -- only summation order separates the two engines."""
    assert got_ids.shape == want_ids.shape
    pad_g, pad_w = got_ids < 0, want_ids < 0
    assert np.array_equal(pad_g, pad_w)
    ok = ~pad_w
    scale = np.maximum(np.abs(want_sc[ok]), 1e-300)
    assert np.all(np.abs(got_sc[ok] - want_sc[ok]) <= rtol * scale + 1e-12), "scores differ"
    diff = (got_ids != want_ids) & ok
    if diff.any():
        rel = np.abs(got_sc[diff] - want_sc[diff]) / np.maximum(np.abs(want_sc[diff]), 1e-300)
        assert np.all(rel < 1e-12), "top-k ids differ beyond an fp64 rounding tie"
        assert diff.mean() < 1e-3
