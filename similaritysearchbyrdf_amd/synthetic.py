"""Seeded synthetic inputs: hash families, partitioner functions, base/query sets.

The reference draws its hash families from the JVM's RNG and breeze
(AngleHashFamily.scala:37-85,121-149, PStableHashFamily.scala:37-77), which
cannot be reproduced bit for bit; the forest therefore takes the families as
data.  This module generates them with numpy in the same *shape*: a pool of
`family_size` functions, `table_num` chains of `chain_length` functions pulled
from the pool with replacement, each chain shuffled `permutation_num` times
(angle only), and one independent `partition_bits x 32` partitioner per table
(DensevectorRDFInit.scala:63-77).
"""
import numpy as np


def make_family(dim, table_num=10, permutation_num=3, chain_length=32, family="angle",
                family_size=100, partition_bits=3, orthogonal=True, pstable_mu=0.0,
                pstable_sigma=1.0, pstable_w=4, seed=88387, part_seed=88388):
    """-> dict(fns, b, w, chain_fn, part_fns, part_b, part_w, family, partition_bits)."""
    rng = np.random.default_rng(seed)
    if family == "angle":
        if orthogonal:
            # AngleHashFamily.scala:73-85: Q of a QR of a uniform random matrix.
            m = rng.random((max(family_size, dim), dim))
            q, _ = np.linalg.qr(m)
            fns = np.ascontiguousarray(q[:family_size])
        else:
            v = rng.random((family_size, dim)) * rng.choice([-1.0, 1.0], size=(family_size, dim))
            fns = v / np.sqrt((v * v).sum(axis=1, keepdims=True))
        b = w = None
        L = table_num * permutation_num
        chain_fn = np.empty((L, chain_length), dtype=np.int32)
        for t in range(table_num):
            pulled = rng.integers(0, family_size, size=chain_length)
            for p in range(permutation_num):
                chain_fn[permutation_num * t + p] = rng.permutation(pulled)
    elif family == "pStable":
        fns = rng.normal(pstable_mu, pstable_sigma, size=(family_size, dim))
        b = rng.random(family_size) * pstable_w
        w = np.full(family_size, pstable_w, dtype=np.int32)
        L = table_num  # PStableHashFamily.pick returns tableNum chains (PStableHashFamily.scala:68)
        chain_fn = rng.integers(0, family_size, size=(L, chain_length)).astype(np.int32)
    else:
        raise ValueError(family)
    prng = np.random.default_rng(part_seed)
    pb = partition_bits
    if family == "angle":
        part_fns = np.empty((L, pb, 32))
        for t in range(L):
            m = prng.random((32, 32))
            q, _ = np.linalg.qr(m)
            part_fns[t] = q[prng.integers(0, 32, size=pb)] if pb else np.empty((0, 32))
        part_b = part_w = None
    else:
        part_fns = prng.normal(pstable_mu, pstable_sigma, size=(L, pb, 32))
        part_b = prng.random((L, pb)) * pstable_w
        part_w = np.full((L, pb), pstable_w, dtype=np.int32)
    return dict(family=family, fns=np.ascontiguousarray(fns, dtype=np.float64), b=b, w=w,
                chain_fn=chain_fn, part_fns=np.ascontiguousarray(part_fns, dtype=np.float64),
                part_b=part_b, part_w=part_w, partition_bits=pb)


def make_gaussian_family(dim, table_num, chain_length=32, partition_bits=3, seed=88387,
                         part_seed=88388):
    """One distinct N(0,1) function per chain position (pool = table_num*chain_length)."""
    rng = np.random.default_rng(seed)
    F = table_num * chain_length
    fns = rng.normal(size=(F, dim))
    chain_fn = np.arange(F, dtype=np.int32).reshape(table_num, chain_length)
    prng = np.random.default_rng(part_seed)
    part_fns = prng.normal(size=(table_num, partition_bits, 32))
    return dict(family="angle", fns=fns, b=None, w=None, chain_fn=chain_fn, part_fns=part_fns,
                part_b=None, part_w=None, partition_bits=partition_bits)


def make_clustered(n, dim, nq, n_clusters=1000, sigma=0.3, noise=0.05, seed=1234, qseed=4321,
                   dtype=np.float32, chunk=100000):
    """Mixture of `n_clusters` Gaussians (centres N(0,I), std `sigma`); queries are
    base rows plus N(0, noise^2) noise.  -> (X [n,dim], Q [nq,dim], picked rows)."""
    rng = np.random.default_rng(seed)
    centres = rng.normal(size=(n_clusters, dim)).astype(np.float32)
    X = np.empty((n, dim), dtype=dtype)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        lab = rng.integers(0, n_clusters, size=e - s)
        X[s:e] = centres[lab] + sigma * rng.standard_normal((e - s, dim), dtype=np.float32)
    qrng = np.random.default_rng(qseed)
    pick = qrng.integers(0, n, size=nq)
    Q = (X[pick].astype(np.float32) + noise * qrng.standard_normal((nq, dim), dtype=np.float32)).astype(dtype)
    return X, Q, pick


def make_sift_like(n, dim, nq, seed=99, dtype=np.float32):
    """Integer-valued 0..255 components stored as floats (SIFT-shaped)."""
    rng = np.random.default_rng(seed)
    centres = rng.integers(0, 256, size=(256, dim)).astype(np.float32)
    lab = rng.integers(0, 256, size=n)
    X = np.clip(np.rint(centres[lab] + 12.0 * rng.standard_normal((n, dim), dtype=np.float32)), 0, 255).astype(dtype)
    pick = rng.integers(0, n, size=nq)
    Q = np.clip(np.rint(X[pick] + 4.0 * rng.standard_normal((nq, dim), dtype=np.float32)), 0, 255).astype(dtype)
    return X, Q, pick


def make_unit_gaussian(n, dim, nq, noise=0.05, seed=777, dtype=np.float32, chunk=100000):
    """Unit-normalised N(0, I) vectors; queries are base rows plus N(0, noise^2 / dim) noise per
    component, normalised again.  -> (X [n,dim], Q [nq,dim], picked rows)."""
    rng = np.random.default_rng(seed)
    X = np.empty((n, dim), dtype=dtype)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        v = rng.standard_normal((e - s, dim), dtype=np.float32)
        X[s:e] = v / np.linalg.norm(v, axis=1, keepdims=True)
    pick = rng.integers(0, n, size=nq)
    q = X[pick] + (noise / np.sqrt(dim)) * rng.standard_normal((nq, dim), dtype=np.float32)
    Q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(dtype)
    return X, Q, pick


def more_queries(kind, X, nq, seed):
    """Another query batch over the base X in the recipe of make_clustered ("clustered"), make_sift_like
    ("sift") or make_unit_gaussian ("unit"), drawn from its own seed.  -> (Q [nq,dim], picked rows)."""
    rng = np.random.default_rng(seed)
    n, dim = X.shape
    pick = rng.integers(0, n, size=nq)
    g = rng.standard_normal((nq, dim), dtype=np.float32)
    if kind == "clustered":
        Q = X[pick].astype(np.float32) + np.float32(0.05) * g
    elif kind == "sift":
        Q = np.clip(np.rint(X[pick] + np.float32(4.0) * g), 0, 255)
    elif kind == "unit":
        q = X[pick] + np.float32(0.05 / np.sqrt(dim)) * g
        Q = q / np.linalg.norm(q, axis=1, keepdims=True)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(Q, dtype=X.dtype), pick


def fresh_queries(kind, X, nq, seed):
    """Out-of-sample queries: NOT noisy copies of base rows but fresh draws from the distribution the base of that
    recipe was drawn from ("clustered": the same 1000-centre mixture, "sift": the same 256 centres, "unit": fresh unit
    Gaussian vectors), or "uniform": components uniform in [0, 1) as in the reference's README test
    (README.md:37-38).  -> (Q [nq, dim], ids [nq] = -1 .. : none of them is a stored row)."""
    rng = np.random.default_rng(seed)
    n, dim = X.shape
    if kind == "clustered":
        centres = np.random.default_rng(1234).normal(size=(1000, dim)).astype(np.float32)      # make_clustered's first draw
        Q = centres[rng.integers(0, 1000, size=nq)] + np.float32(0.3) * rng.standard_normal((nq, dim), dtype=np.float32)
    elif kind == "sift":
        centres = np.random.default_rng(99).integers(0, 256, size=(256, dim)).astype(np.float32)   # make_sift_like's first draw
        Q = np.clip(np.rint(centres[rng.integers(0, 256, size=nq)] + np.float32(12.0) * rng.standard_normal((nq, dim), dtype=np.float32)), 0, 255)
    elif kind == "unit":
        q = rng.standard_normal((nq, dim), dtype=np.float32)
        Q = q / np.linalg.norm(q, axis=1, keepdims=True)
    elif kind == "uniform":
        Q = rng.random((nq, dim), dtype=np.float32)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(Q, dtype=X.dtype), np.full(nq, -1, dtype=np.int32)


# Row-addressable recipes: row r depends only on (seed, r // ROW_CHUNK, r % ROW_CHUNK), so a rank can produce any
# range of a base without producing - or holding - the rows before it.
ROW_CHUNK = 65536


def unit_gaussian_rows(lo, hi, dim, seed=777, dtype=np.float32):
    """Rows [lo, hi) of a base of unit-normalised N(0, I) vectors (the c5 recipe, chunk-seeded)."""
    out = np.empty((max(hi - lo, 0), dim), dtype=dtype)
    r = lo
    while r < hi:
        c = r // ROW_CHUNK
        c0 = c * ROW_CHUNK
        e = min(hi, c0 + ROW_CHUNK)
        v = np.random.default_rng([seed, c]).standard_normal((ROW_CHUNK, dim), dtype=np.float32)[r - c0:e - c0]
        out[r - lo:e - lo] = v / np.linalg.norm(v, axis=1, keepdims=True)
        r = e
    return out


def unit_gaussian_queries(n, dim, nq, noise=0.05, seed=777, qseed=778, dtype=np.float32, batches=1):
    """Queries over the unit_gaussian_rows base: rows picked at random plus N(0, noise^2 / dim) noise per component,
    normalised again; the base is generated chunk by chunk, once, for all `batches` batches together (only chunks that
    hold a picked row).  -> (Q [nq, dim], picked rows), or a list of `batches` such pairs."""
    rng = np.random.default_rng(qseed)
    total = nq * batches
    pick = rng.integers(0, n, size=total)
    rows = np.empty((total, dim), dtype=np.float32)
    order = np.argsort(pick, kind="stable")
    i = 0
    while i < total:
        c = int(pick[order[i]]) // ROW_CHUNK
        j = i
        while j < total and int(pick[order[j]]) // ROW_CHUNK == c:
            j += 1
        chunk = unit_gaussian_rows(c * ROW_CHUNK, min(n, (c + 1) * ROW_CHUNK), dim, seed=seed)
        rows[order[i:j]] = chunk[pick[order[i:j]] - c * ROW_CHUNK]
        i = j
    q = rows + np.float32(noise / np.sqrt(dim)) * rng.standard_normal((total, dim), dtype=np.float32)
    Q = np.ascontiguousarray(q / np.linalg.norm(q, axis=1, keepdims=True), dtype=dtype)
    if batches == 1:
        return Q, pick
    return [(np.ascontiguousarray(Q[b * nq:(b + 1) * nq]), pick[b * nq:(b + 1) * nq]) for b in range(batches)]


def unit_gaussian_chunks(n, dim, seed=777, dtype=np.float32):
    """The unit_gaussian_rows base of n rows as a generator of ROW_CHUNK-row pieces in id order."""
    for lo in range(0, n, ROW_CHUNK):
        yield unit_gaussian_rows(lo, min(n, lo + ROW_CHUNK), dim, seed=seed, dtype=dtype)
