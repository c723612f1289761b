#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ (run in the build container,
where the reference checkout is mounted; the GPU box only reads the committed outputs).

Inputs: the hash-family DATA files the reference's own tests hold
  src/test/resources/hashFamily/lsh-bestHashFamily-angle-TableNum-10   (10 chains x 32 fns x 100-d)
  src/test/resources/hashFamily/partition-bestHashFamily-angle-TableNum-1  (2 fns x 32-d)
re-serialised as float64 arrays, plus seeded synthetic vectors.
Outputs: the arrays, and the CPU oracle's answers on them (hashes, sub-index ids,
candidate sets for steps 0..pb, top-10 by dot and l2, tag-quirk count).  The reference
itself cannot run here (no JVM), so these are the restatement's answers, not the JVM's.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/src/test/resources/hashFamily"


def parse_family(path):
    rows = []
    for line in open(path):
        line = line.strip()
        if not line:
            continue
        head, idx, vals = line.split(",[")
        vid, size = [int(x) for x in head.strip("(").split(",")]
        idx = [int(x) for x in idx.strip("]").split(",") if x]
        vals = [float(x) for x in vals.replace("])", "").split(",") if x]
        assert idx == list(range(size)) and len(vals) == size
        rows.append(vals)
    return np.asarray(rows, dtype=np.float64)


def golden_problem():
    fns = np.load(os.path.join(HERE, "lsh_best_family_angle_320x100.npy"))
    pfile = np.load(os.path.join(HERE, "partition_best_family_angle_2x32.npy"))
    L, m, pb = 10, 32, 3
    rng = np.random.default_rng(20181)
    part = np.empty((L, pb, 32))
    for t in range(L):
        part[t, :2] = pfile                      # the reference's saved pb=2 partitioner
        part[t, 2] = rng.normal(size=32)         # padded with a seeded third function per table
    params = dict(family="angle", fns=fns, chain_fn=np.arange(L * m, dtype=np.int32).reshape(L, m),
                  part_fns=part, partition_bits=pb, bucket_bits=28, dir_node_size=32, exclude_self="compat")
    centres = rng.normal(size=(12, 100))
    lab = rng.integers(0, 12, size=2000)
    X = (centres[lab] + 0.25 * rng.normal(size=(2000, 100))).astype(np.float32)
    pick = np.array([0, 1, 64, 127, 128, 129, 500, 777, 1000, 1500, 1999, 42, 43, 44, 45, 46])
    Q = (X[pick] + 0.02 * rng.normal(size=(16, 100)).astype(np.float32)).astype(np.float32)
    Q[:6] = X[pick[:6]]                          # exact copies: self-exclusion boundary cases
    qids = pick.astype(np.int32).copy()
    qids[10] = -5
    return params, X, Q, qids


def main():
    from oracle import oracle as O
    np.save(os.path.join(HERE, "lsh_best_family_angle_320x100.npy"),
            parse_family(os.path.join(REF, "lsh-bestHashFamily-angle-TableNum-10")))
    np.save(os.path.join(HERE, "partition_best_family_angle_2x32.npy"),
            parse_family(os.path.join(REF, "partition-bestHashFamily-angle-TableNum-1")))
    params, X, Q, qids = golden_problem()
    out = {"X": X, "Q": Q, "qids": qids, "part_fns": params["part_fns"]}
    X64, Q64 = X.astype(np.float64), Q.astype(np.float64)
    for B in (4, 500):
        p = dict(params, bucket_overflow=B)
        orc = O.OracleIndex(p)
        orc.fit(X64, threads=4)
        h = np.stack([orc.fitted_hashes(t)[0] for t in range(orc.L)], axis=1)
        pt = np.stack([orc.fitted_hashes(t)[1] for t in range(orc.L)], axis=1)
        out["hashes"] = h
        out["parts"] = pt
        out["misflag_B%d" % B] = np.int64(orc.misflag_events)
        out["leaves_B%d" % B] = np.int64(sum(orc.table_stats(t)[1] for t in range(orc.L)))
        for steps in range(4):
            off, ids = orc.query_batch(qids, Q64, steps=steps)
            out["off_B%d_s%d" % (B, steps)] = off
            out["ids_B%d_s%d" % (B, steps)] = ids
            if steps == 1:
                for metric in ("dot", "l2"):
                    ti, ts = orc.topk(Q64, off, ids, k=10, metric=metric)
                    out["top_ids_B%d_%s" % (B, metric)] = ti
                    out["top_sc_B%d_%s" % (B, metric)] = ts
        print("B=%d leaves=%d misflag=%d sets(step1)=%s" % (B, out["leaves_B%d" % B], orc.misflag_events,
                                                            np.diff(out["off_B%d_s1" % B])[:6]))
    np.savez_compressed(os.path.join(HERE, "golden_c1.npz"), **out)


if __name__ == "__main__":
    main()
