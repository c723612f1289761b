"""The host mirror of the reference's objects (LSHServer / LSH / DensevectorRDFInit /
Vectors / Config): parsers and family files on the CPU, the README "simple test" flow and
topKAndPrecisionScore on the GPU against the oracle."""
import os

import numpy as np
import pytest

from similaritysearchbyrdf_amd import host
from similaritysearchbyrdf_amd.host import Config, DenseVector, DensevectorRDFInit, LSH, LSHServer, MclabError, Vectors

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "reference_fixtures")

# the keys of src/test/scala/mclab/TestSettings.scala:8-45 that the dense path reads
BASE_CONF = """
mclab.confType=lsh
mclab.lsh.name = angle
mclab.lsh.generateByPulling = true
mclab.lsh.IsOrthogonal = true
mclab.lsh.generateMethod = default
mclab.lsh.familyFilePath = "hashFamily/lsh-bestHashFamily-angle-TableNum-10"
mclab.lsh.partitionFamilyFilePath="hashFamily/partition-bestHashFamily-angle"
mclab.lsh.family.pstable.mu = 0.0
mclab.lsh.family.pstable.sigma = 1.0
mclab.lsh.family.pstable.w = 4
mclab.lsh.familySize = 100
mclab.lsh.vectorDim = 100
mclab.lsh.tableNum = 10
mclab.lsh.permutationNum = 3
mclab.lsh.typeOfIndex = original
mclab.lsh.featureDataFormat = sparse
mclab.lshTable.bufferOverflow=500
mclab.lshTable.bucketBits=28
mclab.lshTable.dirNodeSize=32
mclab.lshTable.chainLength = 32
mclab.dataTable.numPartitions=2
mclab.lsh.partitionBits=3
mclab.lsh.ramThreshold=2147483647
mclab.lsh.workingDirRoot="PersistIndex"
mclab.insertThreadNum=5
mclab.queryThreadNum=5
mclab.lsh.topK = 10
"""


def conf(**over):
    c = Config.parseString(BASE_CONF)
    for k, v in over.items():
        c = c.withValue(k.replace("__", "."), v)
    return c


# ------------------------------------------------------------------ CPU: parsers, config, files

def test_vectors_fromString_fixture():
    # src/test/scala/mclab/lsh/vector/VectorSuite.scala:9-25 on VectorTest/sparsevectorfile
    lines = open(os.path.join(FIX, "sparsevectorfile")).read().splitlines()
    got = [Vectors.fromString(l) for l in lines]
    assert [(g[0], g[1], g[2].tolist(), g[3].tolist()) for g in got] == [
        (3, 3, [0, 1, 2], [1.0, 2.0, 3.0]), (4, 3, [0, 1, 2], [4.0, 5.0, 6.0]), (5, 2, [0, 1], [1.0, 2.0])]
    with pytest.raises(MclabError):
        Vectors.fromString("(1,3,[0,1]")


def test_vectors_parseDense_and_analysisKNN():
    # Vector.scala:215-219: "[1,[0.1,0.2,0.4,0.9]]" written by the authors' python tooling
    vid, vals = Vectors.parseDense("[7, [0.1, 0.2, 0.4, 0.9]]")
    assert vid == 7 and vals.tolist() == [0.1, 0.2, 0.4, 0.9]
    vid, vals = Vectors.parseDense("12,-1.5e-3,2")
    assert vid == 12 and vals.tolist() == [-1.5e-3, 2.0]
    with pytest.raises(MclabError):
        Vectors.parseDense("x,1.0")
    # VectorSuite.scala:46
    assert Vectors.analysisKNN("[1,2,3,4,5,7,8, 8,91,10,12,23]", 12).tolist() == [1, 2, 3, 4, 5, 7, 8, 8, 91, 10, 12, 23]
    assert Vectors.analysisKNN("[1,2,3,4]", 2).tolist() == [1, 2]
    with pytest.raises(MclabError):
        Vectors.analysisKNN("[1,2]", 3)


def test_config_surface():
    c = conf()
    assert c.getInt("mclab.lsh.tableNum") == 10 and c.getString("mclab.lsh.name") == "angle"
    assert c.getBoolean("mclab.lsh.IsOrthogonal") is True and c.getDouble("mclab.lsh.family.pstable.sigma") == 1.0
    assert c.getString("mclab.lsh.workingDirRoot") == "PersistIndex"
    over = Config.parseString("mclab.lsh.tableNum = 1").withFallback(c)
    assert over.getInt("mclab.lsh.tableNum") == 1 and over.getInt("mclab.lsh.partitionBits") == 3
    with pytest.raises(MclabError):
        c.getInt("mclab.missing.key")


def _install(c):
    LSHServer.isUseDense = True
    LSHServer.lshEngine = LSH(c)
    return LSHServer.lshEngine.family()


def test_lsh_default_family_shapes():
    # AngleHashFamily.scala:121-149: tableNum*permutationNum chains; the permutationNum chains of a
    # table are shuffles of the same chainLength functions, pulled from a familySize pool
    fam = _install(conf())
    assert fam["fns"].shape == (100, 100) and fam["chain_fn"].shape == (30, 32)
    q = fam["fns"]
    assert np.allclose(q.T @ q, np.eye(100), atol=1e-9)            # orthogonal family
    for t in range(10):
        a, b, c3 = (sorted(fam["chain_fn"][3 * t + p].tolist()) for p in range(3))
        assert a == b == c3
    assert not np.array_equal(fam["chain_fn"][0], fam["chain_fn"][1])
    # PStableHashFamily.scala:65-77: only tableNum chains
    fam = _install(conf(mclab__lsh__name="pStable"))
    assert fam["family"] == "pStable" and fam["chain_fn"].shape == (10, 32)
    assert fam["w"].tolist() == [4] * 100 and ((fam["b"] >= 0) & (fam["b"] < 4)).all()
    # not pulling: every chain position gets a fresh unit vector
    fam = _install(conf(mclab__lsh__generateByPulling="false", mclab__lsh__IsOrthogonal="false", mclab__lsh__tableNum=2))
    assert fam["fns"].shape == (64, 100) and np.allclose((fam["fns"] ** 2).sum(axis=1), 1.0)
    with pytest.raises(MclabError):
        LSHServer.lshEngine = LSH(conf(mclab__lsh__typeOfIndex="nope"))


def test_family_file_round_trip(tmp_path):
    # LSH.outPutTheHashFunctionsIntoFile (LSH.scala:173-195) -> generateTableChainFromFile
    # (AngleHashFamily.scala:158-177, PStableHashFamily.scala:88-108)
    for name in ("angle", "pStable"):
        c = conf(mclab__lsh__name=name, mclab__lsh__permutationNum=1, mclab__lsh__tableNum=3)
        fam = _install(c)
        path = tmp_path / ("family-" + name)
        LSHServer.lshEngine.outPutTheHashFunctionsIntoFile(path)
        text = open(path, "rb").read()
        assert text.count(b"\r\n") == 96 and text.startswith(b"(0,100,[0,1,2,")
        back = _install(c.withValue("mclab.lsh.generateMethod", "fromfile").withValue("mclab.lsh.familyFilePath", str(path)))
        assert back["chain_fn"].shape == (3, 32)
        want = fam["fns"][fam["chain_fn"].reshape(-1)]
        assert np.array_equal(back["fns"][back["chain_fn"].reshape(-1)], want)
        if name == "pStable":
            assert np.array_equal(back["b"][back["chain_fn"].reshape(-1)], fam["b"][fam["chain_fn"].reshape(-1)])


def test_reference_family_fixtures():
    # the reference's saved partitioner family (2 functions over the 32 hash bits) loads as one chain
    c = conf(mclab__confType="partition", mclab__lsh__vectorDim=32, mclab__lshTable__chainLength=2,
             mclab__lsh__generateMethod="fromfile",
             mclab__lsh__partitionFamilyFilePath=os.path.join(FIX, "partition-bestHashFamily-angle-TableNum-1"))
    fam = _install(c)
    assert fam["fns"].shape == (2, 32) and fam["chain_fn"].tolist() == [[0, 1]]
    assert fam["fns"][0, 0] == -0.17855945281171176
    # AngleHashSuite.scala:52-73's file lists 2 of 3 indices per function: the dense path would multiply
    # those values positionally (SimilarityCalculator.scala:46), so such files are rejected
    with pytest.raises(MclabError) as ei:
        LSHServer.lshEngine = LSH(conf(mclab__lsh__vectorDim=3, mclab__lshTable__chainLength=3,
                                       mclab__lsh__generateMethod="fromfile",
                                       mclab__lsh__familyFilePath=os.path.join(FIX, "angleHashFile")))
    assert "positionally" in str(ei.value)
    with pytest.raises(MclabError):
        LSHServer.lshEngine = LSH(conf(mclab__lsh__generateMethod="fromfile", mclab__lsh__familyFilePath="/nonexistent"))


# ------------------------------------------------------------------ GPU: the README flow

def _write_dense_file(path, X, ids=None):
    with open(path, "w") as f:
        for i, row in enumerate(X):
            f.write("[%d, [%s]]\n" % (i if ids is None else ids[i], ", ".join(repr(float(v)) for v in row)))


def _oracle_for_mirror(c, L, pb, B):
    from oracle import oracle as O
    fam = LSHServer.lshEngine.family()
    DensevectorRDFInit.set_shape(L, pb)
    pf, pbv, pw = DensevectorRDFInit.partitioner()
    p = dict(family=fam["family"], fns=fam["fns"], b=fam["b"], w=fam["w"], chain_fn=fam["chain_fn"][:L],
             part_fns=pf, partition_bits=pb, bucket_overflow=B)
    if fam["family"] == "pStable":
        p.update(part_b=pbv, part_w=pw)
    return O.OracleIndex(p)


@pytest.mark.gpu
@pytest.mark.parametrize("name,perm", [("angle", 3), ("pStable", 1)])
def test_readme_simple_test_flow(tmp_path, name, perm):
    # README.md:32-43 of the reference, with a generated file in place of the GloVe one
    rng = np.random.default_rng(3)
    centres = rng.normal(size=(15, 100))
    X = centres[rng.integers(0, 15, size=3000)] + 0.3 * rng.normal(size=(3000, 100))
    path = tmp_path / "vectors.DenseVector.txt"
    _write_dense_file(path, X, ids=np.arange(3000)[::-1])     # file ids are NOT the index ids (Dense:181,190)
    c = conf(mclab__lsh__name=name, mclab__lsh__permutationNum=perm, mclab__lsh__tableNum=4,
             mclab__lshTable__bufferOverflow=40, mclab__lsh__family__pstable__w=2)
    LSHServer.lshEngine = None
    LSHServer.isUseDense = True
    LSHServer.lshEngine = LSH(c)
    n = DensevectorRDFInit.newMultiThreadFit(str(path), c)
    assert n == 3000 and DensevectorRDFInit.size() == 3000
    values = [X[0], X[1] + 0.01, rng.random(100)]
    results = DensevectorRDFInit.NewMultiThreadQueryBatch([0, 1, 2], [DenseVector(i, v) for i, v in enumerate(values)], 0)
    orc = _oracle_for_mirror(c, 4 * perm, 3, 40)
    orc.fit(X)
    for steps in (0, 1, 2):
        got = DensevectorRDFInit.NewMultiThreadQueryBatch([0, 1, 2], np.stack(values), steps)
        off, ids = orc.query_batch([0, 1, 2], np.stack(values), steps=steps)
        assert got == [set(ids[off[i]:off[i + 1]].tolist()) for i in range(3)]
    assert results[0] and 0 not in results[0]          # id 0 is inside the Integer cache: excluded (RDT:982)
    # LSH.calculateIndex on the device equals the oracle's hash
    for t in (0, 3):
        assert LSHServer.lshEngine.calculateIndex(DenseVector(0, values[2]), t)[0] == orc.hash(t, values[2])
    dt, ht = DensevectorRDFInit.getDtAndHtNumDistribution()
    assert dt[0] == 3000 and ht.shape == (8,) and abs(ht.sum() - 3000) < 1e-9
    # the query-insert overload (Dense:372-399): vectors get ids 3000.. and are then findable
    new = np.stack([X[5] + 0.001, X[6]])
    got = DensevectorRDFInit.NewMultiThreadQueryBatch(new, 1)
    orc.fit(new)
    off, ids = orc.query_batch([3000, 3001], new, steps=1)
    assert got == [set(ids[off[i]:off[i + 1]].tolist()) for i in range(2)]
    assert 3001 in got[1] or 6 in got[1]
    DensevectorRDFInit.clearAndClose()
    assert DensevectorRDFInit.size() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["continueBitsCount", "sampling", "angleNewMethod"])
def test_calculateIndex_all_tables_skips_the_transform(kind):
    # LSH.scala:135-138: tableId < 0 returns every chain's value as computed, WITHOUT the typeOfIndex transform
    # that the single-table branch applies (LSH.scala:152-161)
    from oracle import oracle as O
    c = conf(mclab__lsh__typeOfIndex=kind, mclab__lsh__tableNum=3, mclab__lsh__permutationNum=2)
    LSHServer.lshEngine = None
    LSHServer.isUseDense = True
    LSHServer.lshEngine = LSH(c)
    fam = LSHServer.lshEngine.family()
    base = dict(family=fam["family"], fns=fam["fns"], b=fam["b"], w=fam["w"], chain_fn=fam["chain_fn"],
                part_fns=np.zeros((6, 0, 32), dtype=np.int32), partition_bits=0, bucket_overflow=50)
    if kind == "sampling":        # Sampling.scala: the bit order is scala.util.Random(88387).shuffle(0 until 32)
        import ctypes as C
        si = np.empty(32, dtype=np.int32)
        O.lib().orc_scala_shuffle_0_31(88387, si.ctypes.data_as(C.POINTER(C.c_int32)))
        base["sampling_index"] = si
    raw, cooked = O.OracleIndex(dict(base, type_of_index="original")), O.OracleIndex(dict(base, type_of_index=kind))
    v = np.random.default_rng(5).normal(size=100)
    every = LSHServer.lshEngine.calculateIndex(DenseVector(0, v), -1)
    assert list(every) == [raw.hash(t, v) for t in range(6)]
    one = [LSHServer.lshEngine.calculateIndex(DenseVector(0, v), t)[0] for t in range(6)]
    assert one == [cooked.hash(t, v) for t in range(6)]
    assert one != list(every)


@pytest.mark.gpu
def test_topKAndPrecisionScore_matches_oracle(tmp_path):
    rng = np.random.default_rng(8)
    centres = rng.normal(size=(10, 100))
    X = centres[rng.integers(0, 10, size=4000)] + 0.3 * rng.normal(size=(4000, 100))
    c = conf(mclab__lsh__tableNum=3, mclab__lsh__permutationNum=2, mclab__lshTable__bufferOverflow=60)
    LSHServer.lshEngine = None
    LSHServer.isUseDense = True
    LSHServer.lshEngine = LSH(c)
    DensevectorRDFInit.fitVectors(X, c)
    nq = 300
    # ground truth by inner product over everything (what the authors' files hold), written and read back
    scores = X[:nq] @ X.T
    gt = np.argsort(-scores, axis=1)[:, :10]
    gpath = tmp_path / "gt.txt"
    with open(gpath, "w") as f:
        for row in gt:
            f.write("[" + ", ".join(str(int(x)) for x in row) + "]\n")
    groundTruth = DensevectorRDFInit.getTopKGroundTruth(str(gpath), 10)
    assert groundTruth[3] == set(gt[3].tolist())
    topk, score = DensevectorRDFInit.topKAndPrecisionScore(X, groundTruth, c, steps=1)
    orc = _oracle_for_mirror(c, 6, 3, 60)
    orc.fit(X)
    off, ids = orc.query_batch(np.arange(nq), X[:nq], steps=1)
    wi, ws = orc.topk(X[:nq], off, ids, k=10, metric="dot")
    want, acc = [], 0.0
    for q in range(nq):
        row = [int(x) for x in wi[q] if x >= 0]
        if not row:
            continue
        want.append(row)
        acc += len(set(row) & groundTruth[q]) / nq
    assert topk == want
    assert abs(score - acc / 10) < 1e-12 and 0.0 < score <= 1.0


@pytest.mark.gpu
def test_query_before_fit_and_wrong_switch():
    c = conf(mclab__lsh__tableNum=1, mclab__lsh__permutationNum=1)
    DensevectorRDFInit.clearAndClose()
    # Dense:422-424: prints "need to fit the data first", returns empty sets
    got = DensevectorRDFInit.NewMultiThreadQueryBatch([0, 1], np.zeros((2, 100)), 0)
    assert got == [set(), set()]
    LSHServer.lshEngine = LSH(c)
    LSHServer.isUseDense = False                      # README step 3: the dense path needs the switch on
    with pytest.raises(MclabError) as ei:
        DensevectorRDFInit.fitVectors(np.zeros((4, 100)), c)
    assert "isUseDense" in str(ei.value)
    LSHServer.isUseDense = True
    # pStable (and file-loaded) engines hold tableNum chains; tableNum*permutationNum tables overrun them (LSH.scala:154)
    pc = conf(mclab__lsh__name="pStable", mclab__lsh__tableNum=2, mclab__lsh__permutationNum=3)
    LSHServer.lshEngine = LSH(pc)
    with pytest.raises(MclabError) as ei:
        DensevectorRDFInit.fitVectors(np.zeros((4, 100)), pc)
    assert "IndexOutOfBounds" in str(ei.value)


# ------------------------------------------------------------------ SparsevectorRDFInit

def test_vectors_fromPythonString():
    # Vector.scala:189-208: "[1, 3, [1, 2, 3], [1.0, 2.0, 3.0]]", spaces anywhere
    vid, size, idx, vals = Vectors.fromPythonString("[1, 3, [1, 2, 3], [1.0, 2.0, 3.0]]")
    assert (vid, size, idx.tolist(), vals.tolist()) == (1, 3, [1, 2, 3], [1.0, 2.0, 3.0])
    vid, size, idx, vals = Vectors.fromPythonString("[7,100,[0,99],[ -0.5 , 2e-3 ]]")
    assert (vid, size, idx.tolist(), vals.tolist()) == (7, 100, [0, 99], [-0.5, 0.002])
    vid, size, idx, vals = Vectors.fromPythonString("[2, 5, [], []]")
    assert (vid, size, idx.size, vals.size) == (2, 5, 0, 0)
    with pytest.raises(MclabError):
        Vectors.fromPythonString("[1, 3, [1, 2, 3]]")


def _sparse_rows(X, rng, keep=0.6):
    """Rows of X with about `keep` of the entries kept, as SparseVectors (the dropped entries are zeros)."""
    from similaritysearchbyrdf_amd.host import SparseVector
    out, dense = [], np.zeros_like(X)
    for i, x in enumerate(X):
        idx = np.flatnonzero(rng.random(x.shape[0]) < keep)
        out.append(SparseVector(i, x.shape[0], idx, x[idx]))
        dense[i, idx] = x[idx]
    return out, dense


@pytest.mark.gpu
@pytest.mark.parametrize("name,perm", [("angle", 2), ("pStable", 1)])
def test_sparse_object_flow_matches_oracle(tmp_path, name, perm):
    # SparsevectorRDFInit: fit from a file of python-style sparse lines; the keyed batch query walks the hash of the
    # key's own vector only (RandomDrawTreeMap.java:630-675 via SparsevectorRDFInit.scala:407), query() with vectors
    # walks the bit flips (RandomDrawTreeMap.java:742-797); both against the oracle fitted on the dense forms
    from similaritysearchbyrdf_amd.host import SparsevectorRDFInit
    rng = np.random.default_rng(11)
    centres = rng.normal(size=(12, 100))
    X = centres[rng.integers(0, 12, size=2500)] + 0.3 * rng.normal(size=(2500, 100))
    rows, D = _sparse_rows(X, rng)
    path = tmp_path / "vectors.SparseVector.txt"
    with open(path, "w") as f:
        for i, v in enumerate(rows):
            f.write("[%d, %d, [%s], [%s]]\n" % (2500 - i, v.size, ", ".join(str(int(j)) for j in v.indices),
                                                 ", ".join(repr(float(x)) for x in v.values)))
    c = conf(mclab__lsh__name=name, mclab__lsh__permutationNum=perm, mclab__lsh__tableNum=3,
             mclab__lshTable__bufferOverflow=30, mclab__lsh__family__pstable__w=2)
    LSHServer.lshEngine = None
    LSHServer.isUseDense = False
    LSHServer.lshEngine = LSH(c)
    n = SparsevectorRDFInit.newMultiThreadFit(str(path), c)
    assert n == 2500 and SparsevectorRDFInit.size() == 2500
    from oracle import oracle as O
    fam = LSHServer.lshEngine.family()
    L = 3 * perm
    pf, pbv, pw = SparsevectorRDFInit.partitioner(L, 3)
    p = dict(family=fam["family"], fns=fam["fns"], b=fam["b"], w=fam["w"], chain_fn=fam["chain_fn"][:L], part_fns=pf,
             partition_bits=3, bucket_overflow=30)
    if fam["family"] == "pStable":
        p.update(part_b=pbv, part_w=pw)
    orc = O.OracleIndex(p)
    orc.fit(D)
    keys = [0, 5, 127, 128, 700, 2499]
    for steps in (0, 1, 3):
        orc.set_probe_own(True)
        off, ids = orc.query_batch(keys, D[keys], steps=steps)
        got = SparsevectorRDFInit.NewMultiThreadQueryBatch(keys, steps)
        assert got == [set(ids[off[i]:off[i + 1]].tolist()) for i in range(len(keys))]
        orc.set_probe_own(False)
        off, ids = orc.query_batch(keys, D[keys], steps=steps)
        got = SparsevectorRDFInit.query(keys, [rows[k] for k in keys], steps)
        assert got == [set(ids[off[i]:off[i + 1]].tolist()) for i in range(len(keys))]
    assert SparsevectorRDFInit.querySingleKey(700, rows[700], 3) == got[4]
    # a key the index does not hold: the reference's worker prints "need to fit the data first" and leaves the set empty
    assert SparsevectorRDFInit.NewMultiThreadQueryBatch([999999], 1) == [set()]
    # the insert-then-query overload (SparsevectorRDFInit.scala:360-387): new ids 2500.., looked up by key
    new_rows, newD = _sparse_rows(X[5:7] + 1e-3, rng)
    got = SparsevectorRDFInit.NewMultiThreadQueryBatch(new_rows, 1)
    orc.fit(newD)
    orc.set_probe_own(True)
    off, ids = orc.query_batch([2500, 2501], newD, steps=1)
    assert got == [set(ids[off[i]:off[i + 1]].tolist()) for i in range(2)]
    # top-k + recall through query() (SparsevectorRDFInit.scala:458-502)
    orc.set_probe_own(False)
    nq = 100
    allD = np.vstack([D, newD])
    gt = np.argsort(-(allD[:nq] @ allD.T), axis=1)[:, :10]
    groundTruth = [set(r.tolist()) for r in gt]
    topk, score, ms = SparsevectorRDFInit.topKAndPrecisionScore(allD, groundTruth, c, steps=1)
    off, ids = orc.query_batch(np.arange(nq), allD[:nq], steps=1)
    wi, ws = orc.topk(allD[:nq], off, ids, k=10, metric="dot")
    want = [[int(x) for x in wi[q] if x >= 0] for q in range(nq)]
    assert topk == [r for r in want if r] and 0.0 < score <= 1.0 and ms >= 0
    # the dense switch must be off for this object (Hasher.scala:47-51)
    LSHServer.isUseDense = True
    with pytest.raises(MclabError) as ei:
        SparsevectorRDFInit.fitVectors(rows[:4], c)
    assert "isUseDense" in str(ei.value)
    LSHServer.isUseDense = False
    SparsevectorRDFInit.clearAndClose()
    assert SparsevectorRDFInit.size() == 0
