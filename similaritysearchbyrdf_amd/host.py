"""Python spelling of the reference's host objects for the dense batch-query path.

Thin ctypes veneer over the C++ mirror in csrc/host/mclab.cpp (declared in
include/mclab.hpp), which itself sits on the C ABI of include/dpf.h.  Names and
argument meaning follow the Scala sources of the reference:

    LSHServer.lshEngine = LSH(conf); LSHServer.isUseDense = True
    allDenseVectors = DensevectorRDFInit.newMultiThreadFit(fileName, conf)
    results = DensevectorRDFInit.NewMultiThreadQueryBatch([0, 1], [DenseVector(0, v1), DenseVector(1, v2)], 0)

(README.md:32-43 of the reference, its "simple test").
"""
import ctypes as C

import numpy as np

from . import _lib


class MclabError(RuntimeError):
    pass


def _L():
    L = _lib.lib()
    if getattr(L, "_mclab_ready", False):
        return L
    ip, dp, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int64)
    L.mclab_last_error.restype = C.c_char_p
    L.mclab_set_resource_root.argtypes = [C.c_char_p]
    L.mclab_LSHServer_set.argtypes = [C.c_char_p, C.c_int]
    L.mclab_LSH_shape.argtypes = [i64p] * 5
    L.mclab_LSH_copy_family.argtypes = [dp, dp, ip, ip]
    L.mclab_LSH_calculateIndex.argtypes = [dp, C.c_int, C.c_int, ip, C.POINTER(C.c_int)]
    L.mclab_LSH_outPutTheHashFunctionsIntoFile.argtypes = [C.c_char_p]
    L.mclab_Dense_fit_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, i64p]
    L.mclab_Dense_fit_array.argtypes = [dp, C.c_int64, C.c_int, C.c_char_p]
    L.mclab_Dense_NewMultiThreadQueryBatch.argtypes = [ip, dp, C.c_int64, C.c_int, C.c_int, C.c_int, i64p, C.POINTER(ip)]
    L.mclab_Dense_NewMultiThreadQueryBatch_insert.argtypes = [dp, C.c_int64, C.c_int, C.c_int, C.c_int, i64p, C.POINTER(ip)]
    L.mclab_Dense_topKAndPrecisionScore.argtypes = [dp, C.c_int64, C.c_int, i64p, ip, C.c_char_p, C.c_int, C.c_int, ip, i64p, dp]
    L.mclab_Dense_getTopKGroundTruth.argtypes = [C.c_char_p, C.c_int, i64p, C.POINTER(ip)]
    L.mclab_Dense_getHtDistribution.argtypes = [dp, C.POINTER(C.c_int)]
    L.mclab_Dense_copy_partitioner.argtypes = [dp, dp, ip]
    L.mclab_Dense_handle.restype = C.c_void_p
    L.mclab_Dense_size.restype = C.c_int64
    L.mclab_Vectors_parseDense.argtypes = [C.c_char_p, ip, dp, C.c_int, C.POINTER(C.c_int)]
    L.mclab_Vectors_fromString.argtypes = [C.c_char_p, ip, ip, ip, dp, C.c_int, C.POINTER(C.c_int)]
    L.mclab_Vectors_analysisKNN.argtypes = [C.c_char_p, C.c_int, ip]
    L.mclab_Vectors_fromPythonString.argtypes = [C.c_char_p, ip, ip, ip, dp, C.c_int, C.POINTER(C.c_int)]
    L.mclab_Sparse_fit_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, i64p]
    L.mclab_Sparse_fit_csr.argtypes = [i64p, ip, dp, C.c_int64, C.c_int, C.c_char_p]
    L.mclab_Sparse_NewMultiThreadQueryBatch_keys.argtypes = [ip, C.c_int64, C.c_int, C.c_int, i64p, C.POINTER(ip)]
    L.mclab_Sparse_NewMultiThreadQueryBatch_insert.argtypes = [i64p, ip, dp, C.c_int64, C.c_int, C.c_int, C.c_int, i64p, C.POINTER(ip)]
    L.mclab_Sparse_query.argtypes = [ip, i64p, ip, dp, C.c_int64, C.c_int, C.c_int, C.c_int, i64p, C.POINTER(ip)]
    L.mclab_Sparse_topKAndPrecisionScore.argtypes = [dp, C.c_int64, C.c_int, i64p, ip, C.c_char_p, C.c_int, C.c_int, ip, i64p, dp, i64p]
    L.mclab_Sparse_copy_partitioner.argtypes = [dp, dp, ip]
    L.mclab_Sparse_handle.restype = C.c_void_p
    L.mclab_Sparse_size.restype = C.c_int64
    L._mclab_ready = True
    return L


def _check(rc):
    if rc != 0:
        raise MclabError(_L().mclab_last_error().decode())


def set_resource_root(path):
    """Directory that plays the JVM classpath's resource root (getClassLoader.getResource)."""
    _L().mclab_set_resource_root(str(path).encode())


class Config:
    """com.typesafe.config.Config for flat dotted keys (TestSettings.scala:6-73)."""

    def __init__(self, entries=None):
        self.entries = dict(entries or {})

    @staticmethod
    def parseString(text):
        c = Config()
        for raw in text.splitlines():
            line = raw.split("#", 1)[0].strip()
            if not line:
                continue
            for sep in ("=", ":"):
                if sep in line:
                    k, v = line.split(sep, 1)
                    break
            else:
                raise MclabError("Config: cannot parse line %r" % raw)
            c.entries[k.strip().strip('"')] = v.strip().strip('"')
        return c

    def withFallback(self, other):
        e = dict(other.entries)
        e.update(self.entries)
        return Config(e)

    def withValue(self, key, value):
        e = dict(self.entries)
        e[key] = value
        return Config(e)

    def _get(self, key):
        if key not in self.entries:
            raise MclabError("No configuration setting found for key '%s'" % key)
        return self.entries[key]

    def getInt(self, key):
        return int(float(self._get(key)))

    def getDouble(self, key):
        return float(self._get(key))

    def getString(self, key):
        return str(self._get(key))

    def getBoolean(self, key):
        return str(self._get(key)).lower() in ("true", "yes", "on")

    def text(self):
        out = []
        for k, v in self.entries.items():
            if isinstance(v, bool):
                v = "true" if v else "false"
            out.append('%s = "%s"' % (k, v))
        return "\n".join(out)


class DenseVector:
    """mclab.lsh.vector.DenseVector (Vector.scala:353-364)."""

    def __init__(self, vectorId, values):
        self.vectorId = int(vectorId)
        self.values = np.ascontiguousarray(values, dtype=np.float64)

    @property
    def size(self):
        return self.values.shape[0]

    def toArray(self):
        return self.values


class SparseVector:
    """mclab.lsh.vector.SparseVector (Vector.scala:374-378): (vectorId, size, indices, values)."""

    def __init__(self, vectorId, size, indices, values):
        self.vectorId, self.size = int(vectorId), int(size)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.values = np.ascontiguousarray(values, dtype=np.float64)
        if self.indices.shape != self.values.shape:
            raise MclabError("requirement failed: indices length: %d, values length: %d" % (self.indices.size, self.values.size))

    def toArray(self):
        d = np.zeros(self.size, dtype=np.float64)
        d[self.indices] = self.values
        return d


class Vectors:
    @staticmethod
    def sparse(vectorId, size, indices, values):
        """Vectors.sparse (Vector.scala:86-88)."""
        return SparseVector(vectorId, size, indices, values)

    @staticmethod
    def fromPythonString(line):
        """Vector.scala:194-208: "[1, 3, [1, 2, 3], [1.0, 2.0, 3.0]]" -> (id, size, indices, values)."""
        cap = line.count(",") + 2
        idx = np.empty(cap, dtype=np.int32)
        vals = np.empty(cap, dtype=np.float64)
        vid, size, nnz = C.c_int32(), C.c_int32(), C.c_int()
        _check(_L().mclab_Vectors_fromPythonString(line.encode(), C.byref(vid), C.byref(size), _lib._ip(idx), _lib._dp(vals),
                                                   cap, C.byref(nnz)))
        return vid.value, size.value, idx[:nnz.value].copy(), vals[:nnz.value].copy()

    @staticmethod
    def parseDense(line):
        """Vector.scala:215-219 -> (vectorId, values)."""
        L = _L()
        cap = line.count(",") + 2
        vals = np.empty(cap, dtype=np.float64)
        vid, n = C.c_int32(), C.c_int()
        _check(L.mclab_Vectors_parseDense(line.encode(), C.byref(vid), _lib._dp(vals), cap, C.byref(n)))
        return vid.value, vals[:n.value].copy()

    @staticmethod
    def fromString(line):
        """Vector.scala:162-175 -> (id, size, indices, values)."""
        L = _L()
        cap = line.count(",") + 2
        idx = np.empty(cap, dtype=np.int32)
        vals = np.empty(cap, dtype=np.float64)
        vid, size, nnz = C.c_int32(), C.c_int32(), C.c_int()
        _check(L.mclab_Vectors_fromString(line.encode(), C.byref(vid), C.byref(size), _lib._ip(idx), _lib._dp(vals),
                                          cap, C.byref(nnz)))
        return vid.value, size.value, idx[:nnz.value].copy(), vals[:nnz.value].copy()

    @staticmethod
    def analysisKNN(line, k):
        """Vector.scala:284-293."""
        out = np.empty(max(k, 1), dtype=np.int32)
        _check(_L().mclab_Vectors_analysisKNN(line.encode(), k, _lib._ip(out)))
        return out[:k].copy()


class LSH:
    """mclab.lsh.LSH (LSH.scala:17-196): construction builds the hash chains on the host;
    calculateIndex runs on the device."""

    def __init__(self, conf):
        self.conf = conf

    def _install(self, use_dense):
        _check(_L().mclab_LSHServer_set(self.conf.text().encode(), 1 if use_dense else 0))

    def family(self):
        """-> dict(fns, b, w, chain_fn, family) of the installed engine."""
        L = _L()
        v = [C.c_int64() for _ in range(5)]
        _check(L.mclab_LSH_shape(*[C.byref(x) for x in v]))
        F, dim, chains, m, fam = [x.value for x in v]
        fns = np.empty((F, dim)); chain_fn = np.empty((chains, m), dtype=np.int32)
        b = np.empty(F) if fam == 1 else None
        w = np.empty(F, dtype=np.int32) if fam == 1 else None
        _check(L.mclab_LSH_copy_family(_lib._dp(fns), _lib._dp(b), _lib._ip(w), _lib._ip(chain_fn)))
        return dict(fns=fns, b=b, w=w, chain_fn=chain_fn, family="pStable" if fam == 1 else "angle")

    def calculateIndex(self, vector, tableId):
        L = _L()
        out = np.empty(4096, dtype=np.int32)
        n = C.c_int()
        vals = vector.values
        _check(L.mclab_LSH_calculateIndex(_lib._dp(vals), vals.shape[0], tableId, _lib._ip(out), C.byref(n)))
        return out[:n.value].copy()

    def outPutTheHashFunctionsIntoFile(self, path):
        _check(_L().mclab_LSH_outPutTheHashFunctionsIntoFile(str(path).encode()))


class _LSHServer:
    """mclab.deploy.LSHServer (LSHServer.scala:5-24): process-wide engine + dense switch."""

    def __init__(self):
        self._engine = None
        self._dense = False

    @property
    def lshEngine(self):
        return self._engine

    @lshEngine.setter
    def lshEngine(self, engine):
        self._engine = engine
        if engine is None:
            _check(_L().mclab_LSHServer_set(None, 1 if self._dense else 0))
        else:
            engine._install(self._dense)

    @property
    def isUseDense(self):
        return self._dense

    @isUseDense.setter
    def isUseDense(self, v):
        self._dense = bool(v)
        if self._engine is not None:
            # only the flag changes; the engine (and its random family) must stay the same object
            L = _L()
            L.mclab_LSHServer_set_dense.argtypes = [C.c_int]
            _check(L.mclab_LSHServer_set_dense(1 if self._dense else 0))


LSHServer = _LSHServer()


def _csr_to_sets(offsets, ids):
    return [set(ids[offsets[i]:offsets[i + 1]].tolist()) for i in range(len(offsets) - 1)]


class _DensevectorRDFInit:
    """mclab.deploy.DensevectorRDFInit (DensevectorRDFInit.scala:19-593)."""

    def _ensure_engine(self, conf):
        if LSHServer.lshEngine is None:              # Dense:128-130,162
            LSHServer.lshEngine = LSH(conf)

    def newMultiThreadFit(self, fileName, conf):
        """Dense:161-206 -> number of vectors indexed (ids are the line numbers)."""
        self._ensure_engine(conf)
        n = C.c_int64()
        _check(_L().mclab_Dense_fit_file(str(fileName).encode(), conf.text().encode(), 1, C.byref(n)))
        return n.value

    def newFastFit(self, fileName, conf):
        """Dense:127-151."""
        self._ensure_engine(conf)
        n = C.c_int64()
        _check(_L().mclab_Dense_fit_file(str(fileName).encode(), conf.text().encode(), 0, C.byref(n)))
        return n.value

    def fitVectors(self, X, conf):
        """Same fit from an in-memory [n, dim] array (ids = row numbers)."""
        self._ensure_engine(conf)
        X = np.ascontiguousarray(X, dtype=np.float64)
        _check(_L().mclab_Dense_fit_array(_lib._dp(X), X.shape[0], X.shape[1], conf.text().encode()))

    @staticmethod
    def _vectors(denseVectorArray):
        if isinstance(denseVectorArray, np.ndarray):
            return np.ascontiguousarray(denseVectorArray, dtype=np.float64)
        return np.ascontiguousarray(np.stack([v.values for v in denseVectorArray]), dtype=np.float64)

    def NewMultiThreadQueryBatch(self, queryArray, denseVectorArray=None, steps=0, queryThreadNum=5):
        """Dense:335-360 -> list of sets of ids.  Called without queryArray ids
        (NewMultiThreadQueryBatch(vectors, steps, threads), Dense:372-399) the vectors are
        first appended to the index and queried under their new ids."""
        L = _L()
        pids = C.POINTER(C.c_int32)()
        if denseVectorArray is None or np.isscalar(denseVectorArray):
            Q = self._vectors(queryArray)
            steps = 0 if denseVectorArray is None else int(denseVectorArray)
            off = np.zeros(Q.shape[0] + 1, dtype=np.int64)
            _check(L.mclab_Dense_NewMultiThreadQueryBatch_insert(_lib._dp(Q), Q.shape[0], Q.shape[1], steps, queryThreadNum,
                                                                 off.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(pids)))
        else:
            qids = np.ascontiguousarray(queryArray, dtype=np.int32)
            nq = qids.shape[0]
            Q = self._vectors(denseVectorArray)[:nq] if nq else np.zeros((0, 1))
            off = np.zeros(nq + 1, dtype=np.int64)
            _check(L.mclab_Dense_NewMultiThreadQueryBatch(_lib._ip(qids), _lib._dp(Q), nq, Q.shape[1], steps, queryThreadNum,
                                                          off.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(pids)))
        total = int(off[-1])
        ids = np.ctypeslib.as_array(pids, shape=(max(total, 1),))[:total].copy()
        _lib.lib().dpf_free(pids)
        return _csr_to_sets(off, ids)

    def query(self, queryKeyArray, querySVArray, steps=0, queryThreadNum=10, queryThreadPool=None):
        """Dense:533-557."""
        return self.NewMultiThreadQueryBatch(queryKeyArray, querySVArray, steps, queryThreadNum)

    def queryBatch(self, queryArray, denseVectorArray, steps, L=None):
        """Dense:311-317."""
        return self.NewMultiThreadQueryBatch(queryArray, denseVectorArray, steps, 1)

    def getTopKGroundTruth(self, filename, K):
        """Dense:440-447 -> list of sets."""
        n = C.c_int64()
        pids = C.POINTER(C.c_int32)()
        _check(_L().mclab_Dense_getTopKGroundTruth(str(filename).encode(), K, C.byref(n), C.byref(pids)))
        flat = np.ctypeslib.as_array(pids, shape=(max(n.value * K, 1),))[:n.value * K].copy()
        _lib.lib().dpf_free(pids)
        return [set(int(x) for x in row if x >= 0) for row in flat.reshape(n.value, K)]

    def topKAndPrecisionScore(self, allDenseVectors, groundTruth, conf, steps=0, queryThreadPool=None):
        """Dense:472-507 -> (list of top-K id lists, recall@K)."""
        V = self._vectors(allDenseVectors)
        nq = len(groundTruth)
        topK = conf.getInt("mclab.lsh.topK")
        off = np.zeros(nq + 1, dtype=np.int64)
        flat = []
        for i, g in enumerate(groundTruth):
            flat.extend(sorted(g))
            off[i + 1] = len(flat)
        gt = np.asarray(flat, dtype=np.int32) if flat else np.zeros(1, dtype=np.int32)
        out = np.empty((max(nq, 1), topK), dtype=np.int32)
        rows, score = C.c_int64(), C.c_double()
        Vq = np.ascontiguousarray(V[:nq])
        _check(_L().mclab_Dense_topKAndPrecisionScore(_lib._dp(Vq), nq, V.shape[1], off.ctypes.data_as(C.POINTER(C.c_int64)),
                                                      _lib._ip(gt), conf.text().encode(), steps, topK, _lib._ip(out),
                                                      C.byref(rows), C.byref(score)))
        return [[int(x) for x in r if x >= 0] for r in out[:rows.value]], score.value

    def getDtAndHtNumDistribution(self):
        """Dense:515-530 -> (data-table distribution, mean ids per sub-index over the LSH tables)."""
        out = np.zeros(256, dtype=np.float64)
        n = C.c_int()
        _check(_L().mclab_Dense_getHtDistribution(_lib._dp(out), C.byref(n)))
        return np.array([float(self.size())]), out[:n.value].copy()

    def partitioner(self):
        """The partitioner functions initializeRDFHashMap generated: (part_fns, part_b, part_w)."""
        conf_L, pb = self._shape
        pf = np.empty((conf_L, pb, 32))
        pbv = np.empty((conf_L, pb))
        pw = np.empty((conf_L, pb), dtype=np.int32)
        _check(_L().mclab_Dense_copy_partitioner(_lib._dp(pf), _lib._dp(pbv), _lib._ip(pw)))
        return pf, pbv, pw

    def set_shape(self, L, pb):
        self._shape = (L, pb)

    def size(self):
        return int(_L().mclab_Dense_size())

    def clearAndClose(self):
        """Dense:453-458."""
        _check(_L().mclab_Dense_clearAndClose())


DensevectorRDFInit = _DensevectorRDFInit()



def _csr_of(vectors):
    """list of SparseVector -> (row offsets, indices, values, dim) for the C veneer."""
    dim = vectors[0].size if vectors else 0
    off = np.zeros(len(vectors) + 1, dtype=np.int64)
    for i, v in enumerate(vectors):
        if v.size != dim:
            raise MclabError("requirement failed: vector1 size: %d, vector2 size: %d" % (dim, v.size))
        off[i + 1] = off[i] + v.indices.shape[0]
    idx = np.concatenate([v.indices for v in vectors]).astype(np.int32) if vectors else np.zeros(1, dtype=np.int32)
    val = np.concatenate([v.values for v in vectors]).astype(np.float64) if vectors else np.zeros(1)
    if idx.size == 0:
        idx, val = np.zeros(1, dtype=np.int32), np.zeros(1)
    return off, np.ascontiguousarray(idx), np.ascontiguousarray(val), dim


class _SparsevectorRDFInit:
    """mclab.deploy.SparsevectorRDFInit (SparsevectorRDFInit.scala:20-593): the same tables fed with SparseVectors.
    LSHServer.isUseDense must be False.  A sparse vector hashes exactly as its dense form does, so the vectors are
    densified at the boundary; what differs from the dense object is the keyed batch query, which looks up the hash of
    the key's stored vector only (RandomDrawTreeMap.getSimilarWithStepWise, RandomDrawTreeMap.java:630-675)."""

    def _ensure_engine(self, conf):
        if LSHServer.lshEngine is None:
            LSHServer.lshEngine = LSH(conf)

    def newMultiThreadFit(self, fileName, conf):
        """Sparse:158-206 -> number of vectors indexed (lines in the Vectors.fromPythonString format)."""
        self._ensure_engine(conf)
        n = C.c_int64()
        _check(_L().mclab_Sparse_fit_file(str(fileName).encode(), conf.text().encode(), 1, C.byref(n)))
        return n.value

    def newFastFit(self, fileName, conf):
        """Sparse:124-151."""
        self._ensure_engine(conf)
        n = C.c_int64()
        _check(_L().mclab_Sparse_fit_file(str(fileName).encode(), conf.text().encode(), 0, C.byref(n)))
        return n.value

    def fitVectors(self, vectors, conf):
        """Same fit from a list of SparseVector (ids = positions)."""
        self._ensure_engine(conf)
        off, idx, val, dim = _csr_of(vectors)
        _check(_L().mclab_Sparse_fit_csr(off.ctypes.data_as(C.POINTER(C.c_int64)), _lib._ip(idx), _lib._dp(val), len(vectors), dim,
                                         conf.text().encode()))

    @staticmethod
    def _sets(call, nq):
        pids = C.POINTER(C.c_int32)()
        off = np.zeros(nq + 1, dtype=np.int64)
        _check(call(off.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(pids)))
        total = int(off[-1])
        ids = np.ctypeslib.as_array(pids, shape=(max(total, 1),))[:total].copy()
        _lib.lib().dpf_free(pids)
        return _csr_to_sets(off, ids)

    def NewMultiThreadQueryBatch(self, queryArray, steps=0, queryThreadNum=5):
        """By key (Sparse:324-350): list of int ids -> the keyed walk; a list of SparseVector (Sparse:360-387) is first
        appended to the index and then looked up under its new ids."""
        L = _L()
        if len(queryArray) and isinstance(queryArray[0], SparseVector):
            off, idx, val, dim = _csr_of(list(queryArray))
            return self._sets(lambda o, p: L.mclab_Sparse_NewMultiThreadQueryBatch_insert(
                off.ctypes.data_as(C.POINTER(C.c_int64)), _lib._ip(idx), _lib._dp(val), len(queryArray), dim, steps, queryThreadNum, o, p),
                len(queryArray))
        keys = np.ascontiguousarray(queryArray, dtype=np.int32)
        return self._sets(lambda o, p: L.mclab_Sparse_NewMultiThreadQueryBatch_keys(_lib._ip(keys), keys.shape[0], steps, queryThreadNum, o, p),
                          keys.shape[0])

    def query(self, queryKeyArray, querySVArray, steps=0, queryThreadNum=10):
        """Sparse:529-555: getSimilarWithStepWiseFaster with the given vectors."""
        keys = np.ascontiguousarray(queryKeyArray, dtype=np.int32)
        off, idx, val, dim = _csr_of(list(querySVArray)[:keys.shape[0]])
        return self._sets(lambda o, p: _L().mclab_Sparse_query(_lib._ip(keys), off.ctypes.data_as(C.POINTER(C.c_int64)), _lib._ip(idx),
                                                               _lib._dp(val), keys.shape[0], dim, steps, queryThreadNum, o, p), keys.shape[0])

    def queryBatch(self, queryArray, sparseVectorArray, steps, L=None):
        """Sparse:300-306."""
        return self.query(queryArray, sparseVectorArray, steps, 1)

    def querySingleKey(self, queryKey, sparseVector, steps=0, L=None):
        """Sparse:278-292."""
        return self.query([queryKey], [sparseVector], steps, 1)[0]

    def topKAndPrecisionScore(self, allDenseVectors, groundTruth, conf, steps=0):
        """Sparse:458-502 -> (list of top-K id lists, recall@K, milliseconds the lookup took)."""
        V = np.ascontiguousarray(np.asarray(allDenseVectors, dtype=np.float64))
        nq = len(groundTruth)
        topK = conf.getInt("mclab.lsh.topK")
        off = np.zeros(nq + 1, dtype=np.int64)
        flat = []
        for i, g in enumerate(groundTruth):
            flat.extend(sorted(g))
            off[i + 1] = len(flat)
        gt = np.asarray(flat, dtype=np.int32) if flat else np.zeros(1, dtype=np.int32)
        out = np.empty((max(nq, 1), topK), dtype=np.int32)
        rows, score, ms = C.c_int64(), C.c_double(), C.c_int64()
        Vq = np.ascontiguousarray(V[:nq])
        _check(_L().mclab_Sparse_topKAndPrecisionScore(_lib._dp(Vq), nq, V.shape[1], off.ctypes.data_as(C.POINTER(C.c_int64)),
                                                       _lib._ip(gt), conf.text().encode(), steps, topK, _lib._ip(out),
                                                       C.byref(rows), C.byref(score), C.byref(ms)))
        return [[int(x) for x in r if x >= 0] for r in out[:rows.value]], score.value, ms.value

    def getTopKGroundTruth(self, filename, K):
        """Sparse:427-434 (the dense object's reader)."""
        return DensevectorRDFInit.getTopKGroundTruth(filename, K)

    def partitioner(self, L, pb):
        """The partitioner functions initializeRDFHashMap generated: (part_fns, part_b, part_w)."""
        pf = np.empty((L, pb, 32))
        pbv = np.empty((L, pb))
        pw = np.empty((L, pb), dtype=np.int32)
        _check(_L().mclab_Sparse_copy_partitioner(_lib._dp(pf), _lib._dp(pbv), _lib._ip(pw)))
        return pf, pbv, pw

    def size(self):
        return int(_L().mclab_Sparse_size())

    def clearAndClose(self):
        """Sparse:440-445."""
        _check(_L().mclab_Sparse_clearAndClose())


SparsevectorRDFInit = _SparsevectorRDFInit()
