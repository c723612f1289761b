// C++ mirror of the reference's host-side objects for the dense batch-query path,
// implemented on top of the C ABI in dpf.h.  Names, argument meaning and error
// behaviour follow the Scala sources (paths relative to the reference checkout):
//
//   mclab::Config                          com.typesafe.config.Config as used by
//                                          src/test/scala/mclab/TestSettings.scala:6-73
//   mclab::lsh::vector::DenseVector        src/main/scala/mclab/lsh/vector/Vector.scala:353-364
//   mclab::lsh::vector::Vectors            Vector.scala:67-346 (parseDense, fromString, analysisKNN, ...)
//   mclab::lsh::LSH                        src/main/scala/mclab/lsh/LSH.scala:17-196
//   mclab::deploy::LSHServer               src/main/scala/mclab/deploy/LSHServer.scala:5-24
//   mclab::deploy::DensevectorRDFInit      src/main/scala/mclab/deploy/DensevectorRDFInit.scala:19-593
//   mclab::deploy::SparsevectorRDFInit     src/main/scala/mclab/deploy/SparsevectorRDFInit.scala:20-593
//
// The reference is a JVM program; its Scala toolchain is not part of this build,
// so the drop-in host layer is C++.  A JNI shim binds the same C ABI.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

struct dpf_index;

namespace mclab {

// Thrown where the reference throws (ConfigException.Missing, require(...), parse errors).
struct MclabError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Flat dotted-key subset of HOCON: `a.b.c = value`, `a.b.c=value`, `a.b.c : value`,
// quoted or bare values, `#` and `//` comments.  (TestSettings.scala only uses this subset.)
class Config {
public:
    Config() = default;
    static Config parseString(const std::string &text);          // ConfigFactory.parseString
    static Config parseFile(const std::string &path);            // ConfigFactory.parseFile
    Config withFallback(const Config &other) const;              // Config.withFallback
    bool hasPath(const std::string &key) const;
    int getInt(const std::string &key) const;                    // throws MclabError when missing / not a number
    double getDouble(const std::string &key) const;
    bool getBoolean(const std::string &key) const;
    std::string getString(const std::string &key) const;
    int getIntOr(const std::string &key, int dflt) const;
    std::string getStringOr(const std::string &key, const std::string &dflt) const;
    const std::map<std::string, std::string> &entries() const { return kv_; }

private:
    std::map<std::string, std::string> kv_;
};

namespace lsh {
namespace vector {

struct DenseVector {                                             // Vector.scala:353
    int vectorId = 0;
    std::vector<double> values;
    DenseVector() = default;
    DenseVector(int id, std::vector<double> v) : vectorId(id), values(std::move(v)) {}
    int size() const { return (int)values.size(); }
    std::string toString() const;                                // "[v0,v1,...]"
};

struct SparseVector {                                            // Vector.scala:374-378
    int vectorId = 0, size = 0;
    std::vector<int> indices;
    std::vector<double> values;
};

struct SparseTuple {                                             // (id, size, indices, values)
    int id = 0, size = 0;
    std::vector<int> indices;
    std::vector<double> values;
};

struct Vectors {
    static std::pair<int, std::vector<double>> parseDense(const std::string &line);   // Vector.scala:215-219
    static SparseTuple fromString(const std::string &line);                           // Vector.scala:162-175
    static SparseTuple fromPythonString(const std::string &line);                     // Vector.scala:194-208
    static std::vector<double> fromStringDense(const std::string &line);              // Vector.scala:182-186
    static std::vector<int> analysisKNN(const std::string &line, int k);              // Vector.scala:284-293
};

}  // namespace vector

// Everything LSH.initHashChains (LSH.scala:29-82) produces, as plain arrays.
struct HashFamilyData {
    int family = 0;                  // 0 angle, 1 pStable
    int dim = 0, chainLength = 0, numChains = 0;
    std::vector<double> fns;         // [F][dim]
    std::vector<double> b;           // [F]
    std::vector<int32_t> w;          // [F]
    std::vector<int32_t> chain_fn;   // [numChains][chainLength]
    int F() const { return dim ? (int)(fns.size() / (size_t)dim) : 0; }
};

class LSH {
public:
    explicit LSH(const Config &conf);                            // LSH.scala:17-53
    ~LSH();
    // LSH.calculateIndex(vector: DenseVector, tableId) (LSH.scala:135-166): tableId < 0 -> every
    // chain's raw hash; tableId >= 0 -> Array(1) with the typeOfIndex transform applied.
    std::vector<int> calculateIndex(const vector::DenseVector &v, int tableId);
    const HashFamilyData &family() const { return fam_; }
    const Config &conf() const { return conf_; }
    const std::string &lshFamilyName() const { return name_; }
    // the text form the reference writes with outPutTheHashFunctionsIntoFile (LSH.scala:173-195)
    void outPutTheHashFunctionsIntoFile(const std::string &path) const;
    // generateTableChainFromFile (AngleHashFamily.scala:158-177, PStableHashFamily.scala:88-108)
    static HashFamilyData loadFamilyFile(const std::string &path, int family, int dim, int chainLength);
    static std::string resolveResource(const std::string &name); // getClassLoader.getResource analogue

private:
    Config conf_;
    std::string name_, typeOfIndex_;
    HashFamilyData fam_;
    dpf_index *probe_ = nullptr;                                 // lazily created engine used by calculateIndex
    dpf_index *probeRaw_ = nullptr;                              // the same chains without the typeOfIndex transform
    dpf_index *ensureProbe(bool raw);
};

}  // namespace lsh

namespace deploy {

struct LSHServer {                                               // LSHServer.scala:5-24
    static std::shared_ptr<lsh::LSH> lshEngine;
    static bool isUseDense;
    static std::shared_ptr<lsh::LSH> getLSHEngine() { return lshEngine; }
    static bool getisUseDense() { return isUseDense; }
};

class DensevectorRDFInit {                                       // a Scala `object`: one process-wide instance
public:
    using DenseVector = lsh::vector::DenseVector;
    static void initializeRDFHashMap(const Config &conf);                                   // Dense:50-118
    static std::vector<DenseVector> newFastFit(const std::string &fileName, const Config &conf);        // Dense:127
    static std::vector<DenseVector> newMultiThreadFit(const std::string &fileName, const Config &conf); // Dense:161
    // not in the reference: the same fit from vectors already in memory (ids = positions)
    static void fitVectors(const std::vector<DenseVector> &vectors, const Config &conf);
    static std::set<int> querySingleKey(int queryKey, const DenseVector &v, int steps = 0, int L = -1); // Dense:284
    static std::vector<std::set<int>> queryBatch(const std::vector<int> &queryArray,
                                                 const std::vector<DenseVector> &vectors, int steps, int L); // Dense:311
    static std::vector<std::set<int>> NewMultiThreadQueryBatch(const std::vector<int> &queryArray,
                                                               const std::vector<DenseVector> &vectors,
                                                               int steps = 0, int queryThreadNum = 5);   // Dense:335
    // Dense:372: inserts the query vectors with fresh ids first, then queries them
    static std::vector<std::set<int>> NewMultiThreadQueryBatch(const std::vector<DenseVector> &vectors,
                                                               int steps, int queryThreadNum);
    static std::vector<std::set<int>> query(const std::vector<int> &queryKeyArray,
                                            const std::vector<DenseVector> &vectors, int steps = 0,
                                            int queryThreadNum = 10);                                      // Dense:533
    static std::vector<std::set<int>> getTopKGroundTruth(const std::string &filename, int K);             // Dense:440
    static std::pair<std::vector<std::vector<int>>, double> topKAndPrecisionScore(
        const std::vector<DenseVector> &allDenseVectors, const std::vector<std::set<int>> &groundTruth,
        const Config &conf, int steps = 0);                                                               // Dense:472
    static void clearAndClose();                                                                          // Dense:453
    static std::pair<std::vector<double>, std::vector<double>> getDtAndHtNumDistribution();              // Dense:515
    // CSR form of the batch result, for bindings that do not want std::set
    static void queryBatchCSR(const int32_t *qids, const double *Q, int64_t nq, int steps,
                              std::vector<int64_t> &offsets, std::vector<int32_t> &ids);
    static dpf_index *handle();
    static int64_t size();
};

// The same tables fed with SparseVectors (SparsevectorRDFInit.scala).  Needs LSHServer::isUseDense == false.
class SparsevectorRDFInit {
public:
    using SparseVector = lsh::vector::SparseVector;
    static void initializeRDFHashMap(const Config &conf);                                                 // Sparse:51
    // both return the stored VALUES of every vector (not its dense form), as the reference does (Sparse:136,179)
    static std::vector<std::vector<double>> newFastFit(const std::string &fileName, const Config &conf);  // Sparse:124
    static std::vector<std::vector<double>> newMultiThreadFit(const std::string &fileName, const Config &conf);   // Sparse:158
    // not in the reference: the same fit from vectors already in memory (ids = positions)
    static void fitVectors(const std::vector<SparseVector> &vectors, const Config &conf);
    // by key: RandomDrawTreeMap.getSimilarWithStepWise(key, steps), the hash of the key's stored vector only (Sparse:324, 407)
    static std::vector<std::set<int>> NewMultiThreadQueryBatch(const std::vector<int> &queryArray, int steps = 0,
                                                               int queryThreadNum = 5);
    // Sparse:360: inserts the query vectors with fresh ids first, then queries them by key
    static std::vector<std::set<int>> NewMultiThreadQueryBatch(const std::vector<SparseVector> &vectors, int steps,
                                                               int queryThreadNum);
    // with the vectors given: getSimilarWithStepWiseFaster, as the dense object (Sparse:529, 574)
    static std::vector<std::set<int>> query(const std::vector<int> &queryKeyArray, const std::vector<SparseVector> &vectors,
                                            int steps = 0, int queryThreadNum = 10);
    static std::set<int> querySingleKey(int queryKey, const SparseVector &v, int steps = 0, int L = -1);          // Sparse:278
    static std::vector<std::set<int>> queryBatch(const std::vector<int> &queryArray, const std::vector<SparseVector> &vectors,
                                                 int steps, int L = -1);                                           // Sparse:300
    static std::vector<std::set<int>> getTopKGroundTruth(const std::string &filename, int K);                     // Sparse:427
    // (top-k ids per answered query, recall, milliseconds the lookup took)                                          Sparse:458
    static std::tuple<std::vector<std::vector<int>>, double, long> topKAndPrecisionScore(
        const std::vector<std::vector<double>> &allDenseVectors, const std::vector<std::set<int>> &groundTruth,
        const Config &conf, int steps = 0);
    static void clearAndClose();                                                                                   // Sparse:440
    static std::pair<std::vector<double>, std::vector<double>> getDtAndHtNumDistribution();                       // Sparse:509
    static dpf_index *handle();
    static int64_t size();
};

}  // namespace deploy
}  // namespace mclab
