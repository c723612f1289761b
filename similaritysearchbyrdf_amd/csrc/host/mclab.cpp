// Host-side mirror of the reference's Scala objects for the dense batch-query
// path (declared in include/mclab.hpp), implemented over the C ABI of dpf.h.
#include "../../../include/mclab.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <chrono>
#include <sstream>
#include <tuple>

#include "../../../include/dpf.h"

namespace mclab {

// ---------------------------------------------------------------------------
// Config
// ---------------------------------------------------------------------------

static std::string trim(const std::string &s)
{
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

Config Config::parseString(const std::string &text)
{
    Config c;
    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
        // strip comments outside quotes
        bool inq = false;
        size_t cut = line.size();
        for (size_t i = 0; i < line.size(); i++) {
            if (line[i] == '"') inq = !inq;
            if (!inq && (line[i] == '#' || (line[i] == '/' && i + 1 < line.size() && line[i + 1] == '/'))) { cut = i; break; }
        }
        line = trim(line.substr(0, cut));
        if (line.empty()) continue;
        size_t sep = line.find_first_of("=:");
        if (sep == std::string::npos) throw MclabError("Config: cannot parse line '" + line + "'");
        std::string key = trim(line.substr(0, sep)), val = trim(line.substr(sep + 1));
        if (!val.empty() && val.back() == ',') val = trim(val.substr(0, val.size() - 1));
        if (val.size() >= 2 && val.front() == '"' && val.back() == '"') val = val.substr(1, val.size() - 2);
        if (key.size() >= 2 && key.front() == '"' && key.back() == '"') key = key.substr(1, key.size() - 2);
        if (key.empty()) throw MclabError("Config: empty key in '" + line + "'");
        c.kv_[key] = val;
    }
    return c;
}

Config Config::parseFile(const std::string &path)
{
    std::ifstream f(path);
    if (!f) throw MclabError("Config: cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return parseString(ss.str());
}

Config Config::withFallback(const Config &other) const
{
    Config c = *this;
    for (const auto &e : other.kv_)
        if (!c.kv_.count(e.first)) c.kv_[e.first] = e.second;
    return c;
}

bool Config::hasPath(const std::string &key) const { return kv_.count(key) != 0; }

std::string Config::getString(const std::string &key) const
{
    auto it = kv_.find(key);
    if (it == kv_.end()) throw MclabError("No configuration setting found for key '" + key + "'");
    return it->second;
}

int Config::getInt(const std::string &key) const
{
    std::string v = getString(key);
    char *end = nullptr;
    double d = strtod(v.c_str(), &end);
    if (end == v.c_str() || *end) throw MclabError(key + " has type STRING rather than NUMBER");
    return (int)d;
}

double Config::getDouble(const std::string &key) const
{
    std::string v = getString(key);
    char *end = nullptr;
    double d = strtod(v.c_str(), &end);
    if (end == v.c_str() || *end) throw MclabError(key + " has type STRING rather than NUMBER");
    return d;
}

bool Config::getBoolean(const std::string &key) const
{
    std::string v = getString(key);
    if (v == "true" || v == "yes" || v == "on") return true;
    if (v == "false" || v == "no" || v == "off") return false;
    throw MclabError(key + " has type STRING rather than BOOLEAN");
}

int Config::getIntOr(const std::string &key, int dflt) const { return hasPath(key) ? getInt(key) : dflt; }
std::string Config::getStringOr(const std::string &key, const std::string &dflt) const
{
    return hasPath(key) ? getString(key) : dflt;
}

// ---------------------------------------------------------------------------
// Vectors
// ---------------------------------------------------------------------------
namespace lsh {
namespace vector {

std::string DenseVector::toString() const
{
    std::ostringstream o;
    o.precision(17);
    o << "[";
    for (size_t i = 0; i < values.size(); i++) o << (i ? "," : "") << values[i];
    o << "]";
    return o.str();
}

static std::string strip_chars(const std::string &s, const char *chars)
{
    std::string o;
    o.reserve(s.size());
    for (char c : s)
        if (!strchr(chars, c)) o.push_back(c);
    return o;
}

static std::vector<std::string> split(const std::string &s, char sep)
{
    std::vector<std::string> out;
    size_t a = 0;
    for (;;) {
        size_t b = s.find(sep, a);
        if (b == std::string::npos) { out.push_back(s.substr(a)); break; }
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    // String.split drops trailing empty strings
    while (!out.empty() && out.back().empty()) out.pop_back();
    return out;
}

static int to_int(const std::string &s)
{
    char *end = nullptr;
    long v = strtol(s.c_str(), &end, 10);
    if (end == s.c_str() || *end) throw MclabError("NumberFormatException: For input string: \"" + s + "\"");
    return (int)v;
}

static double to_double(const std::string &s)
{
    std::string t = trim(s);
    char *end = nullptr;
    double v = strtod(t.c_str(), &end);
    if (end == t.c_str() || *end) throw MclabError("NumberFormatException: For input string: \"" + s + "\"");
    return v;
}

// Vector.scala:215-219: drop blanks and brackets, split on ',', first token is the id
std::pair<int, std::vector<double>> Vectors::parseDense(const std::string &line)
{
    std::string s = strip_chars(line, " []\r\n");
    std::vector<std::string> parts = split(s, ',');
    if (parts.empty()) throw MclabError("cannot parse " + line);
    std::vector<double> v;
    v.reserve(parts.size() - 1);
    for (size_t i = 1; i < parts.size(); i++) v.push_back(to_double(parts[i]));
    return {to_int(parts[0]), std::move(v)};
}

// Vector.scala:162-175: "(id,size,[i0,i1,...],[v0,v1,...])"
SparseTuple Vectors::fromString(const std::string &line_)
{
    std::string line = strip_chars(line_, "\r\n");
    std::vector<std::string> p;
    size_t a = 0;
    for (;;) {
        size_t b = line.find(",[", a);
        if (b == std::string::npos) { p.push_back(line.substr(a)); break; }
        p.push_back(line.substr(a, b - a));
        a = b + 2;
    }
    if (p.size() != 3) throw MclabError("cannot parse " + line);
    SparseTuple t;
    std::vector<std::string> head = split(strip_chars(p[0], "("), ',');
    if (head.size() != 2) throw MclabError("cannot parse " + line);
    t.id = to_int(trim(head[0]));
    t.size = to_int(trim(head[1]));
    for (const std::string &s : split(strip_chars(p[1], "]"), ','))
        if (!s.empty()) t.indices.push_back(to_int(trim(s)));
    std::string vals = p[2];
    size_t close = vals.find("])");
    if (close != std::string::npos) vals.erase(close, 2);
    for (const std::string &s : split(vals, ','))
        if (!s.empty()) t.values.push_back(to_double(s));
    return t;
}

// "[1, 3, [1, 2, 3], [1.0, 2.0, 3.0]]": spaces dropped, split on ",[" into id/size, indices, values
SparseTuple Vectors::fromPythonString(const std::string &line_)            // Vector.scala:194-208
{
    const std::string line = strip_chars(line_, " \r\n");
    std::vector<std::string> p;
    size_t a = 0;
    for (;;) {
        size_t b = line.find(",[", a);
        if (b == std::string::npos) { p.push_back(line.substr(a)); break; }
        p.push_back(line.substr(a, b - a));
        a = b + 2;
    }
    if (p.size() != 3) throw MclabError("cannot parse " + line_);
    SparseTuple t;
    std::vector<std::string> head = split(strip_chars(p[0], "["), ',');
    if (head.size() != 2) throw MclabError("cannot parse " + line_);
    t.id = to_int(head[0]);
    t.size = to_int(head[1]);
    for (const std::string &s : split(strip_chars(p[1], "]"), ','))
        if (!s.empty()) t.indices.push_back(to_int(s));
    std::string vals = p[2];
    size_t close = vals.find("]]");
    if (close != std::string::npos) vals.erase(close, 2);
    for (const std::string &s : split(vals, ','))
        if (!s.empty()) t.values.push_back(to_double(s));
    return t;
}

std::vector<double> Vectors::fromStringDense(const std::string &line)      // Vector.scala:182-186
{
    std::vector<double> v;
    for (const std::string &s : split(strip_chars(line, "\r\n"), ',')) v.push_back(to_double(s));
    return v;
}

std::vector<int> Vectors::analysisKNN(const std::string &line, int k)      // Vector.scala:284-293
{
    std::vector<std::string> parts = split(strip_chars(line, " \r\n"), ',');
    if (k > (int)parts.size()) throw MclabError("cannot parse " + line);
    std::vector<int> out;
    for (int i = 0; i < k; i++) out.push_back(to_int(strip_chars(parts[(size_t)i], "[]")));
    return out;
}

}  // namespace vector

// ---------------------------------------------------------------------------
// LSH
// ---------------------------------------------------------------------------

static std::string g_resource_root;

std::string LSH::resolveResource(const std::string &name)
{
    // getClass.getClassLoader.getResource(name): relative to the resource root when one is
    // set (environment DPF_RESOURCE_ROOT or mclab_set_resource_root), else the path as given
    std::string root = g_resource_root;
    if (root.empty()) { const char *e = getenv("DPF_RESOURCE_ROOT"); if (e) root = e; }
    if (!name.empty() && name[0] == '/') return name;
    if (!root.empty()) {
        std::string p = root + "/" + name;
        std::ifstream f(p);
        if (f) return p;
    }
    return name;
}

static int type_of_index_code(const std::string &s)
{
    if (s == "original") return DPF_INDEX_ORIGINAL;
    if (s == "sampling") return DPF_INDEX_SAMPLING;
    if (s == "continueBitsCount") return DPF_INDEX_CONTINUE_BITS;
    if (s == "angleNewMethod") return DPF_INDEX_ANGLE_NEW;
    throw MclabError("scala.MatchError: " + s + " (mclab.lsh.typeOfIndex)");   // LSH.scala:152-162 has no default case
}

// java.util.Random(seed) + Scala 2.10 Random.shuffle(0 until 32): Sampling.scala:6-11 (seed 88387 at LSH.scala:21)
static void scala_shuffle_0_31(int64_t seed0, int32_t out[32])
{
    uint64_t seed = ((uint64_t)seed0 ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
    auto next = [&](int bits) {
        seed = (seed * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
        return (int32_t)((int64_t)seed >> (48 - bits));
    };
    auto next_int = [&](int32_t n) {
        if ((n & -n) == n) return (int32_t)(((int64_t)n * (int64_t)next(31)) >> 31);
        int32_t bits, val;
        do { bits = next(31); val = bits % n; } while ((int32_t)((uint32_t)bits - (uint32_t)val + (uint32_t)(n - 1)) < 0);
        return val;
    };
    for (int i = 0; i < 32; i++) out[i] = i;
    for (int n = 32; n >= 2; n--) { int32_t k = next_int(n); std::swap(out[n - 1], out[k]); }
}

// thin Q (m x n, m >= n) of a Householder QR, row-major
static std::vector<double> thin_q(std::vector<double> a, int m, int n)
{
    std::vector<std::vector<double>> vs;
    for (int k = 0; k < n; k++) {
        std::vector<double> v((size_t)m, 0.0);
        double norm = 0;
        for (int i = k; i < m; i++) { v[(size_t)i] = a[(size_t)i * n + k]; norm += v[(size_t)i] * v[(size_t)i]; }
        norm = std::sqrt(norm);
        if (norm > 0) {
            v[(size_t)k] += (v[(size_t)k] >= 0 ? norm : -norm);
            double vn = 0;
            for (int i = k; i < m; i++) vn += v[(size_t)i] * v[(size_t)i];
            vn = std::sqrt(vn);
            for (int i = k; i < m; i++) v[(size_t)i] /= vn;
            for (int j = k; j < n; j++) {
                double s = 0;
                for (int i = k; i < m; i++) s += v[(size_t)i] * a[(size_t)i * n + j];
                for (int i = k; i < m; i++) a[(size_t)i * n + j] -= 2 * s * v[(size_t)i];
            }
        }
        vs.push_back(std::move(v));
    }
    std::vector<double> q((size_t)m * n, 0.0);
    for (int j = 0; j < n; j++) q[(size_t)j * n + j] = 1.0;
    for (int k = n - 1; k >= 0; k--) {
        const std::vector<double> &v = vs[(size_t)k];
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int i = k; i < m; i++) s += v[(size_t)i] * q[(size_t)i * n + j];
            for (int i = k; i < m; i++) q[(size_t)i * n + j] -= 2 * s * v[(size_t)i];
        }
    }
    return q;
}

HashFamilyData LSH::loadFamilyFile(const std::string &path, int family, int dim, int chainLength)
{
    std::ifstream f(path);
    if (!f) throw MclabError("cannot open hash family file " + path);
    HashFamilyData d;
    d.family = family; d.dim = dim; d.chainLength = chainLength;
    std::string line;
    int n = 0;
    while (std::getline(f, line)) {
        line = trim(line);
        if (line.empty()) continue;
        std::string vec = line;
        if (family == DPF_FAMILY_PSTABLE) {
            // PStableHashFamily.scala:93: "vector;b;w"
            std::vector<std::string> p = vector::split(line, ';');
            if (p.size() != 3) throw MclabError("scala.MatchError: cannot split '" + line + "' into vector;b;w");
            vec = p[0];
            d.b.push_back(vector::to_double(p[1]));
            d.w.push_back(vector::to_int(trim(p[2])));
        }
        vector::SparseTuple t = vector::Vectors::fromString(vec);
        if (t.indices.size() != t.values.size())
            throw MclabError("requirement failed: indices length: " + std::to_string(t.indices.size()) + ", values length: " + std::to_string(t.values.size()));
        // SimilarityCalculator.scala:43,46 zips the stored VALUES with the dense vector: only a
        // function that lists every index 0..dim-1 means what its file says
        if (t.size != dim) throw MclabError("requirement failed: vector1 size: " + std::to_string(t.size) + ", vector2 size: " + std::to_string(dim));
        if ((int)t.values.size() != dim) throw MclabError("hash function " + std::to_string(t.id) + " in " + path + " does not list all " + std::to_string(dim) + " components (the dense path multiplies values positionally)");
        for (int i = 0; i < dim; i++)
            if (t.indices[(size_t)i] != i) throw MclabError("hash function " + std::to_string(t.id) + " in " + path + ": indices must be 0.." + std::to_string(dim - 1));
        d.fns.insert(d.fns.end(), t.values.begin(), t.values.end());
        n++;
    }
    // paraSetList.grouped(chainLength): a short last group becomes a short chain, which
    // PStableHashChain's require (PStableHashFamily.scala:120) rejects; reject it for both
    if (n == 0 || n % chainLength != 0) throw MclabError("requirement failed: " + std::to_string(n) + " functions do not form chains of " + std::to_string(chainLength));
    d.numChains = n / chainLength;
    d.chain_fn.resize((size_t)n);
    for (int i = 0; i < n; i++) d.chain_fn[(size_t)i] = i;
    return d;
}

LSH::LSH(const Config &conf) : conf_(conf)
{
    name_ = conf.getString("mclab.lsh.name");                                 // LSH.scala:19
    typeOfIndex_ = conf.getString("mclab.lsh.typeOfIndex");                   // LSH.scala:22
    const bool generateByPulling = conf.getBoolean("mclab.lsh.generateByPulling");
    const bool isOrthogonal = conf.getBoolean("mclab.lsh.IsOrthogonal");
    const int familySize = conf.getInt("mclab.lsh.familySize");               // LSH.scala:30-33
    const int vectorDim = conf.getInt("mclab.lsh.vectorDim");
    const int chainLength = conf.getInt("mclab.lshTable.chainLength");
    const int permutationNum = conf.getInt("mclab.lsh.permutationNum");
    type_of_index_code(typeOfIndex_);
    if (name_ != "angle" && name_ != "pStable") { fam_ = HashFamilyData(); return; }   // LSH.scala:46: no chains
    const int family = name_ == "angle" ? DPF_FAMILY_ANGLE : DPF_FAMILY_PSTABLE;
    const int tableNum = conf.getInt("mclab.lsh.tableNum");                   // LSH.scala:65-67
    const std::string method = conf.getString("mclab.lsh.generateMethod");
    const std::string confType = conf.getString("mclab.confType");
    if (vectorDim < 1 || chainLength < 1 || tableNum < 1) throw MclabError("vectorDim, chainLength and tableNum must be positive");
    if (method == "fromfile") {
        std::string key = confType == "partition" ? "mclab.lsh.partitionFamilyFilePath" : "mclab.lsh.familyFilePath";
        if (confType != "lsh" && confType != "partition") throw MclabError("NullPointerException: no hash chains for confType " + confType);
        fam_ = loadFamilyFile(resolveResource(conf.getString(key)), family, vectorDim, chainLength);
        return;
    }
    if (method != "default") throw MclabError("NullPointerException: no hash chains for generateMethod " + method);   // LSH.scala:78-80

    std::mt19937_64 rng((uint64_t)conf.getIntOr("mclab.gpu.seed", 88387) * 0x9E3779B97F4A7C15ULL + (uint64_t)vectorDim);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    fam_.family = family; fam_.dim = vectorDim; fam_.chainLength = chainLength;
    auto new_unit_vector = [&](std::vector<double> &out) {                   // AngleHashFamily.scala:37-51
        std::vector<double> v((size_t)vectorDim);
        double sq = 0;
        for (double &x : v) { x = uni(rng); if (rng() & 1) x = -x; sq += x * x; }
        sq = std::sqrt(sq);
        for (double x : v) out.push_back(x / sq);
    };
    if (family == DPF_FAMILY_ANGLE) {
        if (familySize < 1 || permutationNum < 1) throw MclabError("familySize and permutationNum must be positive");
        if (generateByPulling) {
            if (isOrthogonal) {                                              // AngleHashFamily.scala:73-85
                if (familySize < vectorDim) throw MclabError("IndexOutOfBoundsException: orthogonal family needs familySize >= vectorDim");
                std::vector<double> a((size_t)familySize * vectorDim);
                for (double &x : a) x = uni(rng);
                fam_.fns = thin_q(a, familySize, vectorDim);
            } else {
                for (int i = 0; i < familySize; i++) new_unit_vector(fam_.fns);
            }
        }
        fam_.numChains = tableNum * permutationNum;                          // AngleHashFamily.scala:122
        for (int t = 0; t < tableNum; t++) {
            std::vector<int32_t> chain((size_t)chainLength);
            for (int j = 0; j < chainLength; j++) {
                if (generateByPulling) chain[(size_t)j] = (int32_t)(rng() % (uint64_t)familySize);
                else { chain[(size_t)j] = (int32_t)(fam_.fns.size() / (size_t)vectorDim); new_unit_vector(fam_.fns); }
            }
            for (int p = 0; p < permutationNum; p++) {                       // AngleHashFamily.scala:143-146
                std::vector<int32_t> sh = chain;
                std::shuffle(sh.begin(), sh.end(), rng);
                fam_.chain_fn.insert(fam_.chain_fn.end(), sh.begin(), sh.end());
            }
        }
    } else {
        const double mu = conf.getDouble("mclab.lsh.family.pstable.mu");     // LSH.scala:40-42
        const double sigma = conf.getDouble("mclab.lsh.family.pstable.sigma");
        const int w = conf.getInt("mclab.lsh.family.pstable.w");
        if (familySize < 1 || w == 0) throw MclabError("familySize must be positive and w non-zero");
        std::normal_distribution<double> gauss(mu, sigma);
        for (int i = 0; i < familySize; i++) {                               // PStableHashFamily.scala:37-57
            for (int k = 0; k < vectorDim; k++) fam_.fns.push_back(gauss(rng));
            fam_.b.push_back(uni(rng) * w);
            fam_.w.push_back(w);
        }
        fam_.numChains = tableNum;                                           // PStableHashFamily.scala:68
        for (int t = 0; t < tableNum; t++)
            for (int j = 0; j < chainLength; j++) fam_.chain_fn.push_back((int32_t)(rng() % (uint64_t)familySize));
    }
}

LSH::~LSH()
{
    if (probe_) dpf_destroy(probe_);
    if (probeRaw_) dpf_destroy(probeRaw_);
}

dpf_index *LSH::ensureProbe(bool raw)
{
    dpf_index *&probe_ = raw ? this->probeRaw_ : this->probe_;
    if (probe_) return probe_;
    if (fam_.numChains == 0) throw MclabError("IndexOutOfBoundsException: no hash chains (mclab.lsh.name = " + name_ + ")");
    dpf_config c{};
    c.family = fam_.family; c.dim = fam_.dim; c.num_functions = fam_.F(); c.num_tables = fam_.numChains;
    c.chain_length = fam_.chainLength; c.partition_bits = 0; c.bucket_bits = 28; c.dir_node_size = 32;
    c.bucket_overflow = 500; c.type_of_index = raw ? DPF_INDEX_ORIGINAL : type_of_index_code(typeOfIndex_); c.top_k = 1; c.dtype = DPF_F64;
    c.device = conf_.getIntOr("mclab.gpu.device", 0); c.row_begin = 0; c.row_end = -1;
    c.flags = DPF_FLAG_HASH_ONLY;      // no forest behind this handle: any chain length hashes
    int rc = dpf_create(&c, &probe_);
    if (rc != DPF_OK) throw MclabError(std::string("dpf_create: ") + dpf_last_error(nullptr));
    rc = dpf_set_hash_family(probe_, fam_.fns.data(), fam_.b.empty() ? nullptr : fam_.b.data(),
                             fam_.w.empty() ? nullptr : fam_.w.data(), fam_.chain_fn.data());
    if (rc == DPF_OK && c.type_of_index == DPF_INDEX_SAMPLING) {
        int32_t si[32];
        scala_shuffle_0_31(88387, si);
        rc = dpf_set_sampling_index(probe_, si);
    }
    if (rc != DPF_OK) { std::string m = dpf_last_error(probe_); dpf_destroy(probe_); probe_ = nullptr; throw MclabError(m); }
    return probe_;
}

std::vector<int> LSH::calculateIndex(const vector::DenseVector &v, int tableId)
{
    if (v.size() != fam_.dim) throw MclabError("requirement failed: vector1 size: " + std::to_string(fam_.dim) + ", vector2 size: " + std::to_string(v.size()));
    if (tableId >= fam_.numChains) throw MclabError("IndexOutOfBoundsException: " + std::to_string(tableId));
    // LSH.scala:136-138: tableId < 0 = every chain's value WITHOUT the typeOfIndex transform
    dpf_index *probe = ensureProbe(tableId < 0 && typeOfIndex_ != "original");
    std::vector<int32_t> h((size_t)fam_.numChains);
    if (dpf_hash_batch(probe, v.values.data(), 1, h.data(), nullptr) != DPF_OK) throw MclabError(dpf_last_error(probe));
    if (tableId >= 0) return {h[(size_t)tableId]};
    return std::vector<int>(h.begin(), h.end());
}

void LSH::outPutTheHashFunctionsIntoFile(const std::string &path) const
{
    std::ofstream f(path);
    if (!f) throw MclabError("cannot write " + path);
    f.precision(17);
    int id = 0;
    for (int c = 0; c < fam_.numChains; c++)
        for (int j = 0; j < fam_.chainLength; j++) {
            int fn = fam_.chain_fn[(size_t)c * fam_.chainLength + j];
            f << "(" << id++ << "," << fam_.dim << ",[";
            for (int i = 0; i < fam_.dim; i++) f << (i ? "," : "") << i;
            f << "],[";
            for (int i = 0; i < fam_.dim; i++) f << (i ? "," : "") << fam_.fns[(size_t)fn * fam_.dim + i];
            f << "])";
            if (fam_.family == DPF_FAMILY_PSTABLE) f << ";" << fam_.b[(size_t)fn] << ";" << fam_.w[(size_t)fn];
            f << "\r\n";
        }
}

}  // namespace lsh

// ---------------------------------------------------------------------------
// deploy
// ---------------------------------------------------------------------------
namespace deploy {

std::shared_ptr<lsh::LSH> LSHServer::lshEngine;
bool LSHServer::isUseDense = false;

namespace {

struct State {
    dpf_index *ix = nullptr;
    bool createFlag = false;
    int tableNum = 0, permutationNum = 0, L = 0, pb = 0, dim = 0;
    bool store_f32 = false;
    int64_t n = 0;
    std::vector<double> part_fns, part_b;      // what initializeRDFHashMap gave the partitioners
    std::vector<int32_t> part_w;
    std::vector<double> rows;                  // SparsevectorRDFInit: the fitted vectors, dense, for its keyed queries
};
State S;     // DensevectorRDFInit's tables
State T;     // SparsevectorRDFInit's tables (the reference keeps the two objects apart too)

void check(const State &st, int rc, const char *what)
{
    if (rc != DPF_OK) throw MclabError(std::string(what) + ": " + (st.ix ? dpf_last_error(st.ix) : dpf_last_error(nullptr)));
}
void check(int rc, const char *what) { check(S, rc, what); }

int self_mode(const std::string &s)
{
    if (s == "compat") return DPF_SELF_COMPAT;
    if (s == "always") return DPF_SELF_ALWAYS;
    if (s == "never") return DPF_SELF_NEVER;
    throw MclabError("mclab.gpu.excludeSelf must be compat, always or never");
}

std::vector<std::set<int>> to_sets(const std::vector<int64_t> &off, const std::vector<int32_t> &ids)
{
    std::vector<std::set<int>> out(off.size() - 1);
    for (size_t q = 0; q + 1 < off.size(); q++) out[q].insert(ids.begin() + off[q], ids.begin() + off[q + 1]);
    return out;
}

std::vector<double> flatten(const std::vector<lsh::vector::DenseVector> &v, size_t count, int dim)
{
    std::vector<double> flat(count * (size_t)dim);
    for (size_t i = 0; i < count; i++) {
        if (v[i].size() != dim) throw MclabError("requirement failed: vector1 size: " + std::to_string(dim) + ", vector2 size: " + std::to_string(v[i].size()));
        std::copy(v[i].values.begin(), v[i].values.end(), flat.begin() + i * (size_t)dim);
    }
    return flat;
}

void fit_rows(State &st, const double *X, int64_t n)
{
    if (st.store_f32) {
        std::vector<float> f((size_t)n * st.dim);
        for (size_t i = 0; i < f.size(); i++) f[i] = (float)X[i];
        check(st, dpf_fit(st.ix, f.data(), n), "dpf_fit");
    } else {
        check(st, dpf_fit(st.ix, X, n), "dpf_fit");
    }
    st.n += n;
}
void fit_rows(const double *X, int64_t n) { fit_rows(S, X, n); }

}  // namespace

void detail_partitioner(std::vector<double> &pf, std::vector<double> &pb, std::vector<int32_t> &pw, bool sparse)
{
    const State &st = sparse ? T : S;
    pf = st.part_fns; pb = st.part_b; pw = st.part_w;
}

dpf_index *DensevectorRDFInit::handle() { return S.ix; }
int64_t DensevectorRDFInit::size() { return S.n; }

static void init_index(State &S, const Config &conf)
{
    if (!LSHServer::lshEngine) throw MclabError("NullPointerException: LSHServer.lshEngine is null");
    if (S.ix) dpf_destroy(S.ix);
    S = State();
    const lsh::LSH &engine = *LSHServer::lshEngine;
    const lsh::HashFamilyData &fam = engine.family();
    const int tableNum = conf.getInt("mclab.lsh.tableNum");                  // Dense:52-58
    (void)conf.getString("mclab.lsh.workingDirRoot");
    (void)conf.getInt("mclab.lsh.ramThreshold");
    (void)conf.getInt("mclab.dataTable.numPartitions");
    const int partitionBits = conf.getInt("mclab.lsh.partitionBits");
    const int permutationNum = conf.getInt("mclab.lsh.permutationNum");
    (void)conf.getString("mclab.lsh.featureDataFormat");
    const int L = tableNum * permutationNum;                                 // Dense:107
    if (fam.numChains < L)
        // LSH.scala:154 indexes tableIndexGenerators(tableId): pStable.pick and both file loaders
        // return fewer than tableNum*permutationNum chains when permutationNum > 1
        throw MclabError("IndexOutOfBoundsException: the LSH engine holds " + std::to_string(fam.numChains) +
                         " hash chains but tableNum*permutationNum = " + std::to_string(L) + " tables were requested");
    dpf_config c{};
    c.family = fam.family; c.dim = fam.dim; c.num_functions = fam.F(); c.num_tables = L;
    c.chain_length = conf.getInt("mclab.lshTable.chainLength");              // Dense:39-42 (setupTable)
    c.partition_bits = partitionBits;
    c.bucket_bits = conf.getInt("mclab.lshTable.bucketBits");
    c.dir_node_size = conf.getInt("mclab.lshTable.dirNodeSize");
    c.bucket_overflow = conf.getInt("mclab.lshTable.bufferOverflow");
    c.type_of_index = lsh::type_of_index_code(conf.getString("mclab.lsh.typeOfIndex"));
    c.exclude_self = self_mode(conf.getStringOr("mclab.gpu.excludeSelf", "compat"));
    std::string metric = conf.getStringOr("mclab.gpu.metric", "dot");
    if (metric != "dot" && metric != "l2") throw MclabError("mclab.gpu.metric must be dot or l2");
    c.metric = metric == "dot" ? DPF_METRIC_DOT : DPF_METRIC_L2;
    c.top_k = conf.getIntOr("mclab.lsh.topK", 10);
    S.store_f32 = conf.getStringOr("mclab.gpu.storeFloat32", "false") == "true";
    c.dtype = S.store_f32 ? DPF_F32 : DPF_F64;
    c.device = conf.getIntOr("mclab.gpu.device", 0);
    c.row_begin = 0; c.row_end = -1;
    if (c.chain_length != fam.chainLength) throw MclabError("mclab.lshTable.chainLength differs from the LSH engine's chain length");
    check(S, dpf_create(&c, &S.ix), "dpf_create");
    std::vector<int32_t> chains(fam.chain_fn.begin(), fam.chain_fn.begin() + (size_t)L * fam.chainLength);
    check(S, dpf_set_hash_family(S.ix, fam.fns.data(), fam.b.empty() ? nullptr : fam.b.data(),
                              fam.w.empty() ? nullptr : fam.w.data(), chains.data()), "dpf_set_hash_family");
    if (partitionBits > 0) {
        // Dense:63-77: every table gets its own LocalitySensitivePartitioner built from the main
        // conf overlaid with vectorDim=32, tableNum=1, chainLength=partitionBits, generateMethod=default
        std::vector<double> pf((size_t)L * partitionBits * 32), pbv;
        std::vector<int32_t> pw;
        for (int t = 0; t < L; t++) {
            std::ostringstream o;
            o << "mclab.confType=partition\nmclab.lsh.vectorDim=32\nmclab.lsh.tableNum = 1\n"
              << "mclab.lshTable.chainLength=" << partitionBits << "\nmclab.lsh.generateMethod=default\n"
              << "mclab.gpu.seed=" << (conf.getIntOr("mclab.gpu.seed", 88387) + 7919 * (t + 1)) << "\n";
            lsh::LSH part(Config::parseString(o.str()).withFallback(conf));
            const lsh::HashFamilyData &pfam = part.family();
            for (int j = 0; j < partitionBits; j++) {                       // chain 0 (Part:59)
                int fn = pfam.chain_fn[(size_t)j];
                std::copy(pfam.fns.begin() + (size_t)fn * 32, pfam.fns.begin() + (size_t)(fn + 1) * 32,
                          pf.begin() + ((size_t)t * partitionBits + j) * 32);
                if (pfam.family == DPF_FAMILY_PSTABLE) { pbv.push_back(pfam.b[(size_t)fn]); pw.push_back(pfam.w[(size_t)fn]); }
            }
        }
        check(S, dpf_set_partitioner(S.ix, pf.data(), pbv.empty() ? nullptr : pbv.data(), pw.empty() ? nullptr : pw.data()), "dpf_set_partitioner");
        S.part_fns = pf; S.part_b = pbv; S.part_w = pw;
    }
    if (c.type_of_index == DPF_INDEX_SAMPLING) {
        int32_t si[32];
        lsh::scala_shuffle_0_31(88387, si);
        check(S, dpf_set_sampling_index(S.ix, si), "dpf_set_sampling_index");
    }
    S.createFlag = true;
    S.tableNum = tableNum; S.permutationNum = permutationNum; S.L = L; S.pb = partitionBits; S.dim = fam.dim;
    S.n = 0;
}

void DensevectorRDFInit::initializeRDFHashMap(const Config &conf)
{
    if (!LSHServer::lshEngine) throw MclabError("NullPointerException: LSHServer.lshEngine is null");
    if (!LSHServer::isUseDense)
        // Hasher.scala:47-51 would cast the DenseVector to SparseVector
        throw MclabError("ClassCastException: LSHServer.isUseDense must be true for DensevectorRDFInit");
    init_index(S, conf);
}

static std::vector<lsh::vector::DenseVector> fit_file(const std::string &fileName, const Config &conf)
{
    if (!LSHServer::lshEngine) LSHServer::lshEngine = std::make_shared<lsh::LSH>(conf);   // Dense:128-130,162
    DensevectorRDFInit::initializeRDFHashMap(conf);
    std::cout << "Finish initialize the RDF." << std::endl;
    std::string path = lsh::LSH::resolveResource(fileName);
    std::ifstream f(path);
    if (!f) throw MclabError("NullPointerException: resource " + fileName + " not found");
    std::vector<lsh::vector::DenseVector> all;
    std::vector<double> flat;
    std::string line;
    int count = 0;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        auto tmp = lsh::vector::Vectors::parseDense(line);
        if ((int)tmp.second.size() != S.dim) throw MclabError("requirement failed: vector1 size: " + std::to_string(S.dim) + ", vector2 size: " + std::to_string(tmp.second.size()));
        flat.insert(flat.end(), tmp.second.begin(), tmp.second.end());
        all.emplace_back(tmp.first, std::move(tmp.second));   // the file's id; the index id is the line counter (Dense:181,190)
        count++;
        if (count % 10000 == 0) std::cout << count << " objects loaded" << std::endl;
    }
    fit_rows(flat.data(), count);
    std::cout << "finish load, totally " << count << " objects." << std::endl;
    return all;
}

std::vector<lsh::vector::DenseVector> DensevectorRDFInit::newFastFit(const std::string &fileName, const Config &conf)
{
    return fit_file(fileName, conf);
}

std::vector<lsh::vector::DenseVector> DensevectorRDFInit::newMultiThreadFit(const std::string &fileName, const Config &conf)
{
    (void)conf.getInt("mclab.insertThreadNum");                              // Dense:168 (the device replaces the thread pool)
    return fit_file(fileName, conf);
}

void DensevectorRDFInit::fitVectors(const std::vector<DenseVector> &vectors, const Config &conf)
{
    if (!LSHServer::lshEngine) LSHServer::lshEngine = std::make_shared<lsh::LSH>(conf);
    initializeRDFHashMap(conf);
    std::vector<double> flat = flatten(vectors, vectors.size(), S.dim);
    fit_rows(flat.data(), (int64_t)vectors.size());
}

static void query_csr(State &S, const int32_t *qids, const double *Q, int64_t nq, int steps,
                      std::vector<int64_t> &offsets, std::vector<int32_t> &ids);
void DensevectorRDFInit::queryBatchCSR(const int32_t *qids, const double *Q, int64_t nq, int steps,
                                       std::vector<int64_t> &offsets, std::vector<int32_t> &ids)
{
    query_csr(S, qids, Q, nq, steps, offsets, ids);
}
static void query_csr(State &S, const int32_t *qids, const double *Q, int64_t nq, int steps,
                      std::vector<int64_t> &offsets, std::vector<int32_t> &ids)
{
    offsets.assign((size_t)nq + 1, 0);
    ids.clear();
    if (!S.ix || S.n == 0) {
        // Dense:422-424: NullPointerException -> "need to fit the data first", empty result sets
        for (int64_t i = 0; i < nq; i++) std::cout << "need to fit the data first" << std::endl;
        return;
    }
    const int32_t *out = nullptr;                        // the handle's own buffer: copied at once
    int rc;
    if (S.store_f32) {
        std::vector<float> f((size_t)nq * S.dim);
        for (size_t i = 0; i < f.size(); i++) f[i] = (float)Q[i];
        rc = dpf_query_batch_view(S.ix, qids, f.data(), nq, steps, offsets.data(), &out);
    } else {
        rc = dpf_query_batch_view(S.ix, qids, Q, nq, steps, offsets.data(), &out);
    }
    check(S, rc, "dpf_query_batch_view");
    ids.assign(out, out + offsets[(size_t)nq]);
}

std::vector<std::set<int>> DensevectorRDFInit::NewMultiThreadQueryBatch(const std::vector<int> &queryArray,
                                                                        const std::vector<DenseVector> &vectors,
                                                                        int steps, int queryThreadNum)
{
    (void)queryThreadNum;   // Dense:341-350 splits tables over a thread pool; results do not depend on it
    if (vectors.size() < queryArray.size()) throw MclabError("ArrayIndexOutOfBoundsException: " + std::to_string(vectors.size()));
    const size_t nq = queryArray.size();
    std::vector<int64_t> off;
    std::vector<int32_t> ids;
    if (!S.ix || S.n == 0) { queryBatchCSR(nullptr, nullptr, (int64_t)nq, steps, off, ids); return to_sets(off, ids); }
    std::vector<double> flat = flatten(vectors, nq, S.dim);
    std::vector<int32_t> q(queryArray.begin(), queryArray.end());
    queryBatchCSR(q.data(), flat.data(), (int64_t)nq, steps, off, ids);
    return to_sets(off, ids);
}

std::vector<std::set<int>> DensevectorRDFInit::NewMultiThreadQueryBatch(const std::vector<DenseVector> &vectors,
                                                                        int steps, int queryThreadNum)
{
    // Dense:372-374 + 215-251: the query vectors are appended to the index with ids
    // size, size+1, ... and those ids are the query keys
    if (!S.ix) throw MclabError("NullPointerException: index not initialised");
    std::vector<double> flat = flatten(vectors, vectors.size(), S.dim);
    std::vector<int> queryArray(vectors.size());
    for (size_t i = 0; i < vectors.size(); i++) queryArray[i] = (int)(S.n + (int64_t)i);
    fit_rows(flat.data(), (int64_t)vectors.size());
    std::cout << "finish load, dataTable totally has " << S.n << " objects." << std::endl;
    return NewMultiThreadQueryBatch(queryArray, vectors, steps, queryThreadNum);
}

std::vector<std::set<int>> DensevectorRDFInit::query(const std::vector<int> &queryKeyArray,
                                                     const std::vector<DenseVector> &vectors, int steps,
                                                     int queryThreadNum)
{
    return NewMultiThreadQueryBatch(queryKeyArray, vectors, steps, queryThreadNum);
}

std::set<int> DensevectorRDFInit::querySingleKey(int queryKey, const DenseVector &v, int steps, int L)
{
    if (L >= 0 && L != S.L) throw MclabError("querySingleKey over a prefix of the tables (L != tableNum*permutationNum) is not offered on the device");
    return NewMultiThreadQueryBatch(std::vector<int>{queryKey}, std::vector<DenseVector>{v}, steps, 1)[0];
}

std::vector<std::set<int>> DensevectorRDFInit::queryBatch(const std::vector<int> &queryArray,
                                                          const std::vector<DenseVector> &vectors, int steps, int L)
{
    if (L >= 0 && L != S.L) throw MclabError("queryBatch over a prefix of the tables (L != tableNum*permutationNum) is not offered on the device");
    return NewMultiThreadQueryBatch(queryArray, vectors, steps, 1);
}

std::vector<std::set<int>> DensevectorRDFInit::getTopKGroundTruth(const std::string &filename, int K)
{
    std::ifstream f(lsh::LSH::resolveResource(filename));
    if (!f) throw MclabError("NullPointerException: resource " + filename + " not found");
    std::vector<std::set<int>> out;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        std::vector<int> knn = lsh::vector::Vectors::analysisKNN(line, K);
        out.emplace_back(knn.begin(), knn.end());
    }
    return out;
}

std::pair<std::vector<std::vector<int>>, double> DensevectorRDFInit::topKAndPrecisionScore(
    const std::vector<DenseVector> &allDenseVectors, const std::vector<std::set<int>> &groundTruth,
    const Config &conf, int steps)
{
    (void)conf.getInt("mclab.queryThreadNum");                               // Dense:476
    const int topK = conf.getInt("mclab.lsh.topK");                          // Dense:490
    const size_t nq = groundTruth.size();
    if (allDenseVectors.size() < nq) throw MclabError("ArrayIndexOutOfBoundsException: " + std::to_string(allDenseVectors.size()));
    std::vector<std::vector<int>> allTopK;
    if (!S.ix || S.n == 0) {
        for (size_t i = 0; i < nq; i++) std::cout << "need to fit the data first" << std::endl;
        return {allTopK, 0.0};
    }
    if (topK < 1 || topK > 64) throw MclabError("mclab.lsh.topK must be in [1, 64]");
    std::vector<double> flat = flatten(allDenseVectors, nq, S.dim);
    std::vector<int32_t> qids(nq);
    for (size_t i = 0; i < nq; i++) qids[i] = (int32_t)i;                    // Dense:473 groundTruth.indices
    std::vector<int32_t> ids(nq * (size_t)topK);
    std::vector<double> scores(nq * (size_t)topK);
    int rc;
    if (S.store_f32) {
        std::vector<float> f(flat.begin(), flat.end());
        rc = dpf_query_topk(S.ix, qids.data(), f.data(), (int64_t)nq, steps, topK, ids.data(), scores.data());
    } else {
        rc = dpf_query_topk(S.ix, qids.data(), flat.data(), (int64_t)nq, steps, topK, ids.data(), scores.data());
    }
    check(rc, "dpf_query_topk");
    double averageScore = 0.0;
    for (size_t i = 0; i < nq; i++) {
        std::vector<int> one;
        double score = 0.0;
        for (int wv = 0; wv < topK; wv++) {
            int id = ids[i * (size_t)topK + wv];
            if (id < 0) break;
            one.push_back(id);
            if (groundTruth[i].count(id)) score += 1;
        }
        if (one.empty()) continue;                                           // Dense:479: empty result sets are skipped
        allTopK.push_back(std::move(one));
        averageScore += score / (double)nq;
    }
    return {allTopK, averageScore / topK};
}

void DensevectorRDFInit::clearAndClose()
{
    if (S.ix) dpf_destroy(S.ix);
    S = State();
}

std::pair<std::vector<double>, std::vector<double>> DensevectorRDFInit::getDtAndHtNumDistribution()
{
    // Dense:515-530; the data table is a flat array here, so its distribution is one bucket
    std::vector<double> dt{(double)S.n}, ht;
    if (S.ix) {
        ht.assign((size_t)1 << S.pb, 0.0);
        check(dpf_subindex_distribution(S.ix, ht.data()), "dpf_subindex_distribution");
    }
    return {dt, ht};
}

// ---------------------------------------------------------------------------
// SparsevectorRDFInit (src/main/scala/mclab/deploy/SparsevectorRDFInit.scala): the same tables fed with
// SparseVectors.  A sparse vector hashes as its dense form does: the sparse x sparse dot product
// (SimilarityCalculator.scala:9-27) walks the common indices in ascending order and an absent index only
// leaves out a term a_i * 0, which changes no fp64 partial sum.  So the vectors are densified at this
// boundary and the dense engine does the rest; what differs is which walk a query takes (below).
// ---------------------------------------------------------------------------
using lsh::vector::SparseVector;

static std::vector<double> densify(const SparseVector &v, int dim)
{
    if (v.size != dim) throw MclabError("requirement failed: vector1 size: " + std::to_string(dim) + ", vector2 size: " + std::to_string(v.size));
    if (v.indices.size() != v.values.size())
        throw MclabError("requirement failed: indices length: " + std::to_string(v.indices.size()) + ", values length: " + std::to_string(v.values.size()));
    std::vector<double> d((size_t)dim, 0.0);
    for (size_t i = 0; i < v.indices.size(); i++) {
        if (v.indices[i] < 0 || v.indices[i] >= dim) throw MclabError("ArrayIndexOutOfBoundsException: " + std::to_string(v.indices[i]));
        d[(size_t)v.indices[i]] = v.values[i];
    }
    return d;
}

dpf_index *SparsevectorRDFInit::handle() { return T.ix; }
int64_t SparsevectorRDFInit::size() { return T.n; }

void SparsevectorRDFInit::initializeRDFHashMap(const Config &conf)
{
    if (!LSHServer::lshEngine) throw MclabError("NullPointerException: LSHServer.lshEngine is null");
    if (LSHServer::isUseDense)
        // Hasher.scala:47-51 would cast the SparseVector to DenseVector
        throw MclabError("ClassCastException: LSHServer.isUseDense must be false for SparsevectorRDFInit");
    init_index(T, conf);
}

static void sparse_fit(const std::vector<SparseVector> &vectors)
{
    std::vector<double> flat;
    flat.reserve(vectors.size() * (size_t)T.dim);
    for (const SparseVector &v : vectors) { std::vector<double> d = densify(v, T.dim); flat.insert(flat.end(), d.begin(), d.end()); }
    fit_rows(T, flat.data(), (int64_t)vectors.size());
    T.rows.insert(T.rows.end(), flat.begin(), flat.end());
}

static std::vector<std::vector<double>> sparse_fit_file(const std::string &fileName, const Config &conf)
{
    if (!LSHServer::lshEngine) LSHServer::lshEngine = std::make_shared<lsh::LSH>(conf);      // Sparse:125-127,159
    SparsevectorRDFInit::initializeRDFHashMap(conf);
    std::cout << "Finish initialize the RDF." << std::endl;
    std::ifstream f(lsh::LSH::resolveResource(fileName));
    if (!f) throw MclabError("NullPointerException: resource " + fileName + " not found");
    std::vector<SparseVector> all;
    std::vector<std::vector<double>> values;
    std::string line;
    while (std::getline(f, line)) {
        if (line.empty()) continue;
        lsh::vector::SparseTuple t = lsh::vector::Vectors::fromPythonString(line);       // Sparse:134,175
        SparseVector v;
        v.vectorId = t.id; v.size = t.size; v.indices = t.indices; v.values = t.values;
        values.push_back(t.values);                      // Sparse:136,179: the stored values, not the dense form
        all.push_back(std::move(v));
        if (all.size() % 10000 == 0) std::cout << all.size() << " objects loaded" << std::endl;
    }
    sparse_fit(all);
    std::cout << "finish load, totally " << all.size() << " objects." << std::endl;
    return values;
}

std::vector<std::vector<double>> SparsevectorRDFInit::newFastFit(const std::string &fileName, const Config &conf)
{
    return sparse_fit_file(fileName, conf);
}

std::vector<std::vector<double>> SparsevectorRDFInit::newMultiThreadFit(const std::string &fileName, const Config &conf)
{
    (void)conf.getInt("mclab.insertThreadNum");                                          // Sparse:160
    return sparse_fit_file(fileName, conf);
}

void SparsevectorRDFInit::fitVectors(const std::vector<SparseVector> &vectors, const Config &conf)
{
    if (!LSHServer::lshEngine) LSHServer::lshEngine = std::make_shared<lsh::LSH>(conf);
    initializeRDFHashMap(conf);
    sparse_fit(vectors);
}

// Sparse:324-350 -> QueryTask.query (Sparse:402-420) -> RandomDrawTreeMap.getSimilarWithStepWise(key, steps)
// (RandomDrawTreeMap.java:630-675): the key's stored vector is hashed and only that hash is looked up, in every
// sub-index within `steps`.  A key the index does not hold makes the reference's hash() fail on a null vector; the
// worker catches the NullPointerException, prints "need to fit the data first" and leaves that query's set empty.
std::vector<std::set<int>> SparsevectorRDFInit::NewMultiThreadQueryBatch(const std::vector<int> &queryArray, int steps,
                                                                         int queryThreadNum)
{
    (void)queryThreadNum;
    const size_t nq = queryArray.size();
    std::vector<std::set<int>> out(nq);
    if (!T.ix || T.n == 0) { for (size_t i = 0; i < nq; i++) std::cout << "need to fit the data first" << std::endl; return out; }
    std::vector<int32_t> keys;
    std::vector<size_t> where;
    std::vector<double> flat;
    for (size_t i = 0; i < nq; i++) {
        const int key = queryArray[i];
        if (key < 0 || key >= T.n) { std::cout << "need to fit the data first" << std::endl; continue; }
        keys.push_back(key);
        where.push_back(i);
        flat.insert(flat.end(), T.rows.begin() + (size_t)key * T.dim, T.rows.begin() + (size_t)(key + 1) * T.dim);
    }
    if (keys.empty()) return out;
    std::vector<int64_t> off;
    std::vector<int32_t> ids;
    check(T, dpf_set_probe_mode(T.ix, DPF_PROBE_OWN), "dpf_set_probe_mode");
    try { query_csr(T, keys.data(), flat.data(), (int64_t)keys.size(), steps, off, ids); }
    catch (...) { dpf_set_probe_mode(T.ix, DPF_PROBE_FLIPS); throw; }
    check(T, dpf_set_probe_mode(T.ix, DPF_PROBE_FLIPS), "dpf_set_probe_mode");
    for (size_t j = 0; j < keys.size(); j++) out[where[j]].insert(ids.begin() + off[j], ids.begin() + off[j + 1]);
    return out;
}

// Sparse:360-387: the query vectors are put into the index first (ids size, size+1, ...), then looked up by key
std::vector<std::set<int>> SparsevectorRDFInit::NewMultiThreadQueryBatch(const std::vector<SparseVector> &vectors, int steps,
                                                                         int queryThreadNum)
{
    if (!T.ix) throw MclabError("NullPointerException: index not initialised");
    std::vector<int> queryArray(vectors.size());
    for (size_t i = 0; i < vectors.size(); i++) queryArray[i] = (int)(T.n + (int64_t)i);
    sparse_fit(vectors);
    std::cout << "finish load, dataTable totally has " << T.n << " objects." << std::endl;
    return NewMultiThreadQueryBatch(queryArray, steps, queryThreadNum);
}

// Sparse:529-555 -> QueryTaskNew (Sparse:569-586): getSimilarWithStepWiseFaster with the given vectors, as the dense object
std::vector<std::set<int>> SparsevectorRDFInit::query(const std::vector<int> &queryKeyArray, const std::vector<SparseVector> &vectors,
                                                      int steps, int queryThreadNum)
{
    (void)queryThreadNum;
    if (vectors.size() < queryKeyArray.size()) throw MclabError("ArrayIndexOutOfBoundsException: " + std::to_string(vectors.size()));
    const size_t nq = queryKeyArray.size();
    std::vector<int64_t> off;
    std::vector<int32_t> ids;
    if (!T.ix || T.n == 0) { query_csr(T, nullptr, nullptr, (int64_t)nq, steps, off, ids); return to_sets(off, ids); }
    std::vector<double> flat;
    for (size_t i = 0; i < nq; i++) { std::vector<double> d = densify(vectors[i], T.dim); flat.insert(flat.end(), d.begin(), d.end()); }
    std::vector<int32_t> q(queryKeyArray.begin(), queryKeyArray.end());
    query_csr(T, q.data(), flat.data(), (int64_t)nq, steps, off, ids);
    return to_sets(off, ids);
}

std::set<int> SparsevectorRDFInit::querySingleKey(int queryKey, const SparseVector &v, int steps, int L)     // Sparse:278
{
    if (L >= 0 && L != T.L) throw MclabError("querySingleKey over a prefix of the tables (L != tableNum*permutationNum) is not offered on the device");
    return query(std::vector<int>{queryKey}, std::vector<SparseVector>{v}, steps, 1)[0];
}

std::vector<std::set<int>> SparsevectorRDFInit::queryBatch(const std::vector<int> &queryArray, const std::vector<SparseVector> &vectors,
                                                           int steps, int L)                                   // Sparse:300
{
    if (L >= 0 && L != T.L) throw MclabError("queryBatch over a prefix of the tables (L != tableNum*permutationNum) is not offered on the device");
    return query(queryArray, vectors, steps, 1);
}

// Sparse:458-502: the queries are the first groundTruth.length rows of allDenseVectors, looked up with
// getSimilarWithStepWiseFaster (Sparse:468 -> query); top-k by descending inner product; the third value is the
// time the lookup took, in milliseconds
std::tuple<std::vector<std::vector<int>>, double, long> SparsevectorRDFInit::topKAndPrecisionScore(
    const std::vector<std::vector<double>> &allDenseVectors, const std::vector<std::set<int>> &groundTruth, const Config &conf, int steps)
{
    (void)conf.getInt("mclab.queryThreadNum");
    const int topK = conf.getInt("mclab.lsh.topK");
    const size_t nq = groundTruth.size();
    if (allDenseVectors.size() < nq) throw MclabError("ArrayIndexOutOfBoundsException: " + std::to_string(allDenseVectors.size()));
    std::vector<std::vector<int>> allTopK;
    if (!T.ix || T.n == 0) { for (size_t i = 0; i < nq; i++) std::cout << "need to fit the data first" << std::endl; return std::make_tuple(allTopK, 0.0, 0L); }
    if (topK < 1 || topK > 64) throw MclabError("mclab.lsh.topK must be in [1, 64]");
    std::vector<double> flat(nq * (size_t)T.dim);
    for (size_t i = 0; i < nq; i++) {
        if ((int)allDenseVectors[i].size() != T.dim) throw MclabError("requirement failed: vector1 size: " + std::to_string(T.dim) + ", vector2 size: " + std::to_string(allDenseVectors[i].size()));
        std::copy(allDenseVectors[i].begin(), allDenseVectors[i].end(), flat.begin() + i * (size_t)T.dim);
    }
    std::vector<int32_t> qids(nq), ids(nq * (size_t)topK);
    for (size_t i = 0; i < nq; i++) qids[i] = (int32_t)i;
    std::vector<double> scores(nq * (size_t)topK);
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    if (T.store_f32) {
        std::vector<float> f(flat.begin(), flat.end());
        rc = dpf_query_topk(T.ix, qids.data(), f.data(), (int64_t)nq, steps, topK, ids.data(), scores.data());
    } else {
        rc = dpf_query_topk(T.ix, qids.data(), flat.data(), (int64_t)nq, steps, topK, ids.data(), scores.data());
    }
    const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
    check(T, rc, "dpf_query_topk");
    double averageScore = 0.0;
    for (size_t i = 0; i < nq; i++) {
        std::vector<int> one;
        double score = 0.0;
        for (int wv = 0; wv < topK; wv++) {
            const int id = ids[i * (size_t)topK + wv];
            if (id < 0) break;
            one.push_back(id);
            if (groundTruth[i].count(id)) score += 1;
        }
        if (one.empty()) continue;
        allTopK.push_back(std::move(one));
        averageScore += score / (double)nq;
    }
    return std::make_tuple(allTopK, averageScore / topK, ms);
}

std::vector<std::set<int>> SparsevectorRDFInit::getTopKGroundTruth(const std::string &filename, int K)
{
    return DensevectorRDFInit::getTopKGroundTruth(filename, K);                           // Sparse:427-434, same body
}

void SparsevectorRDFInit::clearAndClose()
{
    if (T.ix) dpf_destroy(T.ix);
    T = State();
}

std::pair<std::vector<double>, std::vector<double>> SparsevectorRDFInit::getDtAndHtNumDistribution()
{
    std::vector<double> dt{(double)T.n}, ht;
    if (T.ix) {
        ht.assign((size_t)1 << T.pb, 0.0);
        check(T, dpf_subindex_distribution(T.ix, ht.data()), "dpf_subindex_distribution");
    }
    return {dt, ht};
}

}  // namespace deploy
}  // namespace mclab

// ---------------------------------------------------------------------------
// extern "C" veneer over the mirror (what a ctypes / JNI binding calls)
// ---------------------------------------------------------------------------
namespace {
std::string g_err;
template <typename F> int guarded(F &&f)
{
    try { f(); return 0; }
    catch (const std::exception &e) { g_err = e.what(); return -1; }
}
}  // namespace

using namespace mclab;
using mclab::deploy::DensevectorRDFInit;
using mclab::deploy::LSHServer;
using mclab::lsh::vector::DenseVector;

extern "C" {

const char *mclab_last_error(void) { return g_err.c_str(); }

int mclab_set_resource_root(const char *dir) { lsh::g_resource_root = dir ? dir : ""; return 0; }

int mclab_LSHServer_set(const char *conf_text, int isUseDense)
{
    return guarded([&] {
        LSHServer::lshEngine = conf_text ? std::make_shared<lsh::LSH>(Config::parseString(conf_text)) : nullptr;
        LSHServer::isUseDense = isUseDense != 0;
    });
}

int mclab_LSHServer_set_dense(int isUseDense) { LSHServer::isUseDense = isUseDense != 0; return 0; }

int mclab_LSH_shape(int64_t *F, int64_t *dim, int64_t *chains, int64_t *chainLength, int64_t *family)
{
    return guarded([&] {
        if (!LSHServer::lshEngine) throw MclabError("LSHServer.lshEngine is null");
        const lsh::HashFamilyData &f = LSHServer::lshEngine->family();
        *F = f.F(); *dim = f.dim; *chains = f.numChains; *chainLength = f.chainLength; *family = f.family;
    });
}

int mclab_LSH_copy_family(double *fns, double *b, int32_t *w, int32_t *chain_fn)
{
    return guarded([&] {
        if (!LSHServer::lshEngine) throw MclabError("LSHServer.lshEngine is null");
        const lsh::HashFamilyData &f = LSHServer::lshEngine->family();
        std::copy(f.fns.begin(), f.fns.end(), fns);
        if (b) std::copy(f.b.begin(), f.b.end(), b);
        if (w) std::copy(f.w.begin(), f.w.end(), w);
        std::copy(f.chain_fn.begin(), f.chain_fn.end(), chain_fn);
    });
}

int mclab_LSH_calculateIndex(const double *v, int dim, int tableId, int32_t *out, int *n_out)
{
    return guarded([&] {
        if (!LSHServer::lshEngine) throw MclabError("LSHServer.lshEngine is null");
        std::vector<int> r = LSHServer::lshEngine->calculateIndex(DenseVector(0, std::vector<double>(v, v + dim)), tableId);
        std::copy(r.begin(), r.end(), out);
        *n_out = (int)r.size();
    });
}

int mclab_LSH_outPutTheHashFunctionsIntoFile(const char *path)
{
    return guarded([&] {
        if (!LSHServer::lshEngine) throw MclabError("LSHServer.lshEngine is null");
        LSHServer::lshEngine->outPutTheHashFunctionsIntoFile(path);
    });
}

int mclab_Dense_fit_file(const char *fileName, const char *conf_text, int multiThread, int64_t *n)
{
    return guarded([&] {
        Config conf = Config::parseString(conf_text);
        auto all = multiThread ? DensevectorRDFInit::newMultiThreadFit(fileName, conf) : DensevectorRDFInit::newFastFit(fileName, conf);
        *n = (int64_t)all.size();
    });
}

int mclab_Dense_fit_array(const double *X, int64_t n, int dim, const char *conf_text)
{
    return guarded([&] {
        std::vector<DenseVector> v;
        v.reserve((size_t)n);
        for (int64_t i = 0; i < n; i++) v.emplace_back((int)i, std::vector<double>(X + i * dim, X + (i + 1) * dim));
        DensevectorRDFInit::fitVectors(v, Config::parseString(conf_text));
    });
}

static int32_t *copy_out(const std::vector<int32_t> &ids)
{
    int32_t *p = (int32_t *)malloc(sizeof(int32_t) * std::max<size_t>(ids.size(), 1));
    std::copy(ids.begin(), ids.end(), p);
    return p;
}

int mclab_Dense_NewMultiThreadQueryBatch(const int32_t *qids, const double *Q, int64_t nq, int dim, int steps,
                                         int queryThreadNum, int64_t *offsets, int32_t **ids)
{
    (void)queryThreadNum; (void)dim;
    return guarded([&] {
        std::vector<int64_t> off;
        std::vector<int32_t> out;
        DensevectorRDFInit::queryBatchCSR(qids, Q, nq, steps, off, out);
        std::copy(off.begin(), off.end(), offsets);
        *ids = copy_out(out);
    });
}

int mclab_Dense_NewMultiThreadQueryBatch_insert(const double *Q, int64_t nq, int dim, int steps, int queryThreadNum,
                                                int64_t *offsets, int32_t **ids)
{
    return guarded([&] {
        std::vector<DenseVector> v;
        for (int64_t i = 0; i < nq; i++) v.emplace_back((int)i, std::vector<double>(Q + i * dim, Q + (i + 1) * dim));
        auto sets = DensevectorRDFInit::NewMultiThreadQueryBatch(v, steps, queryThreadNum);
        std::vector<int32_t> out;
        offsets[0] = 0;
        for (size_t q = 0; q < sets.size(); q++) { out.insert(out.end(), sets[q].begin(), sets[q].end()); offsets[q + 1] = (int64_t)out.size(); }
        *ids = copy_out(out);
    });
}

int mclab_Dense_topKAndPrecisionScore(const double *vectors, int64_t nq, int dim, const int64_t *gt_off,
                                      const int32_t *gt_ids, const char *conf_text, int steps, int topK,
                                      int32_t *topk_ids, int64_t *rows, double *score)
{
    return guarded([&] {
        std::vector<DenseVector> v;
        std::vector<std::set<int>> gt((size_t)nq);
        for (int64_t i = 0; i < nq; i++) {
            v.emplace_back((int)i, std::vector<double>(vectors + i * dim, vectors + (i + 1) * dim));
            gt[(size_t)i].insert(gt_ids + gt_off[i], gt_ids + gt_off[i + 1]);
        }
        auto r = DensevectorRDFInit::topKAndPrecisionScore(v, gt, Config::parseString(conf_text), steps);
        *rows = (int64_t)r.first.size();
        for (size_t i = 0; i < r.first.size(); i++)
            for (int j = 0; j < topK; j++) topk_ids[i * (size_t)topK + j] = j < (int)r.first[i].size() ? r.first[i][(size_t)j] : -1;
        *score = r.second;
    });
}

int mclab_Dense_getTopKGroundTruth(const char *file, int K, int64_t *n, int32_t **ids)
{
    return guarded([&] {
        auto gt = DensevectorRDFInit::getTopKGroundTruth(file, K);
        *n = (int64_t)gt.size();
        std::vector<int32_t> flat;
        for (auto &s : gt) { std::vector<int32_t> row(s.begin(), s.end()); row.resize((size_t)K, -1); flat.insert(flat.end(), row.begin(), row.end()); }
        *ids = copy_out(flat);
    });
}

int mclab_Dense_clearAndClose(void) { return guarded([] { DensevectorRDFInit::clearAndClose(); }); }

int mclab_Dense_getHtDistribution(double *out, int *n)
{
    return guarded([&] {
        auto r = DensevectorRDFInit::getDtAndHtNumDistribution();
        std::copy(r.second.begin(), r.second.end(), out);
        *n = (int)r.second.size();
    });
}

int mclab_Dense_copy_partitioner(double *part_fns, double *part_b, int32_t *part_w)
{
    return guarded([&] {
        std::vector<double> pf, pb;
        std::vector<int32_t> pw;
        mclab::deploy::detail_partitioner(pf, pb, pw, false);
        std::copy(pf.begin(), pf.end(), part_fns);
        if (part_b) std::copy(pb.begin(), pb.end(), part_b);
        if (part_w) std::copy(pw.begin(), pw.end(), part_w);
    });
}

int mclab_Sparse_copy_partitioner(double *part_fns, double *part_b, int32_t *part_w)
{
    return guarded([&] {
        std::vector<double> pf, pb;
        std::vector<int32_t> pw;
        mclab::deploy::detail_partitioner(pf, pb, pw, true);
        std::copy(pf.begin(), pf.end(), part_fns);
        if (part_b) std::copy(pb.begin(), pb.end(), part_b);
        if (part_w) std::copy(pw.begin(), pw.end(), part_w);
    });
}

void *mclab_Dense_handle(void) { return DensevectorRDFInit::handle(); }
int64_t mclab_Dense_size(void) { return DensevectorRDFInit::size(); }

// ---- SparsevectorRDFInit: vectors cross this boundary as CSR (row offsets, indices, values) ----
static std::vector<mclab::lsh::vector::SparseVector> csr_vectors(const int64_t *row_off, const int32_t *indices, const double *values,
                                                                 int64_t n, int dim)
{
    std::vector<mclab::lsh::vector::SparseVector> v((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        v[(size_t)i].vectorId = (int)i; v[(size_t)i].size = dim;
        v[(size_t)i].indices.assign(indices + row_off[i], indices + row_off[i + 1]);
        v[(size_t)i].values.assign(values + row_off[i], values + row_off[i + 1]);
    }
    return v;
}
static void sets_out(const std::vector<std::set<int>> &sets, int64_t *offsets, int32_t **ids)
{
    std::vector<int32_t> out;
    offsets[0] = 0;
    for (size_t q = 0; q < sets.size(); q++) { out.insert(out.end(), sets[q].begin(), sets[q].end()); offsets[q + 1] = (int64_t)out.size(); }
    *ids = copy_out(out);
}

int mclab_Sparse_fit_file(const char *fileName, const char *conf_text, int multiThread, int64_t *n)
{
    using mclab::deploy::SparsevectorRDFInit;
    return guarded([&] {
        Config conf = Config::parseString(conf_text);
        auto all = multiThread ? SparsevectorRDFInit::newMultiThreadFit(fileName, conf) : SparsevectorRDFInit::newFastFit(fileName, conf);
        *n = (int64_t)all.size();
    });
}

int mclab_Sparse_fit_csr(const int64_t *row_off, const int32_t *indices, const double *values, int64_t n, int dim, const char *conf_text)
{
    return guarded([&] { mclab::deploy::SparsevectorRDFInit::fitVectors(csr_vectors(row_off, indices, values, n, dim), Config::parseString(conf_text)); });
}

int mclab_Sparse_NewMultiThreadQueryBatch_keys(const int32_t *keys, int64_t nq, int steps, int queryThreadNum, int64_t *offsets, int32_t **ids)
{
    return guarded([&] {
        std::vector<int> q(keys, keys + nq);
        sets_out(mclab::deploy::SparsevectorRDFInit::NewMultiThreadQueryBatch(q, steps, queryThreadNum), offsets, ids);
    });
}

int mclab_Sparse_NewMultiThreadQueryBatch_insert(const int64_t *row_off, const int32_t *indices, const double *values, int64_t nq, int dim,
                                                 int steps, int queryThreadNum, int64_t *offsets, int32_t **ids)
{
    return guarded([&] {
        sets_out(mclab::deploy::SparsevectorRDFInit::NewMultiThreadQueryBatch(csr_vectors(row_off, indices, values, nq, dim), steps, queryThreadNum),
                 offsets, ids);
    });
}

int mclab_Sparse_query(const int32_t *keys, const int64_t *row_off, const int32_t *indices, const double *values, int64_t nq, int dim,
                       int steps, int queryThreadNum, int64_t *offsets, int32_t **ids)
{
    return guarded([&] {
        std::vector<int> q(keys, keys + nq);
        sets_out(mclab::deploy::SparsevectorRDFInit::query(q, csr_vectors(row_off, indices, values, nq, dim), steps, queryThreadNum), offsets, ids);
    });
}

int mclab_Sparse_topKAndPrecisionScore(const double *vectors, int64_t nq, int dim, const int64_t *gt_off, const int32_t *gt_ids,
                                       const char *conf_text, int steps, int topK, int32_t *topk_ids, int64_t *rows, double *score,
                                       int64_t *elapsed_ms)
{
    return guarded([&] {
        std::vector<std::vector<double>> v;
        std::vector<std::set<int>> gt((size_t)nq);
        for (int64_t i = 0; i < nq; i++) {
            v.emplace_back(vectors + i * dim, vectors + (i + 1) * dim);
            gt[(size_t)i].insert(gt_ids + gt_off[i], gt_ids + gt_off[i + 1]);
        }
        auto r = mclab::deploy::SparsevectorRDFInit::topKAndPrecisionScore(v, gt, Config::parseString(conf_text), steps);
        const auto &top = std::get<0>(r);
        *rows = (int64_t)top.size();
        for (size_t i = 0; i < top.size(); i++)
            for (int j = 0; j < topK; j++) topk_ids[i * (size_t)topK + j] = j < (int)top[i].size() ? top[i][(size_t)j] : -1;
        *score = std::get<1>(r);
        *elapsed_ms = (int64_t)std::get<2>(r);
    });
}

int mclab_Sparse_clearAndClose(void) { return guarded([] { mclab::deploy::SparsevectorRDFInit::clearAndClose(); }); }
int64_t mclab_Sparse_size(void) { return mclab::deploy::SparsevectorRDFInit::size(); }
void *mclab_Sparse_handle(void) { return mclab::deploy::SparsevectorRDFInit::handle(); }

int mclab_Vectors_fromPythonString(const char *line, int32_t *id, int32_t *size, int32_t *indices, double *values, int cap, int *nnz)
{
    return guarded([&] {
        auto t = lsh::vector::Vectors::fromPythonString(line);
        *id = t.id; *size = t.size; *nnz = (int)t.values.size();
        for (int i = 0; i < (int)t.indices.size() && i < cap; i++) indices[i] = t.indices[(size_t)i];
        for (int i = 0; i < *nnz && i < cap; i++) values[i] = t.values[(size_t)i];
    });
}

int mclab_Vectors_parseDense(const char *line, int32_t *id, double *values, int cap, int *n)
{
    return guarded([&] {
        auto r = lsh::vector::Vectors::parseDense(line);
        *id = r.first; *n = (int)r.second.size();
        for (int i = 0; i < *n && i < cap; i++) values[i] = r.second[(size_t)i];
    });
}

int mclab_Vectors_fromString(const char *line, int32_t *id, int32_t *size, int32_t *indices, double *values, int cap, int *nnz)
{
    return guarded([&] {
        auto t = lsh::vector::Vectors::fromString(line);
        *id = t.id; *size = t.size; *nnz = (int)t.values.size();
        for (int i = 0; i < (int)t.indices.size() && i < cap; i++) indices[i] = t.indices[(size_t)i];
        for (int i = 0; i < *nnz && i < cap; i++) values[i] = t.values[(size_t)i];
    });
}

int mclab_Vectors_analysisKNN(const char *line, int k, int32_t *out)
{
    return guarded([&] {
        auto r = lsh::vector::Vectors::analysisKNN(line, k);
        std::copy(r.begin(), r.end(), out);
    });
}

}  // extern "C"
