// Sort-based construction of the forest snapshot.
//
// Shape rule.  RandomDrawTreeMap.putInner (RandomDrawTreeMap.java:1662-1790)
// inserts ids one at a time: it descends directory nodes by NBITS-wide slots of
// the hash, and on reaching a bucket (linked chain) either head-inserts, or --
// when the chain already holds >= BUCKET_OVERFLOW entries and the level is >= 1
// (RandomDrawTreeMap.java:1719) -- replaces the chain by ONE new directory one
// level down and redistributes chain + newcomer into it.  The split does not
// cascade, so a child bucket can be born holding more than BUCKET_OVERFLOW ids
// and only splits when a further id arrives.  With ids inserted in ascending
// order (DensevectorRDFInit.newFastFit, DensevectorRDFInit.scala:127-151) this
// makes the final shape a function of the (hash, id) multiset:
//
//   a bucket at level v with n ids, c0 of which were present when it was
//   created (c0 = 0 under a root directory), stays a bucket iff
//   v == 0, or n <= B, or n == c0; otherwise it split when its
//   (max(c0, B) + 1)-th id (by ascending id) arrived, and the ids present at
//   that moment define the c0 of each child.
//
// The build applies that rule top-down over ids sorted by (sub-index, segment,
// trie bits, id); no pointer structure is ever materialised.
#include "dpf_forest.hpp"

#include <algorithm>
#include <atomic>
#include <thread>

namespace dpf {

static int ilog2(int32_t x)
{
    int r = 0;
    while ((1 << (r + 1)) <= x) r++;
    return r;
}

bool ForestGeometry::derive()
{
    if (dir_node_size < 32 || (dir_node_size & (dir_node_size - 1)) != 0 || dir_node_size > 256) return false;
    if (bucket_bits < 16 || bucket_bits > 31) return false;
    if (partition_bits < 0 || partition_bits > 8) return false;
    if (bucket_overflow < 1 || chain_length < 1 || chain_length > 32) return false;
    seg = 1 << (32 - bucket_bits);
    nbits = ilog2(dir_node_size);
    max_level = (chain_length - (32 - bucket_bits)) / nbits - 1;
    num_parts = 1 << partition_bits;
    if (max_level < 0 || (max_level + 1) * nbits > 32) return false;
    return true;
}

namespace {

struct Entry {
    uint64_t key;   // ((part * SEG + seg) << lowbits) | trie bits
    int32_t id;
};

struct TableBuilder {
    const ForestGeometry &g;
    std::vector<Entry> e;
    std::vector<uint32_t> dnodes;     // local node 0 reserved
    std::vector<uint32_t> leaf_off;
    std::vector<int32_t> ids;
    std::vector<uint32_t> roots;      // [P * SEG] local d-node index or 0
    std::vector<int32_t> scratch;
    int64_t misflag = 0, max_leaf = 0;
    int lowbits;

    explicit TableBuilder(const ForestGeometry &geo) : g(geo), lowbits((geo.max_level + 1) * geo.nbits) {}

    uint32_t new_dnode()
    {
        uint32_t idx = (uint32_t)(dnodes.size() / g.dir_node_size);
        dnodes.resize(dnodes.size() + g.dir_node_size, 0u);
        return idx;
    }
    int slot_of(const Entry &x, int level) const
    {
        return (int)((x.key >> (level * g.nbits)) & (uint64_t)(g.dir_node_size - 1));
    }
    uint32_t make_leaf(size_t lo, size_t hi)
    {
        uint32_t leaf = (uint32_t)(leaf_off.size() - 1);
        size_t at = ids.size();
        for (size_t i = lo; i < hi; i++) ids.push_back(e[i].id);
        std::sort(ids.begin() + at, ids.end());
        leaf_off.push_back((uint32_t)ids.size());
        max_leaf = std::max<int64_t>(max_leaf, (int64_t)(hi - lo));
        return (leaf << 1) | 1u;
    }
    // Fills the children of directory `node` (whose slots are the level-`level`
    // bits) from the sorted range [lo, hi).  ids <= thr were present when the
    // directory was created (thr < 0: none, i.e. a root directory).
    void fill_dnode(uint32_t node, size_t lo, size_t hi, int level, int32_t thr)
    {
        size_t i = lo;
        while (i < hi) {
            int s = slot_of(e[i], level);
            size_t j = i;
            int64_t c0 = 0;
            bool has_thr = false;
            while (j < hi && slot_of(e[j], level) == s) {
                c0 += (e[j].id <= thr);
                has_thr |= (e[j].id == thr);
                j++;
            }
            // RandomDrawTreeMap.java:1733-1734: the newcomer that triggered the
            // split is stored with a directory tag, repaired only when a
            // redistributed id shares its slot.  Count the unrepaired cases.
            if (has_thr && c0 == 1) misflag++;
            uint32_t word = build_child(i, j, level, c0);
            dnodes[(size_t)node * g.dir_node_size + s] = word;
            i = j;
        }
    }
    uint32_t build_child(size_t lo, size_t hi, int level, int64_t c0)
    {
        int64_t n = (int64_t)(hi - lo);
        int64_t B = g.bucket_overflow;
        if (level == 0 || n <= B || n == c0) return make_leaf(lo, hi);
        int64_t m = std::max(c0, B) + 1;
        scratch.resize((size_t)n);
        for (int64_t k = 0; k < n; k++) scratch[(size_t)k] = e[lo + (size_t)k].id;
        std::nth_element(scratch.begin(), scratch.begin() + (m - 1), scratch.end());
        int32_t thr = scratch[(size_t)(m - 1)];
        uint32_t node = new_dnode();
        fill_dnode(node, lo, hi, level - 1, thr);
        return node << 1;
    }
    void run(const int32_t *hashes, const int32_t *parts, int64_t n, int table)
    {
        const int L = g.num_tables;
        const uint64_t lowmask = lowbits >= 32 ? 0xffffffffull : ((1ull << lowbits) - 1);
        e.resize((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            uint32_t h = (uint32_t)hashes[i * L + table];
            uint64_t p = (uint64_t)(uint32_t)parts[i * L + table];
            uint64_t sg = h >> g.bucket_bits;
            e[(size_t)i].key = ((p * (uint64_t)g.seg + sg) << lowbits) | ((uint64_t)h & lowmask);
            e[(size_t)i].id = (int32_t)i;
        }
        std::sort(e.begin(), e.end(), [](const Entry &a, const Entry &b) {
            return a.key != b.key ? a.key < b.key : a.id < b.id;
        });
        dnodes.assign((size_t)g.dir_node_size, 0u);   // local node 0: reserved, empty
        leaf_off.assign(1, 0u);
        ids.clear();
        ids.reserve((size_t)n);
        roots.assign((size_t)g.num_parts * g.seg, 0u);
        size_t i = 0;
        while (i < (size_t)n) {
            uint64_t grp = e[i].key >> lowbits;
            size_t j = i;
            while (j < (size_t)n && (e[j].key >> lowbits) == grp) j++;
            uint32_t root = new_dnode();
            roots[(size_t)grp] = root;
            fill_dnode(root, i, j, g.max_level, -1);
            i = j;
        }
    }
};

}  // namespace

int build_forest(const ForestGeometry &g, const int32_t *hashes, const int32_t *parts,
                 int64_t n, int threads, ForestSnapshot &out)
{
    const int L = g.num_tables;
    if ((int64_t)n * L >= (int64_t)0x7fffffff) return -1;
    std::vector<TableBuilder *> tb((size_t)L, nullptr);
    std::atomic<int> next(0);
    auto worker = [&]() {
        for (;;) {
            int t = next.fetch_add(1);
            if (t >= L) break;
            tb[(size_t)t] = new TableBuilder(g);
            tb[(size_t)t]->run(hashes, parts, n, t);
        }
    };
    threads = std::max(1, std::min(threads, L));
    std::vector<std::thread> pool;
    for (int i = 0; i < threads; i++) pool.emplace_back(worker);
    for (auto &th : pool) th.join();

    const int DNS = g.dir_node_size;
    const size_t PS = (size_t)g.num_parts * g.seg;
    uint64_t total_dn = 1, total_leaves = 0, total_ids = 0;
    for (int t = 0; t < L; t++) {
        total_dn += tb[(size_t)t]->dnodes.size() / DNS - 1;
        total_leaves += tb[(size_t)t]->leaf_off.size() - 1;
        total_ids += tb[(size_t)t]->ids.size();
    }
    int rc = 0;
    if (total_dn * (uint64_t)DNS >= 0x7fffffffull || total_leaves >= 0x7fffffffull) rc = -1;
    if (rc == 0) {
        out.roots.assign(PS * (size_t)L, 0u);
        out.dnodes.assign((size_t)total_dn * DNS, 0u);
        out.leaf_off.assign((size_t)total_leaves + 1, 0u);
        out.ids.resize((size_t)total_ids);
        out.table_dnodes.assign((size_t)L, 0);
        out.table_leaves.assign((size_t)L, 0);
        out.part_counts.assign((size_t)L * g.num_parts, 0);
        out.max_leaf = 0;
        out.misflag_events = 0;
        uint32_t dn_base = 0, leaf_base = 0, id_base = 0;   // global d-node = dn_base + local (local >= 1)
        for (int t = 0; t < L; t++) {
            TableBuilder &b = *tb[(size_t)t];
            size_t ndn = b.dnodes.size() / DNS - 1, nlf = b.leaf_off.size() - 1;
            for (size_t k = 0; k < PS; k++)
                out.roots[(size_t)t * PS + k] = b.roots[k] ? b.roots[k] + dn_base : 0u;
            for (size_t node = 1; node <= ndn; node++)
                for (int s = 0; s < DNS; s++) {
                    uint32_t w = b.dnodes[node * DNS + s];
                    if (w) w = (w & 1u) ? ((((w >> 1) + leaf_base) << 1) | 1u) : (((w >> 1) + dn_base) << 1);
                    out.dnodes[((size_t)dn_base + node) * DNS + s] = w;
                }
            for (size_t k = 0; k < nlf; k++) out.leaf_off[(size_t)leaf_base + k + 1] = b.leaf_off[k + 1] + id_base;
            std::copy(b.ids.begin(), b.ids.end(), out.ids.begin() + id_base);
            out.table_dnodes[(size_t)t] = (int64_t)ndn;
            out.table_leaves[(size_t)t] = (int64_t)nlf;
            out.max_leaf = std::max(out.max_leaf, b.max_leaf);
            out.misflag_events += b.misflag;
            dn_base += (uint32_t)ndn;
            leaf_base += (uint32_t)nlf;
            id_base += (uint32_t)b.ids.size();
        }
        for (int64_t i = 0; i < n; i++)
            for (int t = 0; t < L; t++) out.part_counts[(size_t)t * g.num_parts + (uint32_t)parts[i * L + t]]++;
    }
    for (auto *p : tb) delete p;
    return rc;
}

}  // namespace dpf
