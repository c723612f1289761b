/*
 * dpf_oracle.c -- CPU restatement of the reference's batch-query path.
 * TEST INFRASTRUCTURE ONLY; see dpf_oracle.h for the citation key and for
 * what pins it.  Build with -ffp-contract=off: the reference never fuses a
 * multiply with an add (JVM doubles), so neither may this file.
 */
#include "dpf_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* dot products                                                              */
/* ------------------------------------------------------------------------ */

/* Sim:40-49 (sparse function x dense vector) and Sim:29-38 (dense x dense):
 * both zip the two VALUES arrays positionally and accumulate
 * similarity = similarity + x1*x2, left to right, starting from 0.0. */
double orc_dot_dense(const double *a, const double *v, int32_t n)
{
    double similarity = 0.0;
    for (int32_t i = 0; i < n; i++) {
        double prod = a[i] * v[i];
        similarity = similarity + prod;
    }
    return similarity;
}

/* Sim:9-27: AND of the two index bit sets, walked in ascending index order,
 * similarity += v1[index] * v2[index].  Indices are strictly increasing. */
double orc_dot_sparse_sparse(int32_t size,
                             const int32_t *idx1, const double *val1, int32_t n1,
                             const int32_t *idx2, const double *val2, int32_t n2)
{
    (void)size;
    double similarity = 0.0;
    int32_t i = 0, j = 0;
    while (i < n1 && j < n2) {
        if (idx1[i] < idx2[j]) i++;
        else if (idx1[i] > idx2[j]) j++;
        else {
            double prod = val1[i] * val2[j];
            similarity += prod;
            i++; j++;
        }
    }
    return similarity;
}

/* ------------------------------------------------------------------------ */
/* angle family                                                              */
/* ------------------------------------------------------------------------ */

/* Angle:184 -- sign(x) = 0 iff x <= 0 (NaN compares false, so NaN -> 1). */
int32_t orc_angle_sign(double x) { return (x <= 0) ? 0 : 1; }

/* Java's int << n uses n mod 32. */
static inline int32_t jshl(int32_t x, int32_t n)
{
    return (int32_t)((uint32_t)x << (n & 31));
}
static inline int32_t jushr(int32_t x, int32_t n)
{
    return (int32_t)((uint32_t)x >> (n & 31));
}

/* Angle:187-195 -- fns is chain x dim, function 0 ends up in the top bit. */
int32_t orc_angle_hash_dense(const double *fns, int32_t chain, int32_t dim, const double *v)
{
    int32_t result = 0;
    for (int32_t f = 0; f < chain; f++) {
        int32_t s = orc_angle_sign(orc_dot_dense(fns + (int64_t)f * dim, v, dim));
        result = jshl(result, 1) | s;
    }
    return jshl(result, 32 - chain);
}

/* Angle:204-219 -- sparse functions against a sparse vector. */
int32_t orc_angle_hash_sparse(int32_t size, int32_t chain,
                              const int32_t *const *fidx, const double *const *fval,
                              const int32_t *fn_nnz,
                              const int32_t *vidx, const double *vval, int32_t vnnz)
{
    int32_t result = 0;
    for (int32_t f = 0; f < chain; f++) {
        double s = orc_dot_sparse_sparse(size, fidx[f], fval[f], fn_nnz[f], vidx, vval, vnnz);
        result = jshl(result, 1) | orc_angle_sign(s);
    }
    return jshl(result, 32 - chain);
}

/* ------------------------------------------------------------------------ */
/* p-stable family                                                           */
/* ------------------------------------------------------------------------ */

/* Scala Double.toInt == JVM d2i: NaN -> 0, saturating, truncation toward 0. */
int32_t orc_d2i(double x)
{
    if (x != x) return 0;
    if (x >= 2147483647.0) return INT32_MAX;
    if (x <= -2147483648.0) return INT32_MIN;
    return (int32_t)x;
}

/* java.util.Arrays.hashCode(byte[]) as used by BAW:11-14. */
int32_t orc_bytes_hashcode(const int8_t *bytes, int32_t n)
{
    uint32_t r = 1;
    for (int32_t i = 0; i < n; i++) r = 31u * r + (uint32_t)(int32_t)bytes[i];
    return (int32_t)r;
}

/* PStable:130,140-142 -- ((sum + b) / w).toInt per function, each int as four
 * big-endian bytes (ByteBuffer.putInt), concatenated, Arrays.hashCode. */
int32_t orc_pstable_hash_from_sums(const double *sums, const double *b, const int32_t *w, int32_t chain)
{
    uint32_t r = 1;
    for (int32_t f = 0; f < chain; f++) {
        double t = sums[f] + b[f];
        t = t / (double)w[f];
        int32_t idx = orc_d2i(t);
        for (int s = 24; s >= 0; s -= 8) {
            int8_t byte = (int8_t)(((uint32_t)idx >> s) & 0xff);
            r = 31u * r + (uint32_t)(int32_t)byte;
        }
    }
    return (int32_t)r;
}

/* PStable:122-143 */
int32_t orc_pstable_hash_dense(const double *fns, const double *b, const int32_t *w,
                               int32_t chain, int32_t dim, const double *v)
{
    double *sums = (double *)malloc(sizeof(double) * (size_t)(chain > 0 ? chain : 1));
    for (int32_t f = 0; f < chain; f++) sums[f] = orc_dot_dense(fns + (int64_t)f * dim, v, dim);
    int32_t h = orc_pstable_hash_from_sums(sums, b, w, chain);
    free(sums);
    return h;
}

/* ------------------------------------------------------------------------ */
/* typeOfIndex post-transforms (LSH:152-162)                                 */
/* ------------------------------------------------------------------------ */

/* SigB:11-67 */
int32_t orc_continue_bits_count(int32_t key, const int32_t nb[4])
{
    uint32_t first_four = (uint32_t)key >> 28;
    int32_t a[4] = {0, 0, 0, 0};
    int32_t count = 0;
#define ORC_FLUSH()                                                        \
    do {                                                                   \
        if (count >= nb[0]) { a[0]++; a[1]++; a[2]++; a[3]++; }            \
        else if (count >= nb[1]) { a[1]++; a[2]++; a[3]++; }               \
        else if (count >= nb[2]) { a[2]++; a[3]++; }                       \
        else if (count >= nb[3]) { a[3]++; }                               \
        count = 0;                                                         \
    } while (0)
    for (int i = 0; i < 28; i++) {
        if ((((uint32_t)key >> i) & 1u) == 1u) {
            count += 1;
            if (i == 27) ORC_FLUSH();
        } else {
            ORC_FLUSH();
        }
    }
#undef ORC_FLUSH
    /* newIndexArray.reverse(i) << ((3 - i) * 7), i = 0..3 */
    uint32_t tmp = 0;
    for (int i = 0; i < 4; i++) tmp += (uint32_t)a[3 - i] << ((3 - i) * 7);
    tmp += first_four << 28;
    return (int32_t)tmp;
}

/* Samp:32-39 */
int32_t orc_sampling_one_key(int32_t key, const int32_t si[32])
{
    uint32_t tmp = 0;
    for (int j = 0; j < 32; j++)
        tmp += (((uint32_t)key >> (si[j] & 31)) & 1u) << (31 - j);
    return (int32_t)tmp;
}

/* Samp:6-11 -- new scala.util.Random(seed).shuffle(0 until 32).
 * scala.util.Random delegates to java.util.Random (48-bit LCG); Scala 2.10's
 * shuffle is "for n = len down to 2: swap(n-1, nextInt(n))".  Neither library
 * ships with the reference, so this is restated from their published
 * algorithms: parity unpinned. */
static int32_t jrand_next(uint64_t *seed, int bits)
{
    *seed = (*seed * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
    return (int32_t)((int64_t)*seed >> (48 - bits));
}
static int32_t jrand_next_int(uint64_t *seed, int32_t n)
{
    if ((n & -n) == n) return (int32_t)(((int64_t)n * (int64_t)jrand_next(seed, 31)) >> 31);
    int32_t bits, val;
    do {
        bits = jrand_next(seed, 31);
        val = bits % n;
    } while ((int32_t)((uint32_t)bits - (uint32_t)val + (uint32_t)(n - 1)) < 0);
    return val;
}
void orc_scala_shuffle_0_31(int64_t seed0, int32_t out[32])
{
    uint64_t seed = ((uint64_t)seed0 ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
    for (int i = 0; i < 32; i++) out[i] = i;
    for (int n = 32; n >= 2; n--) {
        int32_t k = jrand_next_int(&seed, n);
        int32_t t = out[n - 1]; out[n - 1] = out[k]; out[k] = t;
    }
}

/* SigB:100-112 -- angle (degrees) between the low 28 bits as a 0/1 vector and
 * the all-ones vector.  Integer dot products, then doubles. */
double orc_angle_distance(int32_t key)
{
    int32_t pc = __builtin_popcount((uint32_t)key & 0x0fffffffu);
    double num = (double)pc;                 /* v1 . base (Int)              */
    double lx = sqrt((double)28);            /* sqrt(base . base)            */
    double ly = sqrt((double)pc);            /* sqrt(v1 . v1)                */
    double angle = acos(num / (lx * ly));
    return angle * 360 / 2 / 3.141592653589793;
}

/* SigB:113-127 */
int32_t orc_angle_new_method(int32_t key)
{
    static const double metric[9] = {16.0, 25.0, 33.0, 39.0, 46.0, 52.0, 58.0, 66.0, 72.0};
    int32_t index = 0;
    while (index < 9 && (orc_angle_distance(key) > metric[index])) index += 1;
    const uint32_t mask = 0x7f;
    uint32_t k = (uint32_t)key;
    uint32_t first_label = (uint32_t)index;
    uint32_t first4 = (k >> 28) & mask;
    uint32_t first7 = (k >> 21) & mask;
    uint32_t three7 = (k >> 7) & mask;
    uint32_t last7 = k & mask;
    return (int32_t)(last7 + (three7 << 7) + (first_label << 14) + (first7 << 21) + (first4 << 28));
}

/* ------------------------------------------------------------------------ */
/* directory nodes                                                           */
/* ------------------------------------------------------------------------ */

static int32_t ilog2_exact(int32_t x)
{
    /* RDT:449 computes (int)(Math.log(n)/Math.log(2)); for the supported
     * sizes 32/64/128/256 the quotient is exactly the integer log2. */
    int32_t r = 0;
    while ((1 << (r + 1)) <= x) r++;
    return r;
}

/* RDT:1226-1267 -- dir = bitmap words then packed children; returns the
 * offset of the slot's child, negated when the slot's bit is clear. */
int32_t orc_dir_offset_from_slot(const int32_t *dir, int32_t slot, int32_t dir_node_size)
{
    int32_t nbits = ilog2_exact(dir_node_size);
    int32_t bitmap_size = dir_node_size / 32;
    int32_t range_bits = nbits - ilog2_exact(bitmap_size);
    int32_t bitmap_range = 0;
    if (bitmap_size > 1) bitmap_range = (int32_t)((uint32_t)slot >> range_bits);
    int32_t slot_in_range = slot & ((1 << range_bits) - 1);
    int32_t is_set = (int32_t)(((uint32_t)dir[bitmap_range] >> slot_in_range) & 1u);
    is_set <<= 1;
    int32_t offset = 0;
    for (int32_t i = 0; i < bitmap_range; i++) offset += __builtin_popcount((uint32_t)dir[i]);
    uint32_t mask_before = (uint32_t)((1u << slot_in_range) - 1u);
    offset += bitmap_size + __builtin_popcount((uint32_t)dir[bitmap_range] & mask_before);
    return -offset + is_set * offset;
}

/* RDT:613-621 */
void orc_step_sub_indexes(int32_t current, int32_t steps, int32_t num_partitions,
                          int32_t *out, int32_t *out_n)
{
    int32_t n = 0;
    for (int32_t s = 0; s < num_partitions; s++)
        if (__builtin_popcount((uint32_t)(s ^ current)) <= steps) out[n++] = s;
    *out_n = n;
}

/* RDT:753 -- new int[32 - Integer.numberOfLeadingZeros(h) - 4]; negative for
 * 0 <= h < 8 (the reference throws NegativeArraySizeException there). */
int32_t orc_probe_count(int32_t h)
{
    int32_t nlz = (h == 0) ? 32 : __builtin_clz((uint32_t)h);
    return 32 - nlz - 4;
}

/* ------------------------------------------------------------------------ */
/* the index                                                                 */
/* ------------------------------------------------------------------------ */

typedef struct { int32_t *data; int32_t len; } orc_dir;
typedef struct { uint32_t next; int32_t key; } orc_ln;      /* RDT:171-183 */

/* one record store ("engine") per (table, sub-index), RDT:1415-1459 */
typedef struct {
    int initialized;
    orc_dir *dirs; uint32_t ndirs, capdirs;     /* recid 1.. */
    orc_ln *lns;   uint32_t nlns, caplns;       /* recid 1.. */
    uint32_t *root;                             /* [SEG] */
} orc_engine;

struct orc_index {
    orc_config cfg;
    int32_t probe_own;            /* 1: RDT:630-675 instead of RDT:742-797 */
    int32_t sub_group, sub_groups; /* sub-index shard filter (sub_groups <= 1: none) */
    int32_t seg, nbits, mask, max_level, bitmap_size, num_partitions;
    double *fns; double *b; int32_t *w; int32_t *chain_fn;
    double *part_fns; double *part_b; int32_t *part_w;
    int32_t sampling_index[32];
    orc_engine *engines;                        /* [L][P] */
    double *vectors; int64_t n, cap;            /* vectorIdToVector */
    int32_t *hashes;                            /* [L][cap] filled on insert */
    int64_t *misflag;                           /* [L] */
    int64_t short_hash;                         /* probe-count < 0 events    */
    pthread_mutex_t mu;
};

static int32_t apply_type_of_index(const orc_index *ix, int32_t h)
{
    static const int32_t nb[4] = {6, 4, 2, 1};             /* LSH:158 */
    switch (ix->cfg.type_of_index) {
    case ORC_INDEX_SAMPLING:      return orc_sampling_one_key(h, ix->sampling_index);
    case ORC_INDEX_CONTINUE_BITS: return orc_continue_bits_count(h, nb);
    case ORC_INDEX_ANGLE_NEW:     return orc_angle_new_method(h);
    default:                      return h;
    }
}

orc_index *orc_create(const orc_config *cfg,
                      const double *fns, const double *b, const int32_t *w,
                      const int32_t *chain_fn,
                      const double *part_fns, const double *part_b, const int32_t *part_w,
                      const int32_t *sampling_index)
{
    orc_index *ix = (orc_index *)calloc(1, sizeof(*ix));
    ix->cfg = *cfg;
    /* RDT:435-459 */
    ix->seg = 1 << (32 - cfg->bucket_bits);
    ix->nbits = ilog2_exact(cfg->dir_node_size);
    ix->mask = (1 << ix->nbits) - 1;
    ix->max_level = (cfg->chain_length - (32 - cfg->bucket_bits)) / ix->nbits - 1;
    ix->bitmap_size = cfg->dir_node_size / 32;
    ix->num_partitions = 1 << cfg->partition_bits;          /* Part:28 */
    size_t F = (size_t)cfg->num_functions, d = (size_t)cfg->dim;
    size_t L = (size_t)cfg->num_tables, m = (size_t)cfg->chain_length, pb = (size_t)cfg->partition_bits;
    ix->fns = (double *)malloc(F * d * sizeof(double));
    memcpy(ix->fns, fns, F * d * sizeof(double));
    ix->chain_fn = (int32_t *)malloc(L * m * sizeof(int32_t));
    memcpy(ix->chain_fn, chain_fn, L * m * sizeof(int32_t));
    if (cfg->family == ORC_FAMILY_PSTABLE) {
        ix->b = (double *)malloc(F * sizeof(double)); memcpy(ix->b, b, F * sizeof(double));
        ix->w = (int32_t *)malloc(F * sizeof(int32_t)); memcpy(ix->w, w, F * sizeof(int32_t));
    }
    size_t npf = L * (pb ? pb : 1) * 32;
    ix->part_fns = (double *)calloc(npf, sizeof(double));
    if (pb) memcpy(ix->part_fns, part_fns, L * pb * 32 * sizeof(double));
    if (cfg->family == ORC_FAMILY_PSTABLE && pb) {
        ix->part_b = (double *)malloc(L * pb * sizeof(double)); memcpy(ix->part_b, part_b, L * pb * sizeof(double));
        ix->part_w = (int32_t *)malloc(L * pb * sizeof(int32_t)); memcpy(ix->part_w, part_w, L * pb * sizeof(int32_t));
    }
    if (sampling_index) memcpy(ix->sampling_index, sampling_index, sizeof(ix->sampling_index));
    ix->engines = (orc_engine *)calloc(L * (size_t)ix->num_partitions, sizeof(orc_engine));
    ix->misflag = (int64_t *)calloc(L, sizeof(int64_t));
    pthread_mutex_init(&ix->mu, NULL);
    return ix;
}

void orc_destroy(orc_index *ix)
{
    if (!ix) return;
    size_t ne = (size_t)ix->cfg.num_tables * (size_t)ix->num_partitions;
    for (size_t e = 0; e < ne; e++) {
        orc_engine *en = &ix->engines[e];
        for (uint32_t i = 1; i <= en->ndirs; i++) free(en->dirs[i].data);
        free(en->dirs); free(en->lns); free(en->root);
    }
    free(ix->engines); free(ix->fns); free(ix->b); free(ix->w); free(ix->chain_fn);
    free(ix->part_fns); free(ix->part_b); free(ix->part_w);
    free(ix->vectors); free(ix->hashes); free(ix->misflag);
    pthread_mutex_destroy(&ix->mu);
    free(ix);
}

/* The dot products of one chain, sums[j] = orc_dot_dense(function chain[j], v): four functions advance
 * together so that their four dependent add chains overlap in the pipeline.  Every sum is still its own
 * strict left-to-right accumulation from 0.0 (product, then add), so each equals orc_dot_dense bit for
 * bit (checked in tests/test_oracle_kat.py); only the wall-clock time of a fit changes. */
static void chain_sums(const orc_index *ix, const int32_t *chain, int32_t m, const double *v, double *sums)
{
    const int32_t d = ix->cfg.dim;
    int32_t j = 0;
    for (; j + 4 <= m; j += 4) {
        const double *a0 = ix->fns + (int64_t)chain[j] * d, *a1 = ix->fns + (int64_t)chain[j + 1] * d;
        const double *a2 = ix->fns + (int64_t)chain[j + 2] * d, *a3 = ix->fns + (int64_t)chain[j + 3] * d;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int32_t i = 0; i < d; i++) {
            const double x = v[i];
            double p0 = a0[i] * x, p1 = a1[i] * x, p2 = a2[i] * x, p3 = a3[i] * x;
            s0 = s0 + p0; s1 = s1 + p1; s2 = s2 + p2; s3 = s3 + p3;
        }
        sums[j] = s0; sums[j + 1] = s1; sums[j + 2] = s2; sums[j + 3] = s3;
    }
    for (; j < m; j++) sums[j] = orc_dot_dense(ix->fns + (int64_t)chain[j] * d, v, d);
}

/* Hasher:44-54 -> LSH:135-166 -> chain.compute(DenseVector) */
int32_t orc_hash(const orc_index *ix, int32_t table, const double *v)
{
    const orc_config *c = &ix->cfg;
    const int32_t *chain = ix->chain_fn + (int64_t)table * c->chain_length;
    int32_t h;
    double sums_fixed[64];
    double *sums = c->chain_length <= 64 ? sums_fixed : (double *)malloc(sizeof(double) * (size_t)c->chain_length);
    chain_sums(ix, chain, c->chain_length, v, sums);
    if (c->family == ORC_FAMILY_ANGLE) {
        int32_t result = 0;
        for (int32_t j = 0; j < c->chain_length; j++) {
            double s = sums[j];
            result = jshl(result, 1) | orc_angle_sign(s);
        }
        h = jshl(result, 32 - c->chain_length);
    } else {
        uint32_t r = 1;
        for (int32_t j = 0; j < c->chain_length; j++) {
            int32_t f = chain[j];
            double s = sums[j];
            double t = s + ix->b[f];
            t = t / (double)ix->w[f];
            int32_t idx = orc_d2i(t);
            for (int sh = 24; sh >= 0; sh -= 8) {
                int8_t byte = (int8_t)(((uint32_t)idx >> sh) & 0xff);
                r = 31u * r + (uint32_t)(int32_t)byte;
            }
        }
        h = (int32_t)r;
    }
    if (sums != sums_fixed) free(sums);
    return apply_type_of_index(ix, h);
}

/* Part:40-59 -- bits of the hash become a 32-dim 0/1 sparse vector (values
 * 1.0 at the set bits), hashed by the table's own partitioner LSH
 * (chainLength = partitionBits, vectorDim = 32, chain 0), including that
 * LSH's typeOfIndex transform (LSH:93-124), then >>> (32 - partitionBits).
 * The sparse x sparse dot (Sim:9-27) walks common indices in ascending
 * order; a zero coefficient contributes nothing either way. */
int32_t orc_partition(const orc_index *ix, int32_t table, int32_t h)
{
    const orc_config *c = &ix->cfg;
    int32_t pb = c->partition_bits;
    if (pb == 0) return 0;
    const double *fn = ix->part_fns + (int64_t)table * pb * 32;
    double sums[32];
    for (int32_t j = 0; j < pb; j++) {
        double similarity = 0.0;
        for (int i = 0; i < 32; i++) {
            if ((((uint32_t)h >> i) & 1u) && fn[j * 32 + i] != 0) {
                double prod = fn[j * 32 + i] * 1.0;
                similarity += prod;
            }
        }
        sums[j] = similarity;
    }
    int32_t ph;
    if (c->family == ORC_FAMILY_ANGLE) {
        int32_t result = 0;
        for (int32_t j = 0; j < pb; j++) result = jshl(result, 1) | orc_angle_sign(sums[j]);
        ph = jshl(result, 32 - pb);
    } else {
        ph = orc_pstable_hash_from_sums(sums, ix->part_b + (int64_t)table * pb,
                                        ix->part_w + (int64_t)table * pb, pb);
    }
    ph = apply_type_of_index(ix, ph);
    return jushr(ph, 32 - pb);
}

/* ---- record store helpers ---- */

static uint32_t engine_put_dir(orc_engine *en, int32_t len)
{
    if (en->ndirs + 2 > en->capdirs) {
        en->capdirs = en->capdirs ? en->capdirs * 2 : 64;
        en->dirs = (orc_dir *)realloc(en->dirs, sizeof(orc_dir) * en->capdirs);
    }
    uint32_t id = ++en->ndirs;
    en->dirs[id].data = (int32_t *)calloc((size_t)len, sizeof(int32_t));
    en->dirs[id].len = len;
    return id;
}
static uint32_t engine_put_ln(orc_engine *en, uint32_t next, int32_t key)
{
    if (en->nlns + 2 > en->caplns) {
        en->caplns = en->caplns ? en->caplns * 2 : 1024;
        en->lns = (orc_ln *)realloc(en->lns, sizeof(orc_ln) * en->caplns);
    }
    uint32_t id = ++en->nlns;
    en->lns[id].next = next;
    en->lns[id].key = key;
    return id;
}

/* RDT:1430-1459: SEG empty root directories per sub-index, created lazily
 * on the first put into that sub-index (RDT:1467-1489). */
static void init_partition_if_necessary(orc_index *ix, orc_engine *en)
{
    if (en->initialized) return;
    en->root = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)ix->seg);
    for (int32_t s = 0; s < ix->seg; s++) en->root[s] = engine_put_dir(en, ix->bitmap_size);
    en->initialized = 1;
}

/* RDT:1186-1201 */
static int32_t dir_get_slot(const orc_index *ix, const orc_dir *dir, int32_t slot)
{
    int32_t pos = orc_dir_offset_from_slot(dir->data, slot, ix->cfg.dir_node_size);
    if (pos < 0) return 0;
    return dir->data[pos];
}

/* RDT:1303-1329 (int[] branch) */
static void put_new_record_id_in_dir(const orc_index *ix, orc_dir *dir, int32_t slot, int32_t new_rec)
{
    int32_t offset = orc_dir_offset_from_slot(dir->data, slot, ix->cfg.dir_node_size);
    if (offset < 0) {
        offset = -offset;
        dir->data = (int32_t *)realloc(dir->data, sizeof(int32_t) * (size_t)(dir->len + 1));
        memmove(dir->data + offset + 1, dir->data + offset, sizeof(int32_t) * (size_t)(dir->len - offset));
        dir->len += 1;
        int32_t byte_pos = slot / 32, bit_pos = slot % 32;
        dir->data[byte_pos] = (int32_t)((uint32_t)dir->data[byte_pos] | (1u << bit_pos));
    }
    dir->data[offset] = new_rec;
}

static inline int32_t slot_at(const orc_index *ix, int32_t h, int32_t level)
{
    return (int32_t)(((uint32_t)h >> (level * ix->nbits)) & (uint32_t)ix->mask);
}

/* RDT:1662-1790.  Child words are (recid << 1) | tag with tag 1 = k-node.
 * d-nodes and linked nodes live in separate pools here, so the tag is always
 * meaningful.  One deliberate deviation, counted: at RDT:1733-1734 the
 * reference stores the NEW element's word with tag 0; the tag is only
 * repaired when a redistributed element shares its slot (RDT:1750-1753).
 * When none does, the reference leaves a linked node tagged as a directory
 * (undefined behaviour on the next visit).  This restatement stores the
 * intended k-node tag and counts the event in misflag[table]. */
static void put_inner(orc_index *ix, int32_t table, orc_engine *en, int32_t key, int32_t h)
{
    int32_t seg = (int32_t)((uint32_t)h >> ix->cfg.bucket_bits);
    uint32_t dir_recid = en->root[seg];
    int32_t level = ix->max_level;
    const int32_t *hashes = ix->hashes + (int64_t)table * ix->cap;
    for (;;) {
        orc_dir *dir = &en->dirs[dir_recid];
        int32_t slot = slot_at(ix, h, level);
        int32_t dir_offset = orc_dir_offset_from_slot(dir->data, slot, ix->cfg.dir_node_size);
        int32_t conflict = 0;
        int32_t recid = dir_offset < 0 ? 0 : dir->data[dir_offset];
        if (recid != 0) {
            if ((recid & 1) == 0) {
                dir_recid = (uint32_t)recid >> 1;
                level--;
                continue;
            }
            uint32_t r = (uint32_t)recid >> 1;
            while (r != 0) {
                const orc_ln *ln = &en->lns[r];
                if (ln->key == key) return;        /* value replace: no structural change */
                r = ln->next;
                conflict++;
            }
        }
        if (conflict >= ix->cfg.bucket_overflow && level >= 1) {
            uint32_t new_dir = engine_put_dir(en, ix->bitmap_size);
            dir = &en->dirs[dir_recid];            /* pool may have moved */
            uint32_t new_recid = engine_put_ln(en, 0, key);
            int32_t pos0 = slot_at(ix, h, level - 1);
            int32_t new_word = (int32_t)((new_recid << 1) | 1u);
            put_new_record_id_in_dir(ix, &en->dirs[new_dir], pos0, new_word);
            uint32_t node = dir_offset < 0 ? 0 : ((uint32_t)dir->data[dir_offset] >> 1);
            while (node != 0) {
                orc_ln *n = &en->lns[node];
                uint32_t next = n->next;
                int32_t pos = slot_at(ix, hashes[n->key], level - 1);   /* RDT:1747 hash(n.key) */
                int32_t recid2 = dir_get_slot(ix, &en->dirs[new_dir], pos);
                n->next = (uint32_t)recid2 >> 1;
                put_new_record_id_in_dir(ix, &en->dirs[new_dir], pos, (int32_t)((node << 1) | 1u));
                node = next;
            }
            if (dir_get_slot(ix, &en->dirs[new_dir], pos0) == new_word) ix->misflag[table]++;
            put_new_record_id_in_dir(ix, dir, slot, (int32_t)(new_dir << 1));
            return;
        } else {
            uint32_t head = dir_offset < 0 ? 0 : ((uint32_t)dir->data[dir_offset] >> 1);
            uint32_t new_recid = engine_put_ln(en, head, key);
            put_new_record_id_in_dir(ix, dir, slot, (int32_t)((new_recid << 1) | 1u));
            return;
        }
    }
}

/* RDT:1558-1584 for one table; the vector's hash was computed by the caller
 * (RDT:1498-1516 looks the vector up by id and hashes it). */
static void put_one(orc_index *ix, int32_t table, int32_t key, int32_t h)
{
    int32_t partition = orc_partition(ix, table, h);
    orc_engine *en = &ix->engines[(int64_t)table * ix->num_partitions + partition];
    init_partition_if_necessary(ix, en);
    put_inner(ix, table, en, key, h);
}

typedef struct { orc_index *ix; int64_t first, n; int32_t t0, t1; } fit_job;

static void *fit_worker(void *arg)
{
    fit_job *j = (fit_job *)arg;
    orc_index *ix = j->ix;
    for (int32_t t = j->t0; t < j->t1; t++) {
        int32_t *hashes = ix->hashes + (int64_t)t * ix->cap;
        for (int64_t i = j->first; i < j->first + j->n; i++)
            hashes[i] = orc_hash(ix, t, ix->vectors + i * ix->cfg.dim);
        for (int64_t i = j->first; i < j->first + j->n; i++)
            put_one(ix, t, (int32_t)i, hashes[i]);
    }
    return NULL;
}

/* Dense:127-151 (newFastFit): id = running counter, every table gets
 * put(id) in ascending id order.  Tables are independent structures, so the
 * worker threads split tables, never ids: per table the order is sequential. */
int orc_fit(orc_index *ix, const double *X, int64_t n, int32_t threads)
{
    int64_t first = ix->n;
    int64_t need = first + n;
    int32_t L = ix->cfg.num_tables, d = ix->cfg.dim;
    if (need > ix->cap) {
        int64_t ncap = need;
        double *nv = (double *)malloc(sizeof(double) * (size_t)ncap * (size_t)d);
        if (ix->vectors) memcpy(nv, ix->vectors, sizeof(double) * (size_t)first * (size_t)d);
        free(ix->vectors); ix->vectors = nv;
        int32_t *nh = (int32_t *)malloc(sizeof(int32_t) * (size_t)ncap * (size_t)L);
        for (int32_t t = 0; t < L; t++)
            if (ix->hashes) memcpy(nh + (int64_t)t * ncap, ix->hashes + (int64_t)t * ix->cap, sizeof(int32_t) * (size_t)first);
        free(ix->hashes); ix->hashes = nh;
        ix->cap = ncap;
    }
    memcpy(ix->vectors + first * d, X, sizeof(double) * (size_t)n * (size_t)d);
    ix->n = need;
    if (threads < 1) threads = 1;
    if (threads > L) threads = L;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    fit_job *jobs = (fit_job *)malloc(sizeof(fit_job) * (size_t)threads);
    for (int32_t i = 0; i < threads; i++) {
        jobs[i].ix = ix; jobs[i].first = first; jobs[i].n = n;
        jobs[i].t0 = (int32_t)((int64_t)i * L / threads);
        jobs[i].t1 = (int32_t)((int64_t)(i + 1) * L / threads);
        pthread_create(&th[i], NULL, fit_worker, &jobs[i]);
    }
    for (int32_t i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th); free(jobs);
    return 0;
}

int64_t orc_size(const orc_index *ix) { return ix->n; }
int64_t orc_misflag_events(const orc_index *ix)
{
    int64_t s = 0;
    for (int32_t t = 0; t < ix->cfg.num_tables; t++) s += ix->misflag[t];
    return s;
}
int64_t orc_short_hash_events(const orc_index *ix) { return ix->short_hash; }

void orc_fitted_hashes(const orc_index *ix, int32_t table, int32_t *hashes, int32_t *parts)
{
    const int32_t *src = ix->hashes + (int64_t)table * ix->cap;
    for (int64_t i = 0; i < ix->n; i++) {
        hashes[i] = src[i];
        if (parts) parts[i] = orc_partition(ix, table, src[i]);
    }
}

/* ---- search ---- */

typedef struct { int32_t *v; int64_t n, cap; } ivec;
static void ivec_push(ivec *a, int32_t x)
{
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 256;
        a->v = (int32_t *)realloc(a->v, sizeof(int32_t) * (size_t)a->cap);
    }
    a->v[a->n++] = x;
}

/* RDT:982 `ln.key != key` compares two boxed java.lang.Integer by reference:
 * equal ids are the same object only inside the Integer cache [-128, 127]. */
static inline int self_excluded(const orc_index *ix, int32_t cand, int32_t key)
{
    if (cand != key) return 0;
    switch (ix->cfg.exclude_self) {
    case ORC_SELF_ALWAYS: return 1;
    case ORC_SELF_NEVER:  return 0;
    default:              return key >= -128 && key <= 127;
    }
}

/* RDT:940-994.  Returns 0 for "null", 1 otherwise; appends to out. */
static int search_with_similarity(const orc_index *ix, const orc_engine *en, int32_t key,
                                  uint32_t rec_id, int32_t h, ivec *out)
{
    for (int32_t level = ix->max_level; level >= 0; level--) {
        const orc_dir *dir = &en->dirs[rec_id];
        int32_t slot = slot_at(ix, h, level);
        int32_t rec = dir_get_slot(ix, dir, slot);
        if (rec == 0) return 0;
        if ((rec & 1) != 0) {
            uint32_t working = (uint32_t)rec >> 1;
            for (;;) {
                const orc_ln *ln = &en->lns[working];
                if (!self_excluded(ix, ln->key, key)) ivec_push(out, ln->key);
                if (ln->next == 0) return 1;
                working = ln->next;
            }
        }
        rec_id = (uint32_t)rec >> 1;
    }
    return 1;
}

/* RDT:742-797.  A sub-index that never received a put has no locks and no
 * roots: the reference takes a NullPointerException there and moves on
 * (RDT:784-787); the persisted-store fallback (RDT:771-783) does the same
 * when nothing was ever persisted. */
static void get_similar(orc_index *ix, int32_t table, int32_t key, const double *v,
                        int32_t steps, ivec *out, int *short_hash)
{
    int32_t h = orc_hash(ix, table, v);
    int32_t seg = (int32_t)((uint32_t)h >> ix->cfg.bucket_bits);
    int32_t partition = orc_partition(ix, table, h);
    int32_t subs[256], nsubs = 0;
    orc_step_sub_indexes(partition, steps, ix->num_partitions, subs, &nsubs);
    if (ix->probe_own) {
        /* RDT:630-675 getSimilarWithStepWise(key, steps): the hash itself, in every sub-index within `steps`
         * (what SparsevectorRDFInit.scala:407 queries with); no probe list, so no probe-count quirk either */
        for (int32_t s = 0; s < nsubs; s++) {
            if (ix->sub_groups > 1 && subs[s] % ix->sub_groups != ix->sub_group) continue;
            const orc_engine *en = &ix->engines[(int64_t)table * ix->num_partitions + subs[s]];
            if (!en->initialized) continue;
            search_with_similarity(ix, en, key, en->root[seg], h, out);
        }
        return;
    }
    int32_t nprobe = orc_probe_count(h);
    if (nprobe < 0) { *short_hash = 1; return; }
    for (int32_t i = 0; i < nprobe; i++) {
        int32_t pro = h ^ (int32_t)(1u << i);
        for (int32_t s = 0; s < nsubs; s++) {
            /* not in the reference: the checker's view of a sub-index shard of this build's multi-GPU placement */
            if (ix->sub_groups > 1 && subs[s] % ix->sub_groups != ix->sub_group) continue;
            const orc_engine *en = &ix->engines[(int64_t)table * ix->num_partitions + subs[s]];
            if (!en->initialized) continue;
            search_with_similarity(ix, en, key, en->root[seg], pro, out);
        }
    }
}

void orc_set_probe_own(orc_index *ix, int32_t on) { ix->probe_own = on != 0; }
void orc_set_subindex_shard(orc_index *ix, int32_t group, int32_t groups) { ix->sub_group = group; ix->sub_groups = groups; }

int64_t orc_get_similar(orc_index *ix, int32_t table, int32_t key, const double *v,
                        int32_t steps, int32_t *out, int64_t cap)
{
    ivec a = {0};
    int sh = 0;
    get_similar(ix, table, key, v, steps, &a, &sh);
    if (sh) ix->short_hash++;
    for (int64_t i = 0; i < a.n && i < cap; i++) out[i] = a.v[i];
    int64_t n = a.n;
    free(a.v);
    return n;
}

static int cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}
static int64_t sort_unique(int32_t *v, int64_t n)
{
    if (n == 0) return 0;
    qsort(v, (size_t)n, sizeof(int32_t), cmp_i32);
    int64_t m = 1;
    for (int64_t i = 1; i < n; i++) if (v[i] != v[m - 1]) v[m++] = v[i];
    return m;
}

typedef struct {
    orc_index *ix; const int32_t *qids; const double *Q; int64_t nq;
    int32_t steps, t0, t1; int64_t q0, q1;
    ivec *results;            /* [nq] shared, guarded by ix->mu in mode 0 */
    int mode;
} query_job;

/* Dense:415-431 QueryTask.query: for each query, union over this thread's
 * tables, then merge into resultsArray(i) under one monitor. */
static void *query_worker(void *arg)
{
    query_job *j = (query_job *)arg;
    orc_index *ix = j->ix;
    int32_t d = ix->cfg.dim;
    int64_t sh_events = 0;
    for (int64_t i = j->q0; i < j->q1; i++) {
        ivec one = {0};
        for (int32_t t = j->t0; t < j->t1; t++) {
            int sh = 0;
            get_similar(ix, t, j->qids[i], j->Q + i * d, j->steps, &one, &sh);
            sh_events += sh;
        }
        one.n = sort_unique(one.v, one.n);
        if (j->mode == 0) pthread_mutex_lock(&ix->mu);
        ivec *r = &j->results[i];
        for (int64_t k = 0; k < one.n; k++) ivec_push(r, one.v[k]);
        if (j->mode == 0) pthread_mutex_unlock(&ix->mu);
        free(one.v);
    }
    pthread_mutex_lock(&ix->mu);
    ix->short_hash += sh_events;
    pthread_mutex_unlock(&ix->mu);
    return NULL;
}

/* Dense:335-360.  mode 0: thread i owns tables [i*L/T, (i+1)*L/T) and loops
 * over every query (Dense:344-350).  mode 1: queries split across threads. */
int orc_query_batch(orc_index *ix, const int32_t *qids, const double *Q, int64_t nq,
                    int32_t steps, int32_t threads, int32_t mode,
                    int64_t *offsets, int32_t **ids)
{
    if (threads < 1) threads = 1;
    int32_t L = ix->cfg.num_tables;
    ivec *results = (ivec *)calloc((size_t)(nq > 0 ? nq : 1), sizeof(ivec));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    query_job *jobs = (query_job *)malloc(sizeof(query_job) * (size_t)threads);
    for (int32_t i = 0; i < threads; i++) {
        query_job *j = &jobs[i];
        j->ix = ix; j->qids = qids; j->Q = Q; j->nq = nq; j->steps = steps;
        j->results = results; j->mode = mode;
        if (mode == 0) {
            j->t0 = (int32_t)((int64_t)i * L / threads);
            j->t1 = (int32_t)((int64_t)(i + 1) * L / threads);
            j->q0 = 0; j->q1 = nq;
        } else {
            j->t0 = 0; j->t1 = L;
            j->q0 = nq * i / threads; j->q1 = nq * (i + 1) / threads;
        }
        pthread_create(&th[i], NULL, query_worker, j);
    }
    for (int32_t i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th); free(jobs);
    int64_t total = 0;
    for (int64_t i = 0; i < nq; i++) {
        results[i].n = sort_unique(results[i].v, results[i].n);
        offsets[i] = total;
        total += results[i].n;
    }
    offsets[nq] = total;
    int32_t *all = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
    for (int64_t i = 0; i < nq; i++) {
        if (results[i].n) memcpy(all + offsets[i], results[i].v, sizeof(int32_t) * (size_t)results[i].n);
        free(results[i].v);
    }
    free(results);
    *ids = all;
    return 0;
}

void orc_free(void *p) { free(p); }

/* ---- rerank ---- */

typedef struct { double score; int32_t id; } scored;
static int metric_for_sort;
static int better(const scored *a, const scored *b, int metric)
{
    if (a->score != b->score)
        return metric == ORC_METRIC_DOT ? (a->score > b->score) : (a->score < b->score);
    return a->id < b->id;
}

typedef struct {
    const orc_index *ix; const double *Q; const int64_t *offsets; const int32_t *cand;
    int32_t k, metric; int64_t q0, q1; int32_t *out_ids; double *out_scores;
} topk_job;

/* Dense:477-505: scores = candidateMatrix * query (a dot product per
 * candidate row), argsort, reverse, first topK: descending inner product.
 * breeze's gemv summation order and the order of equal scores are not pinned
 * by anything in the reference; here: left-to-right fp64 sum, ties by
 * ascending id.  metric L2 (sum of squared differences, ascending) is the
 * build's own addition. */
static void *topk_worker(void *arg)
{
    topk_job *j = (topk_job *)arg;
    const orc_index *ix = j->ix;
    int32_t d = ix->cfg.dim, k = j->k;
    scored *best = (scored *)malloc(sizeof(scored) * (size_t)(k + 1));
    for (int64_t q = j->q0; q < j->q1; q++) {
        const double *qv = j->Q + q * d;
        int32_t nb = 0;
        for (int64_t c = j->offsets[q]; c < j->offsets[q + 1]; c++) {
            int32_t id = j->cand[c];
            const double *x = ix->vectors + (int64_t)id * d;
            double s = 0.0;
            if (j->metric == ORC_METRIC_DOT) {
                for (int32_t i = 0; i < d; i++) { double p = x[i] * qv[i]; s = s + p; }
            } else {
                for (int32_t i = 0; i < d; i++) { double df = qv[i] - x[i]; double p = df * df; s = s + p; }
            }
            scored cur = {s, id};
            if (nb == k && !better(&cur, &best[nb - 1], j->metric)) continue;
            int32_t pos = nb < k ? nb : k - 1;
            while (pos > 0 && better(&cur, &best[pos - 1], j->metric)) { best[pos] = best[pos - 1]; pos--; }
            best[pos] = cur;
            if (nb < k) nb++;
        }
        for (int32_t i = 0; i < k; i++) {
            j->out_ids[q * k + i] = i < nb ? best[i].id : -1;
            j->out_scores[q * k + i] = i < nb ? best[i].score : NAN;
        }
    }
    free(best);
    return NULL;
}

int orc_topk(const orc_index *ix, const double *Q, int64_t nq,
             const int64_t *offsets, const int32_t *cand, int32_t k, int32_t metric,
             int32_t threads, int32_t *out_ids, double *out_scores)
{
    (void)metric_for_sort;
    if (threads < 1) threads = 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    topk_job *jobs = (topk_job *)malloc(sizeof(topk_job) * (size_t)threads);
    for (int32_t i = 0; i < threads; i++) {
        topk_job *j = &jobs[i];
        j->ix = ix; j->Q = Q; j->offsets = offsets; j->cand = cand; j->k = k; j->metric = metric;
        j->q0 = nq * i / threads; j->q1 = nq * (i + 1) / threads;
        j->out_ids = out_ids; j->out_scores = out_scores;
        pthread_create(&th[i], NULL, topk_worker, j);
    }
    for (int32_t i = 0; i < threads; i++) pthread_join(th[i], NULL);
    free(th); free(jobs);
    return 0;
}

/* ---- structure statistics ---- */

static void walk_stats(const orc_index *ix, const orc_engine *en, uint32_t dir_id,
                       int64_t *dnodes, int64_t *leaves, int64_t *max_leaf)
{
    const orc_dir *dir = &en->dirs[dir_id];
    (*dnodes)++;
    for (int32_t i = ix->bitmap_size; i < dir->len; i++) {
        int32_t rec = dir->data[i];
        if (rec & 1) {
            int64_t n = 0;
            for (uint32_t r = (uint32_t)rec >> 1; r; r = en->lns[r].next) n++;
            (*leaves)++;
            if (n > *max_leaf) *max_leaf = n;
        } else {
            walk_stats(ix, en, (uint32_t)rec >> 1, dnodes, leaves, max_leaf);
        }
    }
}

void orc_table_stats(const orc_index *ix, int32_t table, int64_t *dnodes, int64_t *leaves, int64_t *max_leaf)
{
    *dnodes = 0; *leaves = 0; *max_leaf = 0;
    for (int32_t p = 0; p < ix->num_partitions; p++) {
        const orc_engine *en = &ix->engines[(int64_t)table * ix->num_partitions + p];
        if (!en->initialized) continue;
        for (int32_t s = 0; s < ix->seg; s++) walk_stats(ix, en, en->root[s], dnodes, leaves, max_leaf);
    }
}
