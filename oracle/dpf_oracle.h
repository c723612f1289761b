/*
 * dpf_oracle.h -- CPU restatement of the reference's batch-query path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared
 * against (tests/, the smoke check and the cpu_baseline leg of the benchmark).
 * Nothing in the shipped library links, loads or calls it.
 *
 * The reference is Scala 2.10 / Java 7 and cannot be built in this image
 * (no JVM, no sbt, no dependency cache), so this file restates its algorithm
 * in plain C.  Every function cites the reference file:line it follows
 * (paths relative to the reference checkout):
 *
 *   Angle   src/main/scala/mclab/lsh/hashFamilies/AngleHashFamily.scala
 *   PStable src/main/scala/mclab/lsh/hashFamilies/PStableHashFamily.scala
 *   BAW     src/main/scala/mclab/storage/ByteArrayWrapper.scala
 *   Sim     src/main/scala/mclab/lsh/vector/SimilarityCalculator.scala
 *   LSH     src/main/scala/mclab/lsh/LSH.scala
 *   Part    src/main/scala/mclab/utils/Partitioner.scala
 *   Dense   src/main/scala/mclab/deploy/DensevectorRDFInit.scala
 *   RDT     src/main/java/mclab/mapdb/RandomDrawTreeMap.java
 *   Samp    src/main/scala/mclab/lsh/hashFamilies/Sampling.scala
 *   SigB    src/main/scala/mclab/lsh/hashFamilies/significantBits.scala
 *
 * Pinning: the known answers the reference's own unit tests hold for this
 * path (AngleHashSuite, PStableHashSuite, SimilarityCalculatorSuite,
 * RandomDrawTreeMapTest slot/offset + split tests) are checked in
 * tests/test_oracle_kat.py.  End-to-end candidate sets, probe generation,
 * partitioner output and rerank order are pinned by no reference test:
 * for those this restatement is the definition ("parity unpinned" beyond
 * the known answers above).
 */
#ifndef DPF_ORACLE_H
#define DPF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_FAMILY_ANGLE = 0, ORC_FAMILY_PSTABLE = 1 };
enum { ORC_INDEX_ORIGINAL = 0, ORC_INDEX_SAMPLING = 1,
       ORC_INDEX_CONTINUE_BITS = 2, ORC_INDEX_ANGLE_NEW = 3 };
enum { ORC_SELF_COMPAT = 0, ORC_SELF_ALWAYS = 1, ORC_SELF_NEVER = 2 };
enum { ORC_METRIC_DOT = 0, ORC_METRIC_L2 = 1 };

typedef struct orc_config {
    int32_t family;           /* mclab.lsh.name: angle | pStable            */
    int32_t dim;              /* mclab.lsh.vectorDim                        */
    int32_t num_functions;    /* size of the function pool A[F][dim]        */
    int32_t num_tables;       /* L = tableNum * permutationNum (Dense:107)  */
    int32_t chain_length;     /* mclab.lshTable.chainLength                 */
    int32_t partition_bits;   /* mclab.lsh.partitionBits                    */
    int32_t bucket_bits;      /* mclab.lshTable.bucketBits                  */
    int32_t dir_node_size;    /* mclab.lshTable.dirNodeSize                 */
    int32_t bucket_overflow;  /* mclab.lshTable.bufferOverflow              */
    int32_t type_of_index;    /* mclab.lsh.typeOfIndex                      */
    int32_t exclude_self;     /* ORC_SELF_*                                 */
} orc_config;

typedef struct orc_index orc_index;

/* ---- scalar pieces (exported so the known-answer tests can call them) ---- */

double  orc_dot_dense(const double *a, const double *v, int32_t n);
double  orc_dot_sparse_sparse(int32_t size,
                              const int32_t *idx1, const double *val1, int32_t n1,
                              const int32_t *idx2, const double *val2, int32_t n2);
int32_t orc_angle_sign(double x);
int32_t orc_angle_hash_dense(const double *fns, int32_t chain, int32_t dim, const double *v);
int32_t orc_angle_hash_sparse(int32_t size, int32_t chain,
                              const int32_t *const *fidx, const double *const *fval,
                              const int32_t *fn_nnz,
                              const int32_t *vidx, const double *vval, int32_t vnnz);
int32_t orc_d2i(double x);
int32_t orc_bytes_hashcode(const int8_t *bytes, int32_t n);
int32_t orc_pstable_hash_from_sums(const double *sums, const double *b, const int32_t *w, int32_t chain);
int32_t orc_pstable_hash_dense(const double *fns, const double *b, const int32_t *w,
                               int32_t chain, int32_t dim, const double *v);
int32_t orc_continue_bits_count(int32_t key, const int32_t num_of_bits[4]);
int32_t orc_sampling_one_key(int32_t key, const int32_t sampling_index[32]);
void    orc_scala_shuffle_0_31(int64_t seed, int32_t out[32]);
double  orc_angle_distance(int32_t key);
int32_t orc_angle_new_method(int32_t key);
int32_t orc_dir_offset_from_slot(const int32_t *dir, int32_t slot,
                                 int32_t dir_node_size);
void    orc_step_sub_indexes(int32_t current, int32_t steps, int32_t num_partitions,
                             int32_t *out, int32_t *out_n);
int32_t orc_probe_count(int32_t h);

/* ---- the index ---------------------------------------------------------- */

/* fns[F][dim]; b[F], w[F] (pStable only, may be NULL for angle);
 * chain_fn[L][m] indexes into the pool;
 * part_fns[L][pb][32]; part_b[L][pb], part_w[L][pb] (pStable only);
 * sampling_index[32] (typeOfIndex = sampling only, else NULL). */
orc_index *orc_create(const orc_config *cfg,
                      const double *fns, const double *b, const int32_t *w,
                      const int32_t *chain_fn,
                      const double *part_fns, const double *part_b, const int32_t *part_w,
                      const int32_t *sampling_index);
void orc_destroy(orc_index *ix);

/* hash of one vector for one table (LSH.calculateIndex, LSH:135-166). */
int32_t orc_hash(const orc_index *ix, int32_t table, const double *v);
/* LocalitySensitivePartitioner.getPartition of a hash for one table (Part:40-59). */
int32_t orc_partition(const orc_index *ix, int32_t table, int32_t h);

/* newFastFit order (Dense:127-151): ids first_id .. first_id+n-1, ascending,
 * every table.  X is [n][dim] row-major doubles and is copied. */
int orc_fit(orc_index *ix, const double *X, int64_t n, int32_t threads);

int64_t orc_size(const orc_index *ix);
int64_t orc_misflag_events(const orc_index *ix);
int64_t orc_short_hash_events(const orc_index *ix);

/* getSimilarWithStepWiseFaster (RDT:742-797) for one table: the raw list,
 * duplicates kept, in reference order.  Returns the count, writes at most cap. */
int64_t orc_get_similar(orc_index *ix, int32_t table, int32_t key, const double *v,
                        int32_t steps, int32_t *out, int64_t cap);

/* NewMultiThreadQueryBatch (Dense:335-431).  Result: per query the Set as a
 * sorted unique id list; offsets[nq+1]; *ids is malloc'ed, free with orc_free.
 * mode 0 = reference threading (thread i owns a table range), 1 = split queries. */
int orc_query_batch(orc_index *ix, const int32_t *qids, const double *Q, int64_t nq,
                    int32_t steps, int32_t threads, int32_t mode,
                    int64_t *offsets, int32_t **ids);
void orc_free(void *p);

/* topKAndPrecisionScore's rerank (Dense:477-505) over given candidate lists.
 * out_ids[nq][k] (-1 padded), out_scores[nq][k] (NaN padded). */
int orc_topk(const orc_index *ix, const double *Q, int64_t nq,
             const int64_t *offsets, const int32_t *cand, int32_t k, int32_t metric,
             int32_t threads, int32_t *out_ids, double *out_scores);

/* structure dump used to compare forests: number of d-nodes / leaves in a table */
void orc_table_stats(const orc_index *ix, int32_t table, int64_t *dnodes, int64_t *leaves,
                     int64_t *max_leaf);
/* all hashes of the fitted vectors for one table, hashes[n] */
void orc_fitted_hashes(const orc_index *ix, int32_t table, int32_t *hashes, int32_t *parts);

#ifdef __cplusplus
}
#endif
/* RDT:630-675 (the keyed walk of SparsevectorRDFInit.scala:407) instead of RDT:742-797 for every later query */
void orc_set_probe_own(orc_index *ix, int32_t on);
/* restricts every later query to the sub-indexes p with p % groups == group (a shard of this build's multi-GPU
 * placement; the reference has no such call) */
void orc_set_subindex_shard(orc_index *ix, int32_t group, int32_t groups);

#endif
