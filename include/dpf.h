/*
 * dpf.h -- C ABI of the MI355X Dynamic Partition Forest batch-query engine.
 *
 * This is the drop-in boundary for the reference's batch-query hot path
 * (LSH projection -> forest bucket probe -> candidate rerank / top-k).  Every
 * entry point names the reference interface it replaces; reference paths are
 * relative to the reference checkout:
 *
 *   Dense  src/main/scala/mclab/deploy/DensevectorRDFInit.scala
 *   LSH    src/main/scala/mclab/lsh/LSH.scala
 *   Hasher src/main/scala/mclab/lsh/Hasher.scala
 *   Part   src/main/scala/mclab/utils/Partitioner.scala
 *   RDT    src/main/java/mclab/mapdb/RandomDrawTreeMap.java
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative DPF_E* code, and dpf_last_error() returns the message
 * of the last failure on that handle (or of the last failed dpf_create when
 * called with NULL).  No exception crosses the boundary.  A handle serves one
 * call at a time (the reference keeps its batch result in one shared field,
 * Dense:324); distinct handles are independent.  "host" pointers are ordinary
 * process memory, "device" pointers are HIP device memory on the handle's
 * device.  All matrices are row-major and dense.
 */
#ifndef DPF_H
#define DPF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPF_OK          0
#define DPF_EINVAL     -1   /* bad argument / unsupported configuration          */
#define DPF_ENOTFIT    -2   /* query before fit: the reference prints "need to fit
                               the data first" and returns empty sets, Dense:422-424 */
#define DPF_ENOFAMILY  -3   /* hash family / partitioner not set                 */
#define DPF_EHIP       -4   /* a HIP runtime call failed                         */
#define DPF_ENOMEM     -5
#define DPF_ECAPACITY  -6   /* index larger than the 32-bit arena limits (RDT:1330-1363) */
#define DPF_EBUSY      -7   /* dpf_query_topk_device_static: another thread's call is using this GPU's scratch array */

enum { DPF_FAMILY_ANGLE = 0, DPF_FAMILY_PSTABLE = 1 };              /* mclab.lsh.name        */
enum { DPF_INDEX_ORIGINAL = 0, DPF_INDEX_SAMPLING = 1,
       DPF_INDEX_CONTINUE_BITS = 2, DPF_INDEX_ANGLE_NEW = 3 };      /* mclab.lsh.typeOfIndex */
enum { DPF_SELF_COMPAT = 0, DPF_SELF_ALWAYS = 1, DPF_SELF_NEVER = 2 };
enum { DPF_METRIC_DOT = 0, DPF_METRIC_L2 = 1 };
enum { DPF_F32 = 0, DPF_F64 = 1 };
/* DPF_FLAG_HASH_ONLY: the handle only serves dpf_hash_batch (LSH.calculateIndex); the forest
 * geometry keys are not validated and fit / query calls are refused. */
enum { DPF_FLAG_HASH_ONLY = 1,
/* DPF_FLAG_GATHER_ONLY: always rerank with the per-query gather kernel, never with the
 * bucket-major scan (same results; for measurement and A/B). */
       DPF_FLAG_GATHER_ONLY = 2,
/* DPF_FLAG_HOST_FOREST: derive the forest on the host (sort-based builder, 16 threads) instead of
 * on the device (same arrays; for cross-checking and A/B). */
       DPF_FLAG_HOST_FOREST = 4,
/* DPF_FLAG_FULL_FILTER: the bucket-major scan's fp16 rows keep every dimension.  Without it, l2 handles whose
 * dimension exceeds a row length the scan has a kernel for by at most 8 (and at most a sixteenth: 100 -> 96,
 * 136 -> 128, 260 -> 256, ...) scan rows cut there and put the remaining dimensions' share back for the scores that
 * pass: the same filter in two stages, same results, fewer bytes per scanned row (the flag is for A/B). */
       DPF_FLAG_FULL_FILTER = 8,
/* DPF_FLAG_NO_INT8: never scan in int8.  Without it an l2 handle of up to 256 dimensions whose resident rows are all
 * integer-valued inside one window of 256 values within +-2048 (SIFT's 0..255) keeps an int8 copy of them beside the fp16 one, and a
 * batch whose queries fit the window too is scanned with exact integer products on half the bytes (same results;
 * other batches take the fp16 scan; the flag is for A/B). */
       DPF_FLAG_NO_INT8 = 16,
/* DPF_FLAG_NO_CENTRE: the fp16 copies of an l2 handle are taken of the rows as they are.  Without it they are taken
 * around the mean of the resident rows (queries likewise): distances do not change, the rounding distances the filter
 * has to allow for shrink with the values - rows with a large common offset stay on the bucket-major path instead of
 * forwarding everything (same results; the flag is for A/B). */
       DPF_FLAG_NO_CENTRE = 32,
/* DPF_FLAG_BUCKET_ORDER_ROWS / DPF_FLAG_NO_BUCKET_ORDER_ROWS: the bucket-major scan can read its fp16 rows from a second
 * copy laid out in bucket order (one row per bucket entry: num_tables times the fp16 table; rows of up to 256 dimensions),
 * a scan tile's rows then being one contiguous run instead of a gather by id.  With neither flag the handle keeps that
 * copy when the rows are no whole number of 128-byte lines, the fp16 table exceeds 384 MiB (tables that stay in the
 * Infinity Cache are read faster by id) and the device has room; the first flag keeps it whenever there is room, the
 * second never.  Same results either way (dpf_stats.bucket_order_bytes tells what happened). */
       DPF_FLAG_BUCKET_ORDER_ROWS = 64,
       DPF_FLAG_NO_BUCKET_ORDER_ROWS = 128 };

/* The subset of the reference's Typesafe-Config keys the path consumes
 * (TestSettings.scala:8-45; read at LSH:19-33, Dense:39-58). */
typedef struct dpf_config {
    int32_t family;           /* mclab.lsh.name                                   */
    int32_t dim;              /* mclab.lsh.vectorDim                              */
    int32_t num_functions;    /* size F of the function pool (mclab.lsh.familySize
                                 when chains are pulled from the family)          */
    int32_t num_tables;       /* L = tableNum * permutationNum (Dense:107)        */
    int32_t chain_length;     /* mclab.lshTable.chainLength                       */
    int32_t partition_bits;   /* mclab.lsh.partitionBits                          */
    int32_t bucket_bits;      /* mclab.lshTable.bucketBits                        */
    int32_t dir_node_size;    /* mclab.lshTable.dirNodeSize                       */
    int32_t bucket_overflow;  /* mclab.lshTable.bufferOverflow                    */
    int32_t type_of_index;    /* mclab.lsh.typeOfIndex                            */
    int32_t exclude_self;     /* DPF_SELF_COMPAT reproduces RDT:982 exactly        */
    int32_t metric;           /* rerank order: DPF_METRIC_DOT = Dense:490          */
    int32_t top_k;            /* mclab.lsh.topK                                   */
    int32_t dtype;            /* DPF_F32 / DPF_F64: element type of base and query
                                 vectors at this boundary and in HBM              */
    int32_t device;           /* HIP device ordinal                               */
    int32_t flags;            /* DPF_FLAG_* */
    int32_t table_begin;      /* table shard: this handle probes only tables          */
    int32_t table_end;        /* table_begin <= t < table_end; table_end <= 0 = all L */
    int32_t subindex_group;   /* sub-index shard: of every table this handle probes only the sub-indexes p with
                                 p % subindex_groups == subindex_group (the reference's own partitioning unit,
                                 Partitioner.scala:27-64); subindex_groups <= 1 = all sub-indexes          */
    int32_t subindex_groups;
    int64_t row_begin;        /* rerank shard: this handle scores only candidates  */
    int64_t row_end;          /* with row_begin <= id < row_end; row_end < 0 = all */
} dpf_config;

typedef struct dpf_stats {
    int64_t num_vectors;         /* ids fitted so far                              */
    int64_t num_dnodes;          /* directory nodes in the snapshot                */
    int64_t num_leaves;          /* buckets (k-node chains) in the snapshot        */
    int64_t max_leaf;            /* longest bucket                                 */
    int64_t misflag_events;      /* splits that hit the RDT:1733-1734 tag quirk    */
    int64_t short_hash_events;   /* (query, table) pairs with 0 <= hash < 8, where
                                    RDT:753 throws; such pairs return nothing      */
    int64_t last_raw_candidates; /* last batch: ids visited before set union       */
    int64_t last_unique_candidates; /* last batch: sum of per-query set sizes      */
    double  last_ms_hash, last_ms_probe, last_ms_dedupe, last_ms_rerank;
    double  last_ms_scan;        /* bucket-major scan kernel of the last top-k batch, summed over its
                                    launches (0: gather path) */
    int64_t last_path;           /* rerank path of the last top-k batch: 0 gather, 1 bucket-major */
    int64_t last_pairs;          /* (query, bucket) incidences of the last bucket-major batch */
    int64_t last_forwarded;      /* candidates forwarded to the exact rerank by the scan */
    int64_t last_scan_launches;  /* launches of the scan kernel in the last bucket-major batch */
    int64_t snapshot_bytes;      /* HBM bytes of the forest snapshot               */
    int64_t base_bytes;          /* HBM bytes of the resident base rows            */
    double  fit_ms_hash;         /* last dpf_fit / dpf_load: copy-in + hashing of the new rows      */
    double  fit_ms_forest;       /* ... forest construction (put / split, RDT:1558-1790) + tiles     */
    double  fit_ms_total;        /* ... whole call                                                   */
    int64_t last_overflowed;     /* queries of the last bucket-major batch whose forward list overflowed and
                                    that were answered by the gather path instead */
    int64_t last_sizes_carried;  /* last bucket-major batch: 0 = the host read the batch's list sizes before
                                    filling the lists, 1 = it did not wait for them (buffers sized by earlier
                                    batches), 2 = as 1, but the batch outgrew the buffers and was run again */
    /* since dpf_create, over the top-k batches of this handle (no device synchronisation needed to read them) */
    int64_t total_topk_batches;      /* dpf_query_topk / dpf_query_topk_device calls with at least one query   */
    int64_t total_sized_on_host;     /* bucket-major batches of kind last_sizes_carried = 0                     */
    int64_t total_carried;           /* ... = 1                                                                 */
    int64_t total_rerun;             /* ... = 2                                                                 */
    int64_t total_overflowed;        /* queries answered by the gather path because their forward list overflowed */
    int64_t total_gather_batches;    /* batches answered by the gather path as a whole                          */
    /* the bucket-major scan's view of the fitted index (0 when the handle reranks by gather only) */
    int64_t base16_bytes;            /* HBM bytes of the fp16 copy of the resident rows                          */
    int64_t tile_rows;               /* rows per scan tile (64, or 16 for rows of more than 768 dimensions)      */
    int64_t num_tiles;               /* tiles the buckets are cut into                                           */
    int64_t resident_ids;            /* bucket entries (id, table) of this handle's row / table shard            */
    int64_t hash_undecided_rows;     /* since dpf_create: vectors whose matrix-core projection fell inside the guard band
                                        of some function and were hashed again by the exact kernel                       */
    int64_t base8_bytes;             /* HBM bytes of the int8 copy of the resident rows (0: the rows do not allow one)   */
    int64_t last_scan_int8;          /* 1: the last bucket-major batch was scanned in int8, 0: in fp16                    */
    int64_t bucket_order_bytes;      /* HBM bytes of the bucket-order copy of the fp16 rows (0: the scan gathers its rows by id) */
    int64_t last_longest_list;       /* last bucket-major batch: entries of the longest forward list (before duplicates are dropped) */
    int64_t debug_bound_violations;  /* since dpf_create: indices outside their arrays met by the scan's loaders; only a build of
                                        the library with -DDPF_DEBUG_BOUNDS looks (and clamps them); always 0 otherwise        */
} dpf_stats;

typedef struct dpf_index dpf_index;

/* new LSH(conf) + DensevectorRDFInit.initializeRDFHashMap(conf)  (LSH:17-53, Dense:50-118). */
int dpf_create(const dpf_config *cfg, dpf_index **out);
/* DensevectorRDFInit.clearAndClose()  (Dense:453-458). */
void dpf_destroy(dpf_index *ix);
const char *dpf_last_error(const dpf_index *ix);

/* The hash chains LSH.initHashChains builds (LSH:29-82): a pool fns[F][dim],
 * b[F] and w[F] for pStable (NULL for angle), and chain_fn[L][chain_length]
 * naming the pool function at each chain position (position 0 = top bit for
 * angle, AngleHashFamily.scala:187-195).  Copied.  Host pointers. */
int dpf_set_hash_family(dpf_index *ix, const double *fns, const double *b, const int32_t *w,
                        const int32_t *chain_fn);
/* One LocalitySensitivePartitioner per table (Dense:77, Part:27-64):
 * part_fns[L][partition_bits][32]; part_b / part_w [L][partition_bits] for pStable. */
int dpf_set_partitioner(dpf_index *ix, const double *part_fns, const double *part_b,
                        const int32_t *part_w);
/* The bit permutation of typeOfIndex = sampling (Sampling.scala:6-11), 32 entries. */
int dpf_set_sampling_index(dpf_index *ix, const int32_t *sampling_index);

/* DensevectorRDFInit.newFastFit / newMultiThreadFit (Dense:127-206) after the
 * text has been parsed: appends n rows X[n][dim] (host, element type cfg.dtype);
 * their ids continue the running counter (Dense:181,190).  Hashes are computed
 * on the device and stay there; the forest snapshot (what put / putInner, RDT:1558-1790,
 * would have grown by inserting the ids in ascending order) is derived on the device from
 * the hashes of all ids fitted so far. */
int dpf_fit(dpf_index *ix, const void *X, int64_t n);
/* (A base too large to hold on the host at once is streamed in: DPF_OPT_EXPECTED_ROWS = total, DPF_OPT_DEFER_BUILD = 1,
 * dpf_fit chunk after chunk - every handle of a sharded index sees every chunk, hashes it and keeps the rows of its shard -
 * then DPF_OPT_DEFER_BUILD = 0 and dpf_fit(ix, NULL, 0).) */

/* LocalitySensitiveHasher.hash for every table (Hasher:44-54 -> LSH:135-166)
 * and LocalitySensitivePartitioner.getPartition of each hash (Part:40-59):
 * hashes[nq][L], parts[nq][L] (parts may be NULL).  Host pointers. */
int dpf_hash_batch(dpf_index *ix, const void *Q, int64_t nq, int32_t *hashes, int32_t *parts);

/* DensevectorRDFInit.NewMultiThreadQueryBatch(queryArray, denseVectorArray, steps, _)
 * (Dense:335-360): per query the Set of similar ids as a sorted unique list,
 * restricted to [row_begin, row_end): query i's ids are cand_ids[cand_offsets[i] .. cand_offsets[i + 1]).
 * cand_offsets[nq+1] is caller memory.  Host pointers.  Three ways to take the lists:
 *   dpf_query_batch           *cand_ids is a malloc'd block the caller owns until dpf_free (the lists pass through a
 *                             pinned staging area in pieces; a batch's Sets can be gigabytes - 2.9 GB for 10k queries
 *                             over 30 trees of 1M rows - and fresh pageable memory takes them at a fraction of the link's rate);
 *   dpf_query_batch_view      *cand_ids points into a pinned host buffer the HANDLE owns, grows and reuses: valid until the
 *                             handle's next call that returns Sets, or dpf_destroy; for callers that copy the lists
 *                             elsewhere at once (the JNI shim into a Java array); never passed to dpf_free;
 *   dpf_query_batch_resident  the lists stay on the device, only cand_offsets (the Set sizes) comes back; dpf_sets_fetch
 *                             then copies the lists of the queries [q0, q1) to out[cand_offsets[q1] - cand_offsets[q0]]
 *                             and dpf_sets_topk reranks every query's resident Set (Dense:477-505) - what a caller that
 *                             looks at few Sets, or only at their sizes and top-k, needs.  The resident Sets belong to the
 *                             handle's last dpf_query_batch* call and are dropped by any other query, hash, rerank or fit call. */
int dpf_query_batch(dpf_index *ix, const int32_t *qids, const void *Q, int64_t nq, int32_t steps,
                    int64_t *cand_offsets, int32_t **cand_ids);
int dpf_query_batch_view(dpf_index *ix, const int32_t *qids, const void *Q, int64_t nq, int32_t steps,
                         int64_t *cand_offsets, const int32_t **cand_ids);
int dpf_query_batch_resident(dpf_index *ix, const int32_t *qids, const void *Q, int64_t nq, int32_t steps,
                             int64_t *cand_offsets);
int dpf_sets_fetch(dpf_index *ix, int64_t q0, int64_t q1, int32_t *out);
/* queries of the resident batch / ids in the resident Sets of the queries [q0, q1); -1 when there are no resident Sets
 * (or the range is outside the batch): lets a binding check its buffer sizes against the library's own offsets */
int64_t dpf_sets_queries(const dpf_index *ix);
int64_t dpf_sets_count(const dpf_index *ix, int64_t q0, int64_t q1);
int dpf_sets_topk(dpf_index *ix, int32_t k, int32_t *ids, double *scores);
/* releases a block dpf_query_batch returned */
void dpf_free(void *p);

/* The query + rerank of DensevectorRDFInit.topKAndPrecisionScore (Dense:472-507)
 * without the ground-truth scoring: ids[nq][k] (-1 padded) and scores[nq][k]
 * (NaN padded), best first.  Host pointers.  Arguments and results in pinned host memory (dpf_host_alloc,
 * hipHostMalloc, hipHostRegister) are copied by the device's copy engines from / to where they lie; ordinary
 * (pageable) memory goes through a pinned staging area of the handle, piece by piece. */
int dpf_query_topk(dpf_index *ix, const int32_t *qids, const void *Q, int64_t nq, int32_t steps,
                   int32_t k, int32_t *ids, double *scores);
/* Pinned host memory for the host-pointer entry points (no counterpart in the reference, whose arrays live on
 * the JVM heap; a JVM caller wraps the block in a direct ByteBuffer).  dpf_host_free releases it. */
int dpf_host_alloc(size_t bytes, void **out);
void dpf_host_free(void *p);

/* Same, every buffer already in HBM, the kernels enqueued on `stream` (a hipStream_t, NULL =
 * the default stream).  The call BLOCKS: it waits on `stream` once per batch (twice on a handle's
 * first bucket-major batch or after a batch outgrew the carried buffers) to read the batch's list
 * sizes and overflow flags, may grow the handle's workspaces (hipMalloc / hipFree), and runs a batch
 * again when the buffers carried from earlier batches turn out too small.  On return the last
 * kernels of the batch (the exact rerank writing d_ids / d_scores) may still be running on `stream`:
 * order later work on that stream, or synchronise it, before reading the results.  (Some kernels
 * of a batch run on a stream of the handle's own, forked from `stream` and joined to it again with
 * events inside the call: DPF_OPT_SIDE_STREAM.) */
int dpf_query_topk_device(dpf_index *ix, const int32_t *d_qids, const void *d_Q, int64_t nq,
                          int32_t steps, int32_t k, int32_t *d_ids, double *d_scores, void *stream);

/* dpf_query_topk_device for callers that cannot allocate device memory (tight serving loops, harnesses
 * that forbid hipMalloc).  It has no counterpart in the reference.  Same arguments, same results bit for
 * bit (it takes the per-query gather path: candidate Set, exact fp64 scores), but every per-batch buffer -
 * padded query rows, hashes, sub-indexes, counts, offsets, the candidate lists - is a fixed slice of one
 * 29 MiB array in the library's device image, and the call makes no hipMalloc / hipFree.  Limits per call,
 * refused with DPF_ECAPACITY beyond them: 1024 queries, 4 MiB of padded query rows, 131072 (query, table)
 * pairs, and 6 291 456 candidate ids over all queries of the call (counted per query as min(ids visited,
 * rows in the shard)).  The call BLOCKS on `stream` once (a small device-to-host copy and a wait, to read
 * that total), so it cannot be recorded into a HIP graph.  All handles on one GPU share that GPU's array,
 * one call at a time: on the device a call's kernels are ordered (by an event made at dpf_create) behind
 * the last kernel of the previous call that used the array, whatever stream or handle it came from, so
 * calls issued one after the other need no synchronisation in between; a call made while another
 * thread's call is still inside the library on the same GPU is refused with DPF_EBUSY. */
int dpf_query_topk_device_static(dpf_index *ix, const int32_t *d_qids, const void *d_Q, int64_t nq,
                                 int32_t steps, int32_t k, int32_t *d_ids, double *d_scores, void *stream);

/* Dense:477-505 alone: rerank caller-supplied candidate lists (host pointers). */
int dpf_rerank(dpf_index *ix, const void *Q, int64_t nq, const int64_t *cand_offsets,
               const int32_t *cand_ids, int32_t k, int32_t *ids, double *scores);

/* Merge `parts` partial top-k lists per query (row- or table-sharded handles on
 * several devices, gathered by the caller; an id present in several lists is
 * taken once): in_ids/in_scores [parts][nq][k] device
 * memory -> d_ids/d_scores [nq][k].  No counterpart in the reference, which is
 * a single process. */
int dpf_merge_topk_device(dpf_index *ix, const int32_t *in_ids, const double *in_scores,
                          int32_t parts, int64_t nq, int32_t k, int32_t *d_ids, double *d_scores,
                          void *stream);
/* The same merge over `parts` packed blocks of part_bytes each, as ONE all-gather
 * of the handles' results delivers them: block s = scores [nq][k] (doubles) at its
 * start, ids [nq][k] (int32) right behind them, padding up to part_bytes (a
 * multiple of 8, at least nq*k*12; `in` 8-byte aligned).  No counterpart in the
 * reference, which is a single process. */
int dpf_merge_topk_packed_device(dpf_index *ix, const void *in, int64_t part_bytes, int32_t parts,
                                 int64_t nq, int32_t k, int32_t *d_ids, double *d_scores, void *stream);

/* A fitted handle to one file and back: configuration, hash family and
 * partitioner, the hashes of every id and the resident rows (the forest is
 * rebuilt from the hashes on load; results after a load are those of the handle
 * that was saved).  Stands where the reference persists its tables to disk
 * (RandomDrawTreeMap.java:2713-2773, the StoreAppend files); the format is this
 * library's own.  dpf_load: device < 0 keeps the saved device ordinal; errors
 * are reported through dpf_last_error(NULL). */
int dpf_save(dpf_index *ix, const char *path);
int dpf_load(const char *path, int32_t device, dpf_index **out);
/* One shard of a saved index: the file (saved by a handle that held every row) is memory-mapped and only what the
 * shard needs is copied to the device - the hashes of all ids (the replicated forest is rebuilt from them) and the
 * rows [row_begin, row_end) - so a base larger than one GPU's memory is brought up rank by rank from one file, each
 * rank naming its rows, tables and sub-index group.  `shard` supplies table_begin / table_end, subindex_group /
 * subindex_groups, row_begin / row_end (its other fields are ignored; NULL = the shard the file was saved with). */
int dpf_load_shard(const char *path, int32_t device, const dpf_config *shard, dpf_index **out);
/* How a handle walks the forest for dpf_query_batch / dpf_query_topk*:
 *   DPF_PROBE_FLIPS (default)  RandomDrawTreeMap.getSimilarWithStepWiseFaster (RDT:742-797): the single-bit flips
 *                              of the hash, never the hash itself; what DensevectorRDFInit queries with;
 *   DPF_PROBE_OWN              RandomDrawTreeMap.getSimilarWithStepWise(key, steps) (RDT:630-675): the hash itself
 *                              only, in every sub-index within `steps`; what SparsevectorRDFInit's keyed batch query
 *                              uses (SparsevectorRDFInit.scala:407).
 * Unlike the dpf_set_option switches this changes results. */
enum { DPF_PROBE_FLIPS = 0, DPF_PROBE_OWN = 1 };
int dpf_set_probe_mode(dpf_index *ix, int32_t mode);
/* The configuration a handle was created (or loaded) with. */
int dpf_get_config(const dpf_index *ix, dpf_config *out);

/* RandomDrawTreeMap.allSubIndexObjectsNumberDistribution via
 * DensevectorRDFInit.getDtAndHtNumDistribution (Dense:515-530):
 * out[2^partition_bits] = mean number of ids per sub-index over the tables. */
int dpf_subindex_distribution(dpf_index *ix, double *out);

int dpf_get_stats(dpf_index *ix, dpf_stats *out);

/* The flattened forest the probe walks, for inspection and cross-checks (what a dump of every
 * table's partitionRootRec / directory nodes / bucket chains, RDT:940-994, would show):
 * sizes[4] = element counts of roots[L][P][SEG], dnodes[(num_dnodes + 1)][dirNodeSize] (child word
 * 0 = empty, (bucket << 1) | 1, directory << 1), leaf_off[num_leaves + 1] and ids[] (bucket
 * contents, ascending; only this handle's row shard).  dpf_snapshot_copy fills host arrays of
 * those sizes; a NULL pointer skips that array. */
int dpf_snapshot_sizes(dpf_index *ix, int64_t *sizes);
int dpf_snapshot_copy(dpf_index *ix, uint32_t *roots, uint32_t *dnodes, uint32_t *leaf_off, int32_t *ids);

/* What the probe stage of the last bucket-major top-k batch counted, for inspection and cross-checks (what summing
 * the sizes of the lists RandomDrawTreeMap.getSimilarWithStepWiseFaster, RDT:742-797, concatenates would show):
 * bucket_counts[num_leaves] = queries that reached each bucket, ids_visited[nq] = bucket entries met per query over
 * its distinct buckets, buckets_reached[nq] = distinct non-empty buckets per query, bucket_lists[nq][list_cap] = the
 * bucket numbers themselves in no particular order (as far as the handle remembered them: DPF_OPT_BUCKET_LIST_CAP;
 * 0xffffffff padded).  NULL skips an array. */
int dpf_last_incidences(dpf_index *ix, uint32_t *bucket_counts, uint64_t *ids_visited, int32_t *buckets_reached,
                        uint32_t *bucket_lists, int32_t list_cap);

/* Per query of the last bucket-major top-k batch: rows the scan forwarded to the exact rerank (`forwarded`, before
 * duplicates are dropped) and rows the exact rerank scored in the end (`kept`: distinct, and not passed by the query's
 * final threshold).  For inspection; NULL skips an array. */
int dpf_last_forward_counts(dpf_index *ix, int32_t *forwarded, int32_t *kept);

/* Device time of the dominant kernel of the last *_topk* call, measured with HIP
 * events on the stream the kernel ran on, and the bytes that launch has to move by
 * its own definition: the gather rerank (dpf_stats.last_path 0) reads every
 * candidate row + id, the query rows, and writes the results; the bucket-major
 * scan (last_path 1) reads every touched bucket's rows + ids once, one query row +
 * threshold per (query, bucket) incidence, and writes the forwarded pairs.  On the
 * bucket-major path *ms and *algorithmic_bytes are sums over the scan kernel's
 * launches of the batch (dpf_stats.last_scan_launches of them). */
int dpf_last_rerank_profile(dpf_index *ix, double *ms, int64_t *algorithmic_bytes);
/* Tuning and test switches of one handle.  None of them changes a result; they choose between paths
 * that give identical answers (the parity tests pin the paths against each other).  No environment
 * variable is read by the query path.  value 0 = the library's own choice unless stated. */
enum {
    DPF_OPT_EMIT_CAP = 1,            /* entries per forward list of the bucket-major scan, pinned (default: 8192,
                                        doubled up to 65536 after a batch in which a list overflowed) */
    DPF_OPT_BUCKET_LIST_CAP = 2,     /* buckets remembered per query by the first probe pass (default: up to 4096) */
    DPF_OPT_SYNC_SIZES = 3,          /* 1: the host reads every batch's list sizes before filling the lists */
    DPF_OPT_FIRST_PASS_PERMILLE = 4, /* share of the scan's work units run before the thresholds are refined (150) */
    DPF_OPT_SEED_SAMPLE = 5,         /* candidates per query whose distances give the first thresholds: 16..4096, 0 = none, or
                                        -1 (default) = chosen at every fit from the forest's bucket sizes: 5 sqrt(tables x
                                        size-biased mean bucket size), within [128, 2048] */
    DPF_OPT_SHORT_RERANK = 6,        /* 0: short candidate lists use the general rerank kernel too (default 1) */
    DPF_OPT_PROJECTION_MIN_ROWS = 7, /* batches of at least this many vectors, with dim x num_functions >= 40000, are hashed
                                        through the matrix-core projection (fp64 MFMA contraction + exact re-evaluation of
                                        undecided rows; same hashes bit for bit); others by the exact vector-ALU kernel alone.
                                        Default 4096; 0 = never; 1 = always, whatever the shape */
    DPF_OPT_SCAN_PASSES = 8,         /* slices the scan's work units are run in, thresholds refined in between: 2, 3, or
                                        0 = 3 for rows of 16 steps of 32 dimensions and more and for forests whose seed sample
                                        comes out at 1024 or more (a query's own buckets hold tens of thousands of rows), 2 otherwise */
    DPF_OPT_EARLY_PASS_PERMILLE = 9, /* with 3 passes: share of the units in the first of them (40) */
    DPF_OPT_EXACT_SEEDS = 10,        /* 1: l2 handles score the seed sample exactly from the fp32 rows, as dot handles do
                                        (default 0: an upper bound of the k-th best distance from the fp16 copies) */
    DPF_OPT_KEEP_FORWARDED = 11,     /* 1: every row the scan forwarded is scored exactly (default 0: rows forwarded under a
                                        threshold the query has tightened past since are dropped before the exact rerank) */
    DPF_OPT_SIDE_STREAM = 12,        /* which kernels of a batch run on a stream of the handle's own beside the caller's (from the
                                        handle's second batch on; both streams are ordered behind the caller's and joined to it
                                        again before the scan): 0 = none, 2 = those that build the bucket lists (beside those that
                                        find the first radii), 1 = also the conversions of the query rows (beside the hashing),
                                        -1 (default) = 1 for handles that probe up to four tables, 2 otherwise */
    DPF_OPT_PROBE_LANES = 13,        /* the counting pass of the bucket-major path: 1 = a lane per (query, table, sub-index) that
                                        walks the hash's own path once and reads the sibling slots on it, 0 = a lane per probe (one
                                        bit flip of the hash) as the pass that materialises the Sets does, -1 (default) = the first
                                        where a query probes at most 16 (table, sub-index) pairs, the second otherwise */
    DPF_OPT_DEFER_BUILD = 14,        /* 1: dpf_fit only appends - copy-in, hashing, the resident rows - and leaves the forest and the
                                        scan's copies unbuilt (queries are refused with DPF_ENOTFIT meanwhile); the first dpf_fit
                                        after the option is cleared (n = 0 will do) builds them once over everything appended.
                                        For a base streamed in chunk by chunk: one forest build instead of one per chunk */
    DPF_OPT_LIST_BINS = 16,          /* 0: the bucket lists of a batch are always built from a global incidence histogram (returning
                                        atomics), its scan and scattered stores.  Default 1: from the handle's second batch on the
                                        counting pass writes (bucket, query) pairs into per-bin regions - a bin is a range of buckets;
                                        the bins are cut from the first batch's counts to hold about equally many pairs - and one
                                        workgroup per bin builds counts, offsets and lists in LDS.  2: as 1, the list entries stored
                                        one by one instead of staged in LDS (measurement) */
    DPF_OPT_EXPECTED_ROWS = 15       /* ids the caller is going to fit in all (0 = unknown): the next dpf_fit sizes the row and hash
                                        buffers for them at once instead of regrowing them chunk by chunk */
};
int dpf_set_option(dpf_index *ix, int32_t option, int64_t value);

/* enable (1) / disable (0) the per-stage HIP-event timing: the events are recorded only for batches
 * that run while it is on (an event between two kernels costs a few microseconds of stream time);
 * dpf_last_rerank_profile and the last_ms_* statistics need it on for the batch they describe */
int dpf_set_profiling(dpf_index *ix, int32_t on);

#ifdef __cplusplus
}
#endif
#endif
