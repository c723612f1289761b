/*
 * dpf_jni.c -- JNI shim between the reference's Scala objects and the C ABI of include/dpf.h.
 *
 * Thin on purpose: every native method of mclab.gpu.Dpf (jni/scala/mclab/gpu/Dpf.scala) pins its
 * arrays, calls ONE dpf_* function and releases them.  No logic lives here.  The Scala side that
 * calls it replaces the bodies of DensevectorRDFInit (src/main/scala/mclab/deploy/
 * DensevectorRDFInit.scala:50-118, 161-206, 335-360, 372-399, 453-458, 472-507 of the reference) and
 * is in jni/scala/mclab/deploy/DensevectorRDFInit.scala.
 *
 * Build (needs a JDK for jni.h; see jni/CMakeLists.txt):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../include dpf_jni.c \
 *       -L../similaritysearchbyrdf_amd -ldpf -o libdpf_jni.so
 *
 * Errors: a failing dpf_* call throws java.lang.IllegalStateException carrying dpf_last_error();
 * DPF_ENOTFIT throws java.lang.NullPointerException, which is what the reference's query tasks catch to
 * print "need to fit the data first" (DensevectorRDFInit.scala:422-424).
 */
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include "dpf.h"

#define DPF_JNI_CONFIG_INTS 20      /* the int32 fields of dpf_config, in declaration order */

static dpf_index *handle_of(jlong h) { return (dpf_index *)(intptr_t)h; }

static void throw_dpf(JNIEnv *env, const dpf_index *ix, int rc)
{
    if (rc == DPF_OK) return;
    jclass ex = (*env)->FindClass(env, rc == DPF_ENOTFIT ? "java/lang/NullPointerException"
                                                        : "java/lang/IllegalStateException");
    if (ex) (*env)->ThrowNew(env, ex, dpf_last_error(ix));
}

/* create(cfgInts[20], rowBegin, rowEnd): cfgInts holds the int32 fields of dpf_config in the order of
 * Dpf.ConfigIntFields; every field is assigned by name, so a field added to the struct without a line
 * here stays zero and tests/test_jni_shim.py fails. */
JNIEXPORT jlong JNICALL Java_mclab_gpu_Dpf_create(JNIEnv *env, jclass cls, jintArray jcfg, jlong rowBegin, jlong rowEnd)
{
    (void)cls;
    if ((*env)->GetArrayLength(env, jcfg) != DPF_JNI_CONFIG_INTS) {
        jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        if (ex) (*env)->ThrowNew(env, ex, "Dpf.create: cfgInts must hold the 20 int fields of dpf_config");
        return 0;
    }
    jint v[DPF_JNI_CONFIG_INTS];
    (*env)->GetIntArrayRegion(env, jcfg, 0, DPF_JNI_CONFIG_INTS, v);
    dpf_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.family = (int32_t)v[0];
    cfg.dim = (int32_t)v[1];
    cfg.num_functions = (int32_t)v[2];
    cfg.num_tables = (int32_t)v[3];
    cfg.chain_length = (int32_t)v[4];
    cfg.partition_bits = (int32_t)v[5];
    cfg.bucket_bits = (int32_t)v[6];
    cfg.dir_node_size = (int32_t)v[7];
    cfg.bucket_overflow = (int32_t)v[8];
    cfg.type_of_index = (int32_t)v[9];
    cfg.exclude_self = (int32_t)v[10];
    cfg.metric = (int32_t)v[11];
    cfg.top_k = (int32_t)v[12];
    cfg.dtype = (int32_t)v[13];
    cfg.device = (int32_t)v[14];
    cfg.flags = (int32_t)v[15];
    cfg.table_begin = (int32_t)v[16];
    cfg.table_end = (int32_t)v[17];
    cfg.subindex_group = (int32_t)v[18];
    cfg.subindex_groups = (int32_t)v[19];
    cfg.row_begin = (int64_t)rowBegin;
    cfg.row_end = (int64_t)rowEnd;
    dpf_index *ix = NULL;
    int rc = dpf_create(&cfg, &ix);
    throw_dpf(env, NULL, rc);
    return (jlong)(intptr_t)ix;
}

JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_destroy(JNIEnv *env, jclass cls, jlong h)
{
    (void)env; (void)cls;
    dpf_destroy(handle_of(h));
}

JNIEXPORT jstring JNICALL Java_mclab_gpu_Dpf_lastError(JNIEnv *env, jclass cls, jlong h)
{
    (void)cls;
    return (*env)->NewStringUTF(env, dpf_last_error(handle_of(h)));
}

/* setHashFamily(h, fns[F*dim], b[F] or null, w[F] or null, chainFn[L*chainLength]) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_setHashFamily(JNIEnv *env, jclass cls, jlong h, jdoubleArray fns, jdoubleArray b,
                                                        jintArray w, jintArray chainFn)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    jdouble *pf = (*env)->GetPrimitiveArrayCritical(env, fns, NULL);
    jdouble *pb = b ? (*env)->GetPrimitiveArrayCritical(env, b, NULL) : NULL;
    jint *pw = w ? (*env)->GetPrimitiveArrayCritical(env, w, NULL) : NULL;
    jint *pc = (*env)->GetPrimitiveArrayCritical(env, chainFn, NULL);
    int rc = dpf_set_hash_family(ix, pf, pb, (const int32_t *)pw, (const int32_t *)pc);
    (*env)->ReleasePrimitiveArrayCritical(env, chainFn, pc, JNI_ABORT);
    if (w) (*env)->ReleasePrimitiveArrayCritical(env, w, pw, JNI_ABORT);
    if (b) (*env)->ReleasePrimitiveArrayCritical(env, b, pb, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, fns, pf, JNI_ABORT);
    throw_dpf(env, ix, rc);
}

/* setPartitioner(h, partFns[L*partitionBits*32], partB[L*partitionBits] or null, partW[...] or null) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_setPartitioner(JNIEnv *env, jclass cls, jlong h, jdoubleArray partFns,
                                                         jdoubleArray partB, jintArray partW)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    jdouble *pf = (*env)->GetPrimitiveArrayCritical(env, partFns, NULL);
    jdouble *pb = partB ? (*env)->GetPrimitiveArrayCritical(env, partB, NULL) : NULL;
    jint *pw = partW ? (*env)->GetPrimitiveArrayCritical(env, partW, NULL) : NULL;
    int rc = dpf_set_partitioner(ix, pf, pb, (const int32_t *)pw);
    if (partW) (*env)->ReleasePrimitiveArrayCritical(env, partW, pw, JNI_ABORT);
    if (partB) (*env)->ReleasePrimitiveArrayCritical(env, partB, pb, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, partFns, pf, JNI_ABORT);
    throw_dpf(env, ix, rc);
}

/* setSamplingIndex(h, samplingIndex[32]) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_setSamplingIndex(JNIEnv *env, jclass cls, jlong h, jintArray idx)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    if ((*env)->GetArrayLength(env, idx) != 32) { throw_dpf(env, ix, DPF_EINVAL); return; }
    jint v[32];
    (*env)->GetIntArrayRegion(env, idx, 0, 32, v);
    throw_dpf(env, ix, dpf_set_sampling_index(ix, (const int32_t *)v));
}

/* fit(h, rows[n*dim] doubles, n): the handle must have been created with dtype = DPF_F64 */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_fit(JNIEnv *env, jclass cls, jlong h, jdoubleArray rows, jlong n)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    jdouble *p = (*env)->GetPrimitiveArrayCritical(env, rows, NULL);
    int rc = dpf_fit(ix, p, (int64_t)n);
    (*env)->ReleasePrimitiveArrayCritical(env, rows, p, JNI_ABORT);
    throw_dpf(env, ix, rc);
}

/* fitFloat(h, rows[n*dim] floats, n): for handles created with dtype = DPF_F32 */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_fitFloat(JNIEnv *env, jclass cls, jlong h, jfloatArray rows, jlong n)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    jfloat *p = (*env)->GetPrimitiveArrayCritical(env, rows, NULL);
    int rc = dpf_fit(ix, p, (int64_t)n);
    (*env)->ReleasePrimitiveArrayCritical(env, rows, p, JNI_ABORT);
    throw_dpf(env, ix, rc);
}

/* hashBatch(h, q[nq*dim] doubles, nq, hashes[nq*L], parts[nq*L] or null) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_hashBatch(JNIEnv *env, jclass cls, jlong h, jdoubleArray q, jlong nq,
                                                    jintArray hashes, jintArray parts)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    jdouble *pq = (*env)->GetPrimitiveArrayCritical(env, q, NULL);
    jint *ph = (*env)->GetPrimitiveArrayCritical(env, hashes, NULL);
    jint *pp = parts ? (*env)->GetPrimitiveArrayCritical(env, parts, NULL) : NULL;
    int rc = dpf_hash_batch(ix, pq, (int64_t)nq, (int32_t *)ph, (int32_t *)pp);
    if (parts) (*env)->ReleasePrimitiveArrayCritical(env, parts, pp, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, hashes, ph, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, q, pq, JNI_ABORT);
    throw_dpf(env, ix, rc);
}

/* queryBatch(h, qids[nq], q[nq*dim] doubles, nq, steps, offsets[nq+1]) -> ids: per query i the sorted unique
 * candidate ids are ids[offsets(i) until offsets(i+1)].  An unfitted index returns an empty array and all-zero
 * offsets after throwing NullPointerException (DensevectorRDFInit.scala:422-424). */
JNIEXPORT jintArray JNICALL Java_mclab_gpu_Dpf_queryBatch(JNIEnv *env, jclass cls, jlong h, jintArray qids, jdoubleArray q,
                                                          jlong nq, jint steps, jlongArray offsets)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    jint *pq = (*env)->GetPrimitiveArrayCritical(env, qids, NULL);
    jdouble *pv = (*env)->GetPrimitiveArrayCritical(env, q, NULL);
    jlong *po = (*env)->GetPrimitiveArrayCritical(env, offsets, NULL);
    int32_t *ids = NULL;
    int rc = dpf_query_batch(ix, (const int32_t *)pq, pv, (int64_t)nq, (int32_t)steps, (int64_t *)po, &ids);
    jlong total = rc == DPF_OK ? po[nq] : 0;
    (*env)->ReleasePrimitiveArrayCritical(env, offsets, po, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, q, pv, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, qids, pq, JNI_ABORT);
    if (rc != DPF_OK) { dpf_free(ids); throw_dpf(env, ix, rc); return NULL; }
    jintArray out = (*env)->NewIntArray(env, (jsize)total);
    if (out && total > 0) (*env)->SetIntArrayRegion(env, out, 0, (jsize)total, (const jint *)ids);
    dpf_free(ids);
    return out;
}

/* queryTopK(h, qids[nq], q[nq*dim] doubles, nq, steps, k, ids[nq*k], scores[nq*k]): ids -1 / scores NaN padded */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_queryTopK(JNIEnv *env, jclass cls, jlong h, jintArray qids, jdoubleArray q, jlong nq,
                                                    jint steps, jint k, jintArray ids, jdoubleArray scores)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    jint *pq = (*env)->GetPrimitiveArrayCritical(env, qids, NULL);
    jdouble *pv = (*env)->GetPrimitiveArrayCritical(env, q, NULL);
    jint *pi = (*env)->GetPrimitiveArrayCritical(env, ids, NULL);
    jdouble *ps = (*env)->GetPrimitiveArrayCritical(env, scores, NULL);
    int rc = dpf_query_topk(ix, (const int32_t *)pq, pv, (int64_t)nq, (int32_t)steps, (int32_t)k, (int32_t *)pi, ps);
    (*env)->ReleasePrimitiveArrayCritical(env, scores, ps, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, ids, pi, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, q, pv, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, qids, pq, JNI_ABORT);
    throw_dpf(env, ix, rc);
}

/* rerank(h, q[nq*dim], nq, candOffsets[nq+1], candIds[], k, ids[nq*k], scores[nq*k]) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_rerank(JNIEnv *env, jclass cls, jlong h, jdoubleArray q, jlong nq, jlongArray candOffsets,
                                                 jintArray candIds, jint k, jintArray ids, jdoubleArray scores)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    jdouble *pv = (*env)->GetPrimitiveArrayCritical(env, q, NULL);
    jlong *po = (*env)->GetPrimitiveArrayCritical(env, candOffsets, NULL);
    jint *pc = (*env)->GetPrimitiveArrayCritical(env, candIds, NULL);
    jint *pi = (*env)->GetPrimitiveArrayCritical(env, ids, NULL);
    jdouble *ps = (*env)->GetPrimitiveArrayCritical(env, scores, NULL);
    int rc = dpf_rerank(ix, pv, (int64_t)nq, (const int64_t *)po, (const int32_t *)pc, (int32_t)k, (int32_t *)pi, ps);
    (*env)->ReleasePrimitiveArrayCritical(env, scores, ps, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, ids, pi, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, candIds, pc, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, candOffsets, po, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, q, pv, JNI_ABORT);
    throw_dpf(env, ix, rc);
}

/* subindexDistribution(h, out[2^partitionBits]) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_subindexDistribution(JNIEnv *env, jclass cls, jlong h, jdoubleArray out)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    jdouble *p = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
    int rc = dpf_subindex_distribution(ix, p);
    (*env)->ReleasePrimitiveArrayCritical(env, out, p, 0);
    throw_dpf(env, ix, rc);
}

/* counters(h, out[6]) = {num_vectors, num_dnodes, num_leaves, max_leaf, misflag_events, short_hash_events} */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_counters(JNIEnv *env, jclass cls, jlong h, jlongArray out)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    dpf_stats st;
    int rc = dpf_get_stats(ix, &st);
    if (rc == DPF_OK && (*env)->GetArrayLength(env, out) >= 6) {
        jlong v[6] = {(jlong)st.num_vectors, (jlong)st.num_dnodes, (jlong)st.num_leaves, (jlong)st.max_leaf,
                      (jlong)st.misflag_events, (jlong)st.short_hash_events};
        (*env)->SetLongArrayRegion(env, out, 0, 6, v);
    }
    throw_dpf(env, ix, rc);
}

/* setProbeMode(h, 0 = bit flips of the hash (RDT:742-797), 1 = the hash itself only (RDT:630-675)) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_setProbeMode(JNIEnv *env, jclass cls, jlong h, jint mode)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    throw_dpf(env, ix, dpf_set_probe_mode(ix, (int32_t)mode));
}

JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_save(JNIEnv *env, jclass cls, jlong h, jstring path)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    int rc = p ? dpf_save(ix, p) : DPF_ENOMEM;
    if (p) (*env)->ReleaseStringUTFChars(env, path, p);
    throw_dpf(env, ix, rc);
}

JNIEXPORT jlong JNICALL Java_mclab_gpu_Dpf_load(JNIEnv *env, jclass cls, jstring path, jint device)
{
    (void)cls;
    dpf_index *ix = NULL;
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    int rc = p ? dpf_load(p, (int32_t)device, &ix) : DPF_ENOMEM;
    if (p) (*env)->ReleaseStringUTFChars(env, path, p);
    throw_dpf(env, NULL, rc);
    return (jlong)(intptr_t)ix;
}

/* loadShard(path, device, shard = {table_begin, table_end, subindex_group, subindex_groups}, rowBegin, rowEnd): one rank's
 * shard of a saved index (the file is memory-mapped; only the shard's rows are copied to the device) */
JNIEXPORT jlong JNICALL Java_mclab_gpu_Dpf_loadShard(JNIEnv *env, jclass cls, jstring path, jint device, jintArray jshard,
                                                     jlong rowBegin, jlong rowEnd)
{
    (void)cls;
    if ((*env)->GetArrayLength(env, jshard) != 4) {
        jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        if (ex) (*env)->ThrowNew(env, ex, "Dpf.loadShard: shard must hold table_begin, table_end, subindex_group, subindex_groups");
        return 0;
    }
    jint v[4];
    (*env)->GetIntArrayRegion(env, jshard, 0, 4, v);
    dpf_config shard;
    memset(&shard, 0, sizeof shard);
    shard.table_begin = (int32_t)v[0];
    shard.table_end = (int32_t)v[1];
    shard.subindex_group = (int32_t)v[2];
    shard.subindex_groups = (int32_t)v[3];
    shard.row_begin = (int64_t)rowBegin;
    shard.row_end = (int64_t)rowEnd;
    dpf_index *ix = NULL;
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    int rc = p ? dpf_load_shard(p, (int32_t)device, &shard, &ix) : DPF_ENOMEM;
    if (p) (*env)->ReleaseStringUTFChars(env, path, p);
    throw_dpf(env, NULL, rc);
    return (jlong)(intptr_t)ix;
}

/* setOption(h, DPF_OPT_*, value): explicit tuning; no environment variable is read on the query path */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_setOption(JNIEnv *env, jclass cls, jlong h, jint option, jlong value)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    throw_dpf(env, ix, dpf_set_option(ix, (int32_t)option, (int64_t)value));
}
