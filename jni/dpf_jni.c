/*
 * dpf_jni.c -- JNI shim between the reference's Scala objects and the C ABI of include/dpf.h.
 *
 * Thin on purpose: every native method of mclab.gpu.Dpf (jni/scala/mclab/gpu/Dpf.scala) checks its array
 * lengths against what the handle's configuration implies, calls ONE dpf_* function and hands the results
 * back.  No logic lives here.  The Scala side that calls it replaces the bodies of DensevectorRDFInit
 * (src/main/scala/mclab/deploy/DensevectorRDFInit.scala:50-118, 161-206, 335-360, 372-399, 453-458, 472-507
 * of the reference) and is in jni/scala/mclab/deploy/DensevectorRDFInit.scala.
 *
 * Arrays: the dpf_* calls block on GPU work (copies, stream waits, seconds for a large fit), so no JNI
 * critical region is held across them.  Inputs are taken with Get<Type>ArrayElements (the JVM pins or
 * copies as it sees fit; the collector keeps running) and released with JNI_ABORT; outputs are written
 * by the library into native buffers and copied into the Java arrays with Set<Type>ArrayRegion afterwards.
 * Every length is checked before the call (IllegalArgumentException on a short array: the library writes
 * nq * k, nq + 1, ... elements through plain pointers), a null qids is passed on as NULL (no
 * self-exclusion), and a result that does not fit a Java array (more than 2^31 - 1 ints) is refused.
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
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
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

static void throw_iae(JNIEnv *env, const char *msg)
{
    jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (ex) (*env)->ThrowNew(env, ex, msg);
}

static void throw_oom(JNIEnv *env, const char *msg)
{
    jclass ex = (*env)->FindClass(env, "java/lang/OutOfMemoryError");
    if (ex) (*env)->ThrowNew(env, ex, msg);
}

/* 1 when the array holds at least `want` elements (a null array only when may_be_null); else throws and returns 0 */
static int long_enough(JNIEnv *env, jarray a, int64_t want, int may_be_null, const char *what)
{
    if (!a) {
        if (may_be_null) return 1;
        throw_iae(env, what);
        return 0;
    }
    if (want < 0 || (int64_t)(*env)->GetArrayLength(env, a) < want) { throw_iae(env, what); return 0; }
    return 1;
}

/* the handle's configuration; throws and returns 0 for a null handle */
static int config_of(JNIEnv *env, dpf_index *ix, dpf_config *cfg)
{
    if (!ix || dpf_get_config(ix, cfg) != DPF_OK) { throw_iae(env, "Dpf: null handle"); return 0; }
    return 1;
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
    dpf_config c;
    if (!config_of(env, ix, &c)) return;
    const int64_t F = c.num_functions;
    if (!long_enough(env, fns, F * c.dim, 0, "Dpf.setHashFamily: fns must hold num_functions * dim doubles") ||
        !long_enough(env, b, F, 1, "Dpf.setHashFamily: b must hold num_functions doubles") ||
        !long_enough(env, w, F, 1, "Dpf.setHashFamily: w must hold num_functions ints") ||
        !long_enough(env, chainFn, (int64_t)c.num_tables * c.chain_length, 0, "Dpf.setHashFamily: chainFn must hold num_tables * chain_length ints"))
        return;
    jdouble *pf = (*env)->GetDoubleArrayElements(env, fns, NULL);
    jdouble *pb = b ? (*env)->GetDoubleArrayElements(env, b, NULL) : NULL;
    jint *pw = w ? (*env)->GetIntArrayElements(env, w, NULL) : NULL;
    jint *pc = (*env)->GetIntArrayElements(env, chainFn, NULL);
    const int pinned = pf && pc && (!b || pb) && (!w || pw);
    int rc = pinned ? dpf_set_hash_family(ix, pf, pb, (const int32_t *)pw, (const int32_t *)pc) : DPF_OK;
    if (pc) (*env)->ReleaseIntArrayElements(env, chainFn, pc, JNI_ABORT);
    if (pw) (*env)->ReleaseIntArrayElements(env, w, pw, JNI_ABORT);
    if (pb) (*env)->ReleaseDoubleArrayElements(env, b, pb, JNI_ABORT);
    if (pf) (*env)->ReleaseDoubleArrayElements(env, fns, pf, JNI_ABORT);
    if (!pinned) return;                                  /* the JVM has thrown OutOfMemoryError */
    throw_dpf(env, ix, rc);
}

/* setPartitioner(h, partFns[L*partitionBits*32], partB[L*partitionBits] or null, partW[...] or null) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_setPartitioner(JNIEnv *env, jclass cls, jlong h, jdoubleArray partFns,
                                                         jdoubleArray partB, jintArray partW)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    dpf_config c;
    if (!config_of(env, ix, &c)) return;
    const int64_t lp = (int64_t)c.num_tables * c.partition_bits;
    if (!long_enough(env, partFns, lp * 32, 0, "Dpf.setPartitioner: partFns must hold num_tables * partition_bits * 32 doubles") ||
        !long_enough(env, partB, lp, 1, "Dpf.setPartitioner: partB must hold num_tables * partition_bits doubles") ||
        !long_enough(env, partW, lp, 1, "Dpf.setPartitioner: partW must hold num_tables * partition_bits ints"))
        return;
    jdouble *pf = (*env)->GetDoubleArrayElements(env, partFns, NULL);
    jdouble *pb = partB ? (*env)->GetDoubleArrayElements(env, partB, NULL) : NULL;
    jint *pw = partW ? (*env)->GetIntArrayElements(env, partW, NULL) : NULL;
    const int pinned = pf && (!partB || pb) && (!partW || pw);
    int rc = pinned ? dpf_set_partitioner(ix, pf, pb, (const int32_t *)pw) : DPF_OK;
    if (pw) (*env)->ReleaseIntArrayElements(env, partW, pw, JNI_ABORT);
    if (pb) (*env)->ReleaseDoubleArrayElements(env, partB, pb, JNI_ABORT);
    if (pf) (*env)->ReleaseDoubleArrayElements(env, partFns, pf, JNI_ABORT);
    if (!pinned) return;
    throw_dpf(env, ix, rc);
}

/* setSamplingIndex(h, samplingIndex[32]) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_setSamplingIndex(JNIEnv *env, jclass cls, jlong h, jintArray idx)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    if (!idx || (*env)->GetArrayLength(env, idx) != 32) { throw_iae(env, "Dpf.setSamplingIndex: 32 entries"); return; }
    jint v[32];
    (*env)->GetIntArrayRegion(env, idx, 0, 32, v);
    throw_dpf(env, ix, dpf_set_sampling_index(ix, (const int32_t *)v));
}

/* fit(h, rows[n*dim] doubles, n): the handle must have been created with dtype = DPF_F64 */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_fit(JNIEnv *env, jclass cls, jlong h, jdoubleArray rows, jlong n)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    dpf_config c;
    if (!config_of(env, ix, &c)) return;
    if (c.dtype != DPF_F64) { throw_iae(env, "Dpf.fit: the handle stores float32 rows, use fitFloat"); return; }
    if (n < 0 || n > INT_MAX || !long_enough(env, rows, (int64_t)n * c.dim, 0, "Dpf.fit: rows must hold n * dim doubles")) {
        if (n < 0 || n > INT_MAX) throw_iae(env, "Dpf.fit: n out of range");
        return;
    }
    jdouble *p = (*env)->GetDoubleArrayElements(env, rows, NULL);
    if (!p) return;
    int rc = dpf_fit(ix, p, (int64_t)n);
    (*env)->ReleaseDoubleArrayElements(env, rows, p, JNI_ABORT);
    throw_dpf(env, ix, rc);
}

/* fitFloat(h, rows[n*dim] floats, n): for handles created with dtype = DPF_F32 */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_fitFloat(JNIEnv *env, jclass cls, jlong h, jfloatArray rows, jlong n)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    dpf_config c;
    if (!config_of(env, ix, &c)) return;
    if (c.dtype != DPF_F32) { throw_iae(env, "Dpf.fitFloat: the handle stores float64 rows, use fit"); return; }
    if (n < 0 || n > INT_MAX || !long_enough(env, rows, (int64_t)n * c.dim, 0, "Dpf.fitFloat: rows must hold n * dim floats")) {
        if (n < 0 || n > INT_MAX) throw_iae(env, "Dpf.fitFloat: n out of range");
        return;
    }
    jfloat *p = (*env)->GetFloatArrayElements(env, rows, NULL);
    if (!p) return;
    int rc = dpf_fit(ix, p, (int64_t)n);
    (*env)->ReleaseFloatArrayElements(env, rows, p, JNI_ABORT);
    throw_dpf(env, ix, rc);
}

/* The query methods below take double rows: handles created with dtype = DPF_F64 (what the Scala object creates). */
static int doubles_handle(JNIEnv *env, const dpf_config *c, jlong nq)
{
    if (c->dtype != DPF_F64) { throw_iae(env, "Dpf: this method passes double rows and needs a handle created with dtype = DPF_F64"); return 0; }
    if (nq < 0 || nq > INT_MAX) { throw_iae(env, "Dpf: nq out of range"); return 0; }
    return 1;
}

/* hashBatch(h, q[nq*dim] doubles, nq, hashes[nq*L], parts[nq*L] or null) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_hashBatch(JNIEnv *env, jclass cls, jlong h, jdoubleArray q, jlong nq,
                                                    jintArray hashes, jintArray parts)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    dpf_config c;
    if (!config_of(env, ix, &c) || !doubles_handle(env, &c, nq)) return;
    const int64_t words = (int64_t)nq * c.num_tables;
    if (words > INT_MAX) { throw_iae(env, "Dpf.hashBatch: nq * num_tables exceeds a Java array"); return; }
    if (!long_enough(env, q, (int64_t)nq * c.dim, 0, "Dpf.hashBatch: q must hold nq * dim doubles") ||
        !long_enough(env, hashes, words, 0, "Dpf.hashBatch: hashes must hold nq * num_tables ints") ||
        !long_enough(env, parts, words, 1, "Dpf.hashBatch: parts must hold nq * num_tables ints"))
        return;
    int32_t *oh = (int32_t *)malloc((size_t)(words > 0 ? words : 1) * 4);
    int32_t *op = parts ? (int32_t *)malloc((size_t)(words > 0 ? words : 1) * 4) : NULL;
    jdouble *pq = oh && (!parts || op) ? (*env)->GetDoubleArrayElements(env, q, NULL) : NULL;
    if (!pq) {
        if (!oh || (parts && !op)) throw_oom(env, "Dpf.hashBatch: native result buffers");
        free(oh); free(op);
        return;
    }
    int rc = dpf_hash_batch(ix, pq, (int64_t)nq, oh, op);
    (*env)->ReleaseDoubleArrayElements(env, q, pq, JNI_ABORT);
    if (rc == DPF_OK && words > 0) {
        (*env)->SetIntArrayRegion(env, hashes, 0, (jsize)words, (const jint *)oh);
        if (parts) (*env)->SetIntArrayRegion(env, parts, 0, (jsize)words, (const jint *)op);
    }
    free(oh); free(op);
    throw_dpf(env, ix, rc);
}

/* queryBatch(h, qids[nq] or null, q[nq*dim] doubles, nq, steps, offsets[nq+1]) -> ids: per query i the sorted unique
 * candidate ids are ids[offsets(i) until offsets(i+1)].  An unfitted index throws NullPointerException after setting
 * all offsets to zero (DensevectorRDFInit.scala:422-424). */
JNIEXPORT jintArray JNICALL Java_mclab_gpu_Dpf_queryBatch(JNIEnv *env, jclass cls, jlong h, jintArray qids, jdoubleArray q,
                                                          jlong nq, jint steps, jlongArray offsets)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    dpf_config c;
    if (!config_of(env, ix, &c) || !doubles_handle(env, &c, nq)) return NULL;
    if (!long_enough(env, qids, nq, 1, "Dpf.queryBatch: qids must hold nq ints") ||
        !long_enough(env, q, (int64_t)nq * c.dim, 0, "Dpf.queryBatch: q must hold nq * dim doubles") ||
        !long_enough(env, offsets, (int64_t)nq + 1, 0, "Dpf.queryBatch: offsets must hold nq + 1 longs"))
        return NULL;
    int64_t *po = (int64_t *)malloc(((size_t)nq + 1) * 8);
    if (!po) { throw_oom(env, "Dpf.queryBatch: native offsets"); return NULL; }
    jint *pq = qids ? (*env)->GetIntArrayElements(env, qids, NULL) : NULL;
    jdouble *pv = (!qids || pq) ? (*env)->GetDoubleArrayElements(env, q, NULL) : NULL;
    if (!pv) {
        if (pq) (*env)->ReleaseIntArrayElements(env, qids, pq, JNI_ABORT);
        free(po);
        return NULL;
    }
    const int32_t *ids = NULL;                            /* the handle's own pinned buffer: copied into the Java array below */
    int rc = dpf_query_batch_view(ix, (const int32_t *)pq, pv, (int64_t)nq, (int32_t)steps, po, &ids);
    (*env)->ReleaseDoubleArrayElements(env, q, pv, JNI_ABORT);
    if (pq) (*env)->ReleaseIntArrayElements(env, qids, pq, JNI_ABORT);
    const int64_t total = rc == DPF_OK ? po[nq] : 0;
    if (rc == DPF_OK || rc == DPF_ENOTFIT) (*env)->SetLongArrayRegion(env, offsets, 0, (jsize)(nq + 1), (const jlong *)po);
    free(po);
    if (rc != DPF_OK) { throw_dpf(env, ix, rc); return NULL; }
    if (total > INT_MAX) {
        jclass ex = (*env)->FindClass(env, "java/lang/IllegalStateException");
        if (ex) (*env)->ThrowNew(env, ex, "Dpf.queryBatch: the batch's candidate lists exceed a Java array (2^31 - 1 ids): use fewer queries per call");
        return NULL;
    }
    jintArray out = (*env)->NewIntArray(env, (jsize)total);
    if (out && total > 0) (*env)->SetIntArrayRegion(env, out, 0, (jsize)total, (const jint *)ids);
    return out;
}

/* queryBatchResident(h, qids[nq] or null, q[nq*dim] doubles, nq, steps, offsets[nq+1]): the Sets stay on the device, only
 * their sizes come back (offsets); setsFetch / setsTopK read them until the handle's next query, hash, rerank or fit call */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_queryBatchResident(JNIEnv *env, jclass cls, jlong h, jintArray qids, jdoubleArray q,
                                                             jlong nq, jint steps, jlongArray offsets)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    dpf_config c;
    if (!config_of(env, ix, &c) || !doubles_handle(env, &c, nq)) return;
    if (!long_enough(env, qids, nq, 1, "Dpf.queryBatchResident: qids must hold nq ints") ||
        !long_enough(env, q, (int64_t)nq * c.dim, 0, "Dpf.queryBatchResident: q must hold nq * dim doubles") ||
        !long_enough(env, offsets, (int64_t)nq + 1, 0, "Dpf.queryBatchResident: offsets must hold nq + 1 longs"))
        return;
    int64_t *po = (int64_t *)malloc(((size_t)nq + 1) * 8);
    if (!po) { throw_oom(env, "Dpf.queryBatchResident: native offsets"); return; }
    jint *pq = qids ? (*env)->GetIntArrayElements(env, qids, NULL) : NULL;
    jdouble *pv = (!qids || pq) ? (*env)->GetDoubleArrayElements(env, q, NULL) : NULL;
    if (!pv) {
        if (pq) (*env)->ReleaseIntArrayElements(env, qids, pq, JNI_ABORT);
        free(po);
        return;
    }
    int rc = dpf_query_batch_resident(ix, (const int32_t *)pq, pv, (int64_t)nq, (int32_t)steps, po);
    (*env)->ReleaseDoubleArrayElements(env, q, pv, JNI_ABORT);
    if (pq) (*env)->ReleaseIntArrayElements(env, qids, pq, JNI_ABORT);
    if (rc == DPF_OK || rc == DPF_ENOTFIT) (*env)->SetLongArrayRegion(env, offsets, 0, (jsize)(nq + 1), (const jlong *)po);
    free(po);
    throw_dpf(env, ix, rc);
}

/* setsFetch(h, q0, q1, count) -> the ids of the resident Sets of the queries [q0, q1), back to back; count = offsets(q1) -
 * offsets(q0) as queryBatchResident returned them */
JNIEXPORT jintArray JNICALL Java_mclab_gpu_Dpf_setsFetch(JNIEnv *env, jclass cls, jlong h, jlong q0, jlong q1, jlong count)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    if (!ix) { throw_iae(env, "Dpf: null handle"); return NULL; }
    if (count < 0 || count > INT_MAX) { throw_iae(env, "Dpf.setsFetch: count does not fit a Java array (fetch fewer queries at a time)"); return NULL; }
    int32_t *buf = (int32_t *)malloc((size_t)(count > 0 ? count : 1) * 4);
    if (!buf) { throw_oom(env, "Dpf.setsFetch: native buffer"); return NULL; }
    /* the library copies offsets(q1) - offsets(q0) ids: refuse a count that disagrees with its own offsets */
    int rc = dpf_sets_count(ix, (int64_t)q0, (int64_t)q1) == (int64_t)count ? dpf_sets_fetch(ix, (int64_t)q0, (int64_t)q1, buf) : DPF_EINVAL;
    if (rc != DPF_OK) {
        free(buf);
        if (dpf_sets_count(ix, (int64_t)q0, (int64_t)q1) != (int64_t)count) throw_iae(env, "Dpf.setsFetch: count is not offsets(q1) - offsets(q0) of the resident Sets");
        else throw_dpf(env, ix, rc);
        return NULL;
    }
    jintArray out = (*env)->NewIntArray(env, (jsize)count);
    if (out && count > 0) (*env)->SetIntArrayRegion(env, out, 0, (jsize)count, (const jint *)buf);
    free(buf);
    return out;
}

/* setsTopK(h, nq, k, ids[nq*k], scores[nq*k]): the rerank of DensevectorRDFInit.scala:477-505 on the resident Sets */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_setsTopK(JNIEnv *env, jclass cls, jlong h, jlong nq, jint k, jintArray ids, jdoubleArray scores)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    if (!ix) { throw_iae(env, "Dpf: null handle"); return; }
    const int64_t cells = (int64_t)nq * k;
    if (nq < 0 || k < 1 || k > 64 || cells > INT_MAX || dpf_sets_count(ix, 0, 0) < 0 || dpf_sets_queries(ix) != (int64_t)nq) {
        throw_iae(env, "Dpf.setsTopK: nq must be the resident batch's query count, k in [1, 64]");
        return;
    }
    if (!long_enough(env, ids, cells, 0, "Dpf.setsTopK: ids must hold nq * k ints") ||
        !long_enough(env, scores, cells, 0, "Dpf.setsTopK: scores must hold nq * k doubles"))
        return;
    int32_t *oi = (int32_t *)malloc((size_t)(cells > 0 ? cells : 1) * 4);
    double *os = (double *)malloc((size_t)(cells > 0 ? cells : 1) * 8);
    if (!oi || !os) { free(oi); free(os); throw_oom(env, "Dpf.setsTopK: native result buffers"); return; }
    int rc = dpf_sets_topk(ix, (int32_t)k, oi, os);
    if (rc == DPF_OK && cells > 0) {
        (*env)->SetIntArrayRegion(env, ids, 0, (jsize)cells, (const jint *)oi);
        (*env)->SetDoubleArrayRegion(env, scores, 0, (jsize)cells, os);
    }
    free(oi); free(os);
    throw_dpf(env, ix, rc);
}

/* queryTopK(h, qids[nq] or null, q[nq*dim] doubles, nq, steps, k, ids[nq*k], scores[nq*k]): ids -1 / scores NaN padded */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_queryTopK(JNIEnv *env, jclass cls, jlong h, jintArray qids, jdoubleArray q, jlong nq,
                                                    jint steps, jint k, jintArray ids, jdoubleArray scores)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    dpf_config c;
    if (!config_of(env, ix, &c) || !doubles_handle(env, &c, nq)) return;
    const int64_t cells = (int64_t)nq * k;
    if (k < 1 || k > 64 || cells > INT_MAX) { throw_iae(env, "Dpf.queryTopK: k must be in [1, 64] and nq * k fit a Java array"); return; }
    if (!long_enough(env, qids, nq, 1, "Dpf.queryTopK: qids must hold nq ints") ||
        !long_enough(env, q, (int64_t)nq * c.dim, 0, "Dpf.queryTopK: q must hold nq * dim doubles") ||
        !long_enough(env, ids, cells, 0, "Dpf.queryTopK: ids must hold nq * k ints") ||
        !long_enough(env, scores, cells, 0, "Dpf.queryTopK: scores must hold nq * k doubles"))
        return;
    int32_t *oi = (int32_t *)malloc((size_t)(cells > 0 ? cells : 1) * 4);
    double *os = (double *)malloc((size_t)(cells > 0 ? cells : 1) * 8);
    if (!oi || !os) { free(oi); free(os); throw_oom(env, "Dpf.queryTopK: native result buffers"); return; }
    jint *pq = qids ? (*env)->GetIntArrayElements(env, qids, NULL) : NULL;
    jdouble *pv = (!qids || pq) ? (*env)->GetDoubleArrayElements(env, q, NULL) : NULL;
    if (!pv) {
        if (pq) (*env)->ReleaseIntArrayElements(env, qids, pq, JNI_ABORT);
        free(oi); free(os);
        return;
    }
    int rc = dpf_query_topk(ix, (const int32_t *)pq, pv, (int64_t)nq, (int32_t)steps, (int32_t)k, oi, os);
    (*env)->ReleaseDoubleArrayElements(env, q, pv, JNI_ABORT);
    if (pq) (*env)->ReleaseIntArrayElements(env, qids, pq, JNI_ABORT);
    if (rc == DPF_OK && cells > 0) {
        (*env)->SetIntArrayRegion(env, ids, 0, (jsize)cells, (const jint *)oi);
        (*env)->SetDoubleArrayRegion(env, scores, 0, (jsize)cells, os);
    }
    free(oi); free(os);
    throw_dpf(env, ix, rc);
}

/* rerank(h, q[nq*dim], nq, candOffsets[nq+1], candIds[], k, ids[nq*k], scores[nq*k]) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_rerank(JNIEnv *env, jclass cls, jlong h, jdoubleArray q, jlong nq, jlongArray candOffsets,
                                                 jintArray candIds, jint k, jintArray ids, jdoubleArray scores)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    dpf_config c;
    if (!config_of(env, ix, &c) || !doubles_handle(env, &c, nq)) return;
    const int64_t cells = (int64_t)nq * k;
    if (k < 1 || k > 64 || cells > INT_MAX) { throw_iae(env, "Dpf.rerank: k must be in [1, 64] and nq * k fit a Java array"); return; }
    if (!long_enough(env, q, (int64_t)nq * c.dim, 0, "Dpf.rerank: q must hold nq * dim doubles") ||
        !long_enough(env, candOffsets, (int64_t)nq + 1, 0, "Dpf.rerank: candOffsets must hold nq + 1 longs") ||
        !long_enough(env, ids, cells, 0, "Dpf.rerank: ids must hold nq * k ints") ||
        !long_enough(env, scores, cells, 0, "Dpf.rerank: scores must hold nq * k doubles"))
        return;
    jlong *po = (*env)->GetLongArrayElements(env, candOffsets, NULL);
    if (!po) return;
    /* ascending offsets from 0; the id array must hold the last of them */
    int sane = po[0] == 0;
    for (jlong i = 0; i < nq && sane; i++) sane = po[i + 1] >= po[i];
    if (!sane || !long_enough(env, candIds, po[nq], po[nq] == 0, "Dpf.rerank: candIds must hold candOffsets(nq) ints")) {
        if (!sane) throw_iae(env, "Dpf.rerank: candOffsets must start at 0 and ascend");
        (*env)->ReleaseLongArrayElements(env, candOffsets, po, JNI_ABORT);
        return;
    }
    int32_t *oi = (int32_t *)malloc((size_t)(cells > 0 ? cells : 1) * 4);
    double *os = (double *)malloc((size_t)(cells > 0 ? cells : 1) * 8);
    jdouble *pv = oi && os ? (*env)->GetDoubleArrayElements(env, q, NULL) : NULL;
    jint *pc = pv && candIds ? (*env)->GetIntArrayElements(env, candIds, NULL) : NULL;
    if (!pv || (candIds && !pc)) {
        if (!oi || !os) throw_oom(env, "Dpf.rerank: native result buffers");
        if (pv) (*env)->ReleaseDoubleArrayElements(env, q, pv, JNI_ABORT);
        (*env)->ReleaseLongArrayElements(env, candOffsets, po, JNI_ABORT);
        free(oi); free(os);
        return;
    }
    int rc = dpf_rerank(ix, pv, (int64_t)nq, (const int64_t *)po, (const int32_t *)pc, (int32_t)k, oi, os);
    if (pc) (*env)->ReleaseIntArrayElements(env, candIds, pc, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, q, pv, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, candOffsets, po, JNI_ABORT);
    if (rc == DPF_OK && cells > 0) {
        (*env)->SetIntArrayRegion(env, ids, 0, (jsize)cells, (const jint *)oi);
        (*env)->SetDoubleArrayRegion(env, scores, 0, (jsize)cells, os);
    }
    free(oi); free(os);
    throw_dpf(env, ix, rc);
}

/* subindexDistribution(h, out[2^partitionBits]) */
JNIEXPORT void JNICALL Java_mclab_gpu_Dpf_subindexDistribution(JNIEnv *env, jclass cls, jlong h, jdoubleArray out)
{
    (void)cls;
    dpf_index *ix = handle_of(h);
    dpf_config c;
    if (!config_of(env, ix, &c)) return;
    const int parts = 1 << (c.partition_bits > 0 && c.partition_bits <= 5 ? c.partition_bits : 0);
    if (!long_enough(env, out, parts, 0, "Dpf.subindexDistribution: out must hold 2^partition_bits doubles")) return;
    double v[32];
    int rc = dpf_subindex_distribution(ix, v);
    if (rc == DPF_OK) (*env)->SetDoubleArrayRegion(env, out, 0, (jsize)parts, v);
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
