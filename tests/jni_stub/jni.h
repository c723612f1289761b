/*
 * TEST FIXTURE, not the JDK's jni.h: just enough of the public JNI C interface (types, the members of
 * JNINativeInterface_ that jni/dpf_jni.c uses, JNIEXPORT / JNICALL / JNI_ABORT) for a -fsyntax-only
 * compile of the shim on a machine without a JDK (tests/test_jni_shim.py).  The member ORDER of the
 * function table is not the JDK's, so nothing compiled against this file may ever be loaded into a JVM;
 * a real build uses $JAVA_HOME/include/jni.h (jni/CMakeLists.txt).
 */
#ifndef DPF_TEST_JNI_STUB_H
#define DPF_TEST_JNI_STUB_H
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

typedef int32_t jint;
typedef int64_t jlong;
typedef float jfloat;
typedef double jdouble;
typedef uint8_t jboolean;
typedef jint jsize;
struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass, jstring, jarray, jintArray, jlongArray, jfloatArray, jdoubleArray, jthrowable;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *, const char *);
    jint (*ThrowNew)(JNIEnv *, jclass, const char *);
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    jint *(*GetIntArrayElements)(JNIEnv *, jintArray, jboolean *);
    jlong *(*GetLongArrayElements)(JNIEnv *, jlongArray, jboolean *);
    jfloat *(*GetFloatArrayElements)(JNIEnv *, jfloatArray, jboolean *);
    jdouble *(*GetDoubleArrayElements)(JNIEnv *, jdoubleArray, jboolean *);
    void (*ReleaseIntArrayElements)(JNIEnv *, jintArray, jint *, jint);
    void (*ReleaseLongArrayElements)(JNIEnv *, jlongArray, jlong *, jint);
    void (*ReleaseFloatArrayElements)(JNIEnv *, jfloatArray, jfloat *, jint);
    void (*ReleaseDoubleArrayElements)(JNIEnv *, jdoubleArray, jdouble *, jint);
    jintArray (*NewIntArray)(JNIEnv *, jsize);
    void (*GetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, jint *);
    void (*SetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, const jint *);
    void (*SetLongArrayRegion)(JNIEnv *, jlongArray, jsize, jsize, const jlong *);
    void (*SetDoubleArrayRegion)(JNIEnv *, jdoubleArray, jsize, jsize, const jdouble *);
    jstring (*NewStringUTF)(JNIEnv *, const char *);
    const char *(*GetStringUTFChars)(JNIEnv *, jstring, jboolean *);
    void (*ReleaseStringUTFChars)(JNIEnv *, jstring, const char *);
};
#endif
