/*
 * jni.h -- TEST INFRASTRUCTURE: the subset of the Java Native Interface that
 * integration/jni/specgpu_jni.c uses, written from the JNI specification so that the shim can be
 * compiled (-Wall -Werror) and its entry points called in a container without a JDK.  Types and
 * function prototypes follow the specification; the function table holds only the entries the shim
 * needs, so this header is NOT layout-compatible with a real JVM -- production builds use the JDK's
 * own jni.h (spectral_analyzer_amd/build.py build_jni).  The fake JNIEnv lives in harness.c.
 */
#ifndef SPECGPU_TEST_JNI_H
#define SPECGPU_TEST_JNI_H

#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_COMMIT 1

typedef int32_t jint;
typedef int64_t jlong;
typedef uint8_t jboolean;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

typedef void *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jarray jdoubleArray;
typedef jarray jfloatArray;
typedef jarray jintArray;
typedef jarray jlongArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *env, const char *name);
    jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    const char *(*GetStringUTFChars)(JNIEnv *env, jstring str, jboolean *isCopy);
    void (*ReleaseStringUTFChars)(JNIEnv *env, jstring str, const char *chars);
    jsize (*GetArrayLength)(JNIEnv *env, jarray array);
    jdouble *(*GetDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jboolean *isCopy);
    void (*ReleaseDoubleArrayElements)(JNIEnv *env, jdoubleArray array, jdouble *elems, jint mode);
    jfloat *(*GetFloatArrayElements)(JNIEnv *env, jfloatArray array, jboolean *isCopy);
    void (*ReleaseFloatArrayElements)(JNIEnv *env, jfloatArray array, jfloat *elems, jint mode);
    jint *(*GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);
    void (*ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);
    jlong *(*GetLongArrayElements)(JNIEnv *env, jlongArray array, jboolean *isCopy);
    void (*ReleaseLongArrayElements)(JNIEnv *env, jlongArray array, jlong *elems, jint mode);
    void *(*GetDirectBufferAddress)(JNIEnv *env, jobject buf);
    jlong (*GetDirectBufferCapacity)(JNIEnv *env, jobject buf);
};

#endif
