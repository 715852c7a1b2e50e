/*
 * specgpu_jni.c -- thin JNI shim between the Java host and the C ABI of
 * include/specgpu.h.  One native method per C entry point; no arithmetic here.
 *
 * Java classes bound: net.kcundercover.spectral_analyzer.services.SpectralService and
 * ...services.ExtractDownConvertService (the '_' of "spectral_analyzer" is mangled as "_1").  The MappedByteBuffer the
 * reference hands to computeMagnitudes (SpectralService.java:33,
 * SigMfHelper.java:84) is a direct buffer, so GetDirectBufferAddress gives the
 * mapped bytes without a copy.
 *
 * Build (needs a JDK for jni.h; `python -m spectral_analyzer_amd.build --jni` compiles it when
 * JAVA_HOME is set).  The authoring container has no JDK: there the file is compiled with
 * -Wall -Werror against the JNI subset in tests/jni_stub/jni.h and its entry points are CALLED
 * through a fake JNIEnv by tests/jni_stub/harness.c (tests/test_jni_shim.py):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       integration/jni/specgpu_jni.c -Lspectral_analyzer_amd/lib -lspecgpu \
 *       -o spectral_analyzer_amd/lib/libspecgpu_jni.so
 */
#include <jni.h>
#include <stdint.h>

#include "specgpu.h"

#define JNI_FN(name) Java_net_kcundercover_spectral_1analyzer_services_SpectralService_##name

/* spec_status -> the exception the reference's own code path would raise */
static void throw_msg(JNIEnv *env, spec_status st, const char *msg) {
    const char *cls = "java/lang/RuntimeException";
    if (st == SPEC_EINVAL) cls = "java/lang/IllegalArgumentException";         /* commons-math3 MathIllegalArgumentException */
    else if (st == SPEC_ERANGE) cls = "java/lang/IndexOutOfBoundsException";  /* ByteBuffer absolute getters */
    else if (st == SPEC_ENOMEM) cls = "java/lang/OutOfMemoryError";
    else if (st == SPEC_EUNSUPPORTED) cls = "java/lang/UnsupportedOperationException";
    jclass c = (*env)->FindClass(env, cls);
    if (c) (*env)->ThrowNew(env, c, msg);
}
/* a failure reported by the library: its own text */
static void throw_status(JNIEnv *env, spec_ctx *ctx, spec_status st) { throw_msg(env, st, spec_last_error(ctx)); }
/* a failure found in the shim itself (array too short, not a direct buffer): never the library's stale text */
static void throw_shim(JNIEnv *env, const char *msg) { throw_msg(env, SPEC_EINVAL, msg); }

JNIEXPORT jlong JNICALL JNI_FN(nativeCreate)(JNIEnv *env, jclass k, jint device, jint flags) {
    (void)k;
    spec_ctx *ctx = NULL;
    spec_status st = spec_create(device, NULL, (uint32_t)flags, &ctx);
    if (st != SPEC_OK) { throw_status(env, NULL, st); return 0; }
    return (jlong)(intptr_t)ctx;
}

/* tuning / testing knobs of spec_set_option / spec_get_option (include/specgpu.h), e.g. "multi_verify" */
JNIEXPORT void JNICALL JNI_FN(nativeSetOption)(JNIEnv *env, jclass k, jlong h, jstring key, jlong value) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    if (!ctx || !key) { throw_shim(env, "setOption: closed service or null key"); return; }
    const char *s = (*env)->GetStringUTFChars(env, key, NULL);
    if (!s) return;  /* OutOfMemoryError is pending */
    const spec_status st = spec_set_option(ctx, s, (int64_t)value);
    (*env)->ReleaseStringUTFChars(env, key, s);
    if (st != SPEC_OK) throw_status(env, ctx, st);
}

JNIEXPORT jlong JNICALL JNI_FN(nativeGetOption)(JNIEnv *env, jclass k, jlong h, jstring key) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    if (!ctx || !key) { throw_shim(env, "getOption: closed service or null key"); return 0; }
    const char *s = (*env)->GetStringUTFChars(env, key, NULL);
    if (!s) return 0;
    int64_t v = 0;
    const spec_status st = spec_get_option(ctx, s, &v);
    (*env)->ReleaseStringUTFChars(env, key, s);
    if (st != SPEC_OK) { throw_status(env, ctx, st); return 0; }
    return (jlong)v;
}

JNIEXPORT void JNICALL JNI_FN(nativeDestroy)(JNIEnv *env, jclass k, jlong h) {
    (void)env; (void)k;
    spec_destroy((spec_ctx *)(intptr_t)h);
}

/* double[] computeMagnitudes(MappedByteBuffer, int startByte, int nfft, String datatype) -- SS:33 */
JNIEXPORT void JNICALL JNI_FN(nativeComputeMagnitudes)(JNIEnv *env, jclass k, jlong h, jobject buffer,
                                                        jint startByte, jint nfft, jstring datatype,
                                                        jboolean bigEndian, jdoubleArray out) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    void *base = (*env)->GetDirectBufferAddress(env, buffer);
    jlong cap = (*env)->GetDirectBufferCapacity(env, buffer);
    if (!base || cap < 0) { throw_shim(env, "computeMagnitudes: not a direct buffer"); return; }
    if (nfft < 0 || (*env)->GetArrayLength(env, out) < nfft) { throw_shim(env, "computeMagnitudes: out is shorter than nfft"); return; }
    const char *dt = (*env)->GetStringUTFChars(env, datatype, NULL);
    jdouble *o = (*env)->GetDoubleArrayElements(env, out, NULL);
    spec_status st = spec_compute_magnitudes(ctx, base, (uint64_t)cap, (int64_t)startByte, (uint32_t)nfft, dt,
                                             bigEndian ? 1 : 0, o);
    (*env)->ReleaseDoubleArrayElements(env, out, o, st == SPEC_OK ? 0 : JNI_ABORT);
    (*env)->ReleaseStringUTFChars(env, datatype, dt);
    if (st != SPEC_OK) throw_status(env, ctx, st);
}

/* batched MainController.updateDisplay loop (MC:980-999): float[nLines*nfft] */
JNIEXPORT void JNICALL JNI_FN(nativeWaterfall)(JNIEnv *env, jclass k, jlong h, jobject buffer, jlong startByte,
                                                jint dtype, jint nfft, jint hop, jlong nLines, jint window,
                                                jdouble eofFill, jfloatArray out) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    void *base = (*env)->GetDirectBufferAddress(env, buffer);
    jlong cap = (*env)->GetDirectBufferCapacity(env, buffer);
    if (!base || cap < 0) { throw_shim(env, "computeWaterfall: not a direct buffer"); return; }
    if (nfft < 0 || nLines < 0 || (jlong)(*env)->GetArrayLength(env, out) < nLines * (jlong)nfft) {
        throw_shim(env, "computeWaterfall: out is shorter than nLines * nfft");
        return;
    }
    jfloat *o = (*env)->GetFloatArrayElements(env, out, NULL);
    spec_status st = spec_waterfall(ctx, base, 0, (uint64_t)cap, (uint64_t)startByte, (spec_dtype)dtype,
                                    (uint32_t)nfft, (uint32_t)hop, (uint64_t)nLines, (spec_window)window,
                                    SPEC_OUT_DB20_F32, eofFill, o, 0);
    (*env)->ReleaseFloatArrayElements(env, out, o, st == SPEC_OK ? 0 : JNI_ABORT);
    if (st != SPEC_OK) throw_status(env, ctx, st);
}

/* the same loop sharded over several contexts, one per device: host buffer in, host tile out */
JNIEXPORT void JNICALL JNI_FN(nativeWaterfallMulti)(JNIEnv *env, jclass k, jlongArray handles, jobject buffer,
                                                     jlong startByte, jint dtype, jint nfft, jint hop, jlong nLines,
                                                     jint window, jdouble eofFill, jfloatArray out) {
    (void)k;
    const jsize n = (*env)->GetArrayLength(env, handles);
    if (n < 1 || n > 64) { throw_shim(env, "computeWaterfallMulti: 1 ... 64 services"); return; }
    void *base = (*env)->GetDirectBufferAddress(env, buffer);
    jlong cap = (*env)->GetDirectBufferCapacity(env, buffer);
    if (!base || cap < 0) { throw_shim(env, "computeWaterfallMulti: not a direct buffer"); return; }
    if (startByte < 0 || hop < 1 || nfft < 1 || nLines < 0) { throw_shim(env, "computeWaterfallMulti: negative startByte / nLines, or hop / nfft < 1"); return; }
    /* overflow-safe: nLines * nfft may not fit a jlong */
    if (nLines > (jlong)(*env)->GetArrayLength(env, out) / (jlong)nfft) {
        throw_shim(env, "computeWaterfallMulti: out is shorter than nLines * nfft");
        return;
    }
    spec_ctx *ctx[64];
    jlong *h = (*env)->GetLongArrayElements(env, handles, NULL);
    if (!h) return;  /* OutOfMemoryError is pending */
    for (jsize i = 0; i < n; ++i) ctx[i] = (spec_ctx *)(intptr_t)h[i];
    (*env)->ReleaseLongArrayElements(env, handles, h, JNI_ABORT);
    for (jsize i = 0; i < n; ++i)
        if (!ctx[i]) { throw_shim(env, "computeWaterfallMulti: a service is closed (null handle)"); return; }
    const void *iq[1] = {base};
    jfloat *o = (*env)->GetFloatArrayElements(env, out, NULL);
    if (!o) return;
    spec_status st = spec_waterfall_multi(ctx, (uint32_t)n, iq, 0, (uint64_t)cap, (uint64_t)startByte, (spec_dtype)dtype,
                                          (uint32_t)nfft, (uint32_t)hop, (uint64_t)nLines, (spec_window)window,
                                          SPEC_OUT_DB20_F32, eofFill, o, 0, 0);
    (*env)->ReleaseFloatArrayElements(env, out, o, st == SPEC_OK ? 0 : JNI_ABORT);
    if (st != SPEC_OK) throw_status(env, ctx[0], st);
}

/* PowerSpectralDensity.calculatePsdWelch call site (ADC:308-312): freq[nfft], psd[nfft] */
JNIEXPORT void JNICALL JNI_FN(nativeWelch)(JNIEnv *env, jclass k, jlong h, jobject buffer, jlong startByte,
                                            jint dtype, jint nfft, jint hop, jint nSeg, jint window, jint scaling,
                                            jdouble fs, jboolean db, jdoubleArray freq, jfloatArray psd) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    void *base = (*env)->GetDirectBufferAddress(env, buffer);
    jlong cap = (*env)->GetDirectBufferCapacity(env, buffer);
    if (!base || cap < 0) { throw_shim(env, "welchPsd: not a direct buffer"); return; }
    if (nfft < 0 || (*env)->GetArrayLength(env, freq) < nfft || (*env)->GetArrayLength(env, psd) < nfft) {
        throw_shim(env, "welchPsd: freq / psd are shorter than nfft");
        return;
    }
    jdouble *f = (*env)->GetDoubleArrayElements(env, freq, NULL);
    jfloat *p = (*env)->GetFloatArrayElements(env, psd, NULL);
    spec_status st = spec_welch_psd(ctx, base, 0, (uint64_t)cap, (uint64_t)startByte, 0, 1, (spec_dtype)dtype,
                                    (uint32_t)nfft, (uint32_t)hop, (uint32_t)nSeg, (spec_window)window,
                                    (spec_psd_scaling)scaling, fs, db ? 1 : 0, f, p, 0);
    (*env)->ReleaseFloatArrayElements(env, psd, p, st == SPEC_OK ? 0 : JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, freq, f, st == SPEC_OK ? 0 : JNI_ABORT);
    if (st != SPEC_OK) throw_status(env, ctx, st);
}

/* a batch of PSDs (nPsd spans psdStrideBytes apart) spread over several services: freq[nfft], psd[nPsd * nfft] */
JNIEXPORT void JNICALL JNI_FN(nativeWelchMulti)(JNIEnv *env, jclass k, jlongArray handles, jobject buffer, jlong startByte,
                                                 jlong psdStrideBytes, jint nPsd, jint dtype, jint nfft, jint hop, jint nSeg,
                                                 jint window, jint scaling, jdouble fs, jboolean db, jdoubleArray freq,
                                                 jfloatArray psd) {
    (void)k;
    const jsize n = (*env)->GetArrayLength(env, handles);
    if (n < 1 || n > 64) { throw_shim(env, "welchPsdMulti: 1 ... 64 services"); return; }
    void *base = (*env)->GetDirectBufferAddress(env, buffer);
    jlong cap = (*env)->GetDirectBufferCapacity(env, buffer);
    if (!base || cap < 0) { throw_shim(env, "welchPsdMulti: not a direct buffer"); return; }
    if (nfft < 1 || nPsd < 0 || startByte < 0 || psdStrideBytes < 0 || hop < 1 || nSeg < 0 || (*env)->GetArrayLength(env, freq) < nfft ||
        (jlong)nPsd > (jlong)(*env)->GetArrayLength(env, psd) / (jlong)nfft) {
        throw_shim(env, "welchPsdMulti: negative argument, or freq is shorter than nfft or psd than nPsd * nfft");
        return;
    }
    spec_ctx *ctx[64];
    jlong *h = (*env)->GetLongArrayElements(env, handles, NULL);
    if (!h) return;  /* OutOfMemoryError is pending */
    for (jsize i = 0; i < n; ++i) ctx[i] = (spec_ctx *)(intptr_t)h[i];
    (*env)->ReleaseLongArrayElements(env, handles, h, JNI_ABORT);
    for (jsize i = 0; i < n; ++i)
        if (!ctx[i]) { throw_shim(env, "welchPsdMulti: a service is closed (null handle)"); return; }
    const void *iq[1] = {base};
    const uint64_t n_bytes[1] = {(uint64_t)cap};
    jdouble *f = (*env)->GetDoubleArrayElements(env, freq, NULL);
    if (!f) return;
    jfloat *p = (*env)->GetFloatArrayElements(env, psd, NULL);
    if (!p) { (*env)->ReleaseDoubleArrayElements(env, freq, f, JNI_ABORT); return; }
    spec_status st = spec_welch_psd_multi(ctx, (uint32_t)n, iq, 0, n_bytes, (uint64_t)startByte, (uint64_t)psdStrideBytes,
                                          (uint32_t)nPsd, (spec_dtype)dtype, (uint32_t)nfft, (uint32_t)hop, (uint32_t)nSeg,
                                          (spec_window)window, (spec_psd_scaling)scaling, fs, db ? 1 : 0, f, p, 0);
    (*env)->ReleaseFloatArrayElements(env, psd, p, st == SPEC_OK ? 0 : JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, freq, f, st == SPEC_OK ? 0 : JNI_ABORT);
    if (st != SPEC_OK) throw_status(env, ctx[0], st);
}

/* one redraw: MainController.updateDisplay() slice loop + renderSpectrogram (MC:962-1049, MC:1261-1291);
 * argb receives height*width IntArgb pixels (B,G,R,A bytes = little-endian int), ready for
 * PixelWriter.setPixels(0, 0, w, h, PixelFormat.getIntArgbInstance(), argb, 0, w) */
JNIEXPORT void JNICALL JNI_FN(nativeWaterfallRender)(JNIEnv *env, jclass k, jlong h, jobject buffer, jlong startByte,
                                                      jint dtype, jint nfft, jint hop, jint width, jint window,
                                                      jint height, jdouble fs, jdouble minDb, jdouble maxDb,
                                                      jint colormap, jintArray argb) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    void *base = (*env)->GetDirectBufferAddress(env, buffer);
    jlong cap = (*env)->GetDirectBufferCapacity(env, buffer);
    if (!base || cap < 0) { throw_shim(env, "waterfallRender: not a direct buffer"); return; }
    if (width < 0 || height < 0 || (jlong)(*env)->GetArrayLength(env, argb) < (jlong)width * height) {
        throw_shim(env, "waterfallRender: argb is shorter than width * height");
        return;
    }
    jint *px = (*env)->GetIntArrayElements(env, argb, NULL);
    spec_status st = spec_waterfall_render(ctx, base, 0, (uint64_t)cap, (uint64_t)startByte, (spec_dtype)dtype,
                                           (uint32_t)nfft, (uint32_t)hop, (uint32_t)width, (spec_window)window,
                                           (uint32_t)height, fs, minDb, maxDb, (spec_colormap)colormap, px, 0);
    (*env)->ReleaseIntArrayElements(env, argb, px, st == SPEC_OK ? 0 : JNI_ABORT);
    if (st != SPEC_OK) throw_status(env, ctx, st);
}

/* PowerSpectralDensity.calculatePsdWelch(double[][] data, double fs, int nfft) -- ADC:308-312 */
JNIEXPORT void JNICALL JNI_FN(nativeWelchPlanar)(JNIEnv *env, jclass k, jlong h, jdoubleArray re, jdoubleArray im,
                                                  jint nfft, jint hop, jint window, jint scaling, jdouble fs,
                                                  jboolean db, jdoubleArray freq, jdoubleArray psd) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    jsize n = (*env)->GetArrayLength(env, re);
    if ((*env)->GetArrayLength(env, im) != n) { throw_shim(env, "calculatePsdWelch: data[0] and data[1] differ in length"); return; }
    if (nfft < 0 || (*env)->GetArrayLength(env, freq) < nfft || (*env)->GetArrayLength(env, psd) < nfft) {
        throw_shim(env, "calculatePsdWelch: freq / psd are shorter than nfft");
        return;
    }
    jdouble *r = (*env)->GetDoubleArrayElements(env, re, NULL);
    jdouble *i = (*env)->GetDoubleArrayElements(env, im, NULL);
    jdouble *f = (*env)->GetDoubleArrayElements(env, freq, NULL);
    jdouble *p = (*env)->GetDoubleArrayElements(env, psd, NULL);
    spec_status st = spec_welch_psd_planar_f64(ctx, r, i, 0, (uint64_t)n, (uint32_t)nfft, (uint32_t)hop,
                                               (spec_window)window, (spec_psd_scaling)scaling, fs, db ? 1 : 0, f, p);
    (*env)->ReleaseDoubleArrayElements(env, psd, p, st == SPEC_OK ? 0 : JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, freq, f, st == SPEC_OK ? 0 : JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, im, i, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, re, r, JNI_ABORT);
    if (st != SPEC_OK) throw_status(env, ctx, st);
}

/* ---- recordings on disk: SigMfHelper hands over path + header size instead of a <= 2 GiB mapping
 * (SigMfHelper.java:69-94); offsets are jlong end to end ------------------------------------------- */
JNIEXPORT jlong JNICALL JNI_FN(nativeOpenRecording)(JNIEnv *env, jclass k, jlong h, jstring path, jlong headerBytes) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    if (headerBytes < 0) { throw_shim(env, "openRecording: negative header size"); return 0; }
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    spec_recording *rec = NULL;
    spec_status st = spec_open_recording(ctx, p, (uint64_t)headerBytes, &rec);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (st != SPEC_OK) { throw_status(env, ctx, st); return 0; }
    return (jlong)(intptr_t)rec;
}

JNIEXPORT jlong JNICALL JNI_FN(nativeRecordingBytes)(JNIEnv *env, jclass k, jlong rec) {
    (void)env; (void)k;
    return (jlong)spec_recording_bytes((const spec_recording *)(intptr_t)rec);
}

JNIEXPORT void JNICALL JNI_FN(nativeCloseRecording)(JNIEnv *env, jclass k, jlong rec) {
    (void)env; (void)k;
    spec_close_recording((spec_recording *)(intptr_t)rec);
}

JNIEXPORT void JNICALL JNI_FN(nativeWaterfallRecording)(JNIEnv *env, jclass k, jlong h, jlong rec, jlong startByte,
                                                         jint dtype, jint nfft, jint hop, jlong nLines, jint window,
                                                         jdouble eofFill, jfloatArray out) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    if (startByte < 0 || nfft < 0 || nLines < 0 || (jlong)(*env)->GetArrayLength(env, out) < nLines * (jlong)nfft) {
        throw_shim(env, "computeWaterfall(recording): negative argument or out shorter than nLines * nfft");
        return;
    }
    jfloat *o = (*env)->GetFloatArrayElements(env, out, NULL);
    spec_status st = spec_waterfall_recording(ctx, (const spec_recording *)(intptr_t)rec, (uint64_t)startByte,
                                              (spec_dtype)dtype, (uint32_t)nfft, (uint32_t)hop, (uint64_t)nLines,
                                              (spec_window)window, SPEC_OUT_DB20_F32, eofFill, o, 0);
    (*env)->ReleaseFloatArrayElements(env, out, o, st == SPEC_OK ? 0 : JNI_ABORT);
    if (st != SPEC_OK) throw_status(env, ctx, st);
}

JNIEXPORT void JNICALL JNI_FN(nativeComputeMagnitudesRecording)(JNIEnv *env, jclass k, jlong h, jlong rec,
                                                                 jlong startByte, jint nfft, jstring datatype,
                                                                 jboolean bigEndian, jdoubleArray out) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    if (nfft < 0 || (*env)->GetArrayLength(env, out) < nfft) { throw_shim(env, "computeMagnitudes(recording): out is shorter than nfft"); return; }
    if (startByte < 0) { throw_msg(env, SPEC_ERANGE, "computeMagnitudes(recording): negative offset"); return; }
    const char *dt = (*env)->GetStringUTFChars(env, datatype, NULL);
    jdouble *o = (*env)->GetDoubleArrayElements(env, out, NULL);
    spec_status st = spec_compute_magnitudes_recording(ctx, (const spec_recording *)(intptr_t)rec, (uint64_t)startByte,
                                                       (uint32_t)nfft, dt, bigEndian ? 1 : 0, o);
    (*env)->ReleaseDoubleArrayElements(env, out, o, st == SPEC_OK ? 0 : JNI_ABORT);
    (*env)->ReleaseStringUTFChars(env, datatype, dt);
    if (st != SPEC_OK) throw_status(env, ctx, st);
}

JNIEXPORT jint JNICALL JNI_FN(nativeDtype)(JNIEnv *env, jclass k, jstring datatype) {
    (void)k;
    const char *dt = (*env)->GetStringUTFChars(env, datatype, NULL);
    jint v = (jint)spec_dtype_from_sigmf(dt);
    (*env)->ReleaseStringUTFChars(env, datatype, dt);
    return v;
}

/* AnalysisDialogController.updateMagnitudeChart / updateFrequencyChart loops (ADC:219-284):
 * which == 0: out[n] = 20 log10(EMA(hypot)); which == 1: out[n-1] = EMA(phase step in Hz) + centerFreq */
JNIEXPORT void JNICALL JNI_FN(nativeTrace)(JNIEnv *env, jclass k, jlong h, jint which, jdoubleArray re,
                                            jdoubleArray im, jdouble alpha, jdouble fs, jdouble centerFreq,
                                            jdoubleArray out) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    jsize n = (*env)->GetArrayLength(env, re);
    if ((*env)->GetArrayLength(env, im) != n || (*env)->GetArrayLength(env, out) < (which ? n - 1 : n)) {
        throw_shim(env, "trace: array lengths do not match");
        return;
    }
    jdouble *r = (*env)->GetDoubleArrayElements(env, re, NULL);
    jdouble *i = (*env)->GetDoubleArrayElements(env, im, NULL);
    jdouble *o = (*env)->GetDoubleArrayElements(env, out, NULL);
    spec_status st = which ? spec_inst_freq_trace(ctx, r, i, 0, (uint64_t)n, alpha, fs, centerFreq, o, 0)
                           : spec_magnitude_trace(ctx, r, i, 0, (uint64_t)n, alpha, o, 0);
    (*env)->ReleaseDoubleArrayElements(env, out, o, st == SPEC_OK ? 0 : JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, im, i, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, re, r, JNI_ABORT);
    if (st != SPEC_OK) throw_status(env, ctx, st);
}

/* ---- net.kcundercover.spectral_analyzer.services.ExtractDownConvertService ------------------- */
#define EDC_FN(name) Java_net_kcundercover_spectral_1analyzer_services_ExtractDownConvertService_##name

JNIEXPORT jlong JNICALL EDC_FN(nativeCreate)(JNIEnv *env, jclass k, jint device, jint flags) {
    return JNI_FN(nativeCreate)(env, k, device, flags);
}

JNIEXPORT void JNICALL EDC_FN(nativeDestroy)(JNIEnv *env, jclass k, jlong h) { JNI_FN(nativeDestroy)(env, k, h); }

/* double[][] extractAndDownConvert(MappedByteBuffer, long startSample, int count, String datatype,
 *                                  double freqOff, int down, boolean fast) -- EDC:54-117;
 * re / im receive count / down samples each */
JNIEXPORT void JNICALL EDC_FN(nativeExtractAndDownConvert)(JNIEnv *env, jclass k, jlong h, jobject buffer,
                                                            jlong startSample, jint count, jstring datatype,
                                                            jboolean bigEndian, jdouble freqOff, jint down,
                                                            jboolean fast, jdoubleArray re, jdoubleArray im) {
    (void)k;
    spec_ctx *ctx = (spec_ctx *)(intptr_t)h;
    void *base = (*env)->GetDirectBufferAddress(env, buffer);
    jlong cap = (*env)->GetDirectBufferCapacity(env, buffer);
    if (!base || cap < 0 || startSample < 0 || count < 0 || down <= 0 ||
        (*env)->GetArrayLength(env, re) < count / down || (*env)->GetArrayLength(env, im) < count / down) {
        throw_shim(env, "extractAndDownConvert: bad buffer, negative argument or output arrays shorter than count / down");
        return;
    }
    const char *s = (*env)->GetStringUTFChars(env, datatype, NULL);
    spec_dtype dt = spec_dtype_from_sigmf(s);
    (*env)->ReleaseStringUTFChars(env, datatype, s);
    /* byte order is the buffer's (SigMfHelper.java:87-91), not the string's */
    if (dt == SPEC_DT_CI16_LE || dt == SPEC_DT_CI16_BE) dt = bigEndian ? SPEC_DT_CI16_BE : SPEC_DT_CI16_LE;
    else if (dt == SPEC_DT_CF64_LE || dt == SPEC_DT_CF64_BE) dt = bigEndian ? SPEC_DT_CF64_BE : SPEC_DT_CF64_LE;
    else if (dt != SPEC_DT_CU8 && dt != SPEC_DT_CI8) dt = bigEndian ? SPEC_DT_CF32_BE : SPEC_DT_CF32_LE; /* EDC:94-96 */
    jdouble *r = (*env)->GetDoubleArrayElements(env, re, NULL);
    jdouble *i = (*env)->GetDoubleArrayElements(env, im, NULL);
    spec_status st = spec_down_convert(ctx, base, 0, (uint64_t)cap, (uint64_t)startSample, (uint64_t)count, dt,
                                       freqOff, (uint32_t)down, fast ? SPEC_DC_FAST : SPEC_DC_LPF, r, i, 0);
    (*env)->ReleaseDoubleArrayElements(env, im, i, st == SPEC_OK ? 0 : JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, re, r, st == SPEC_OK ? 0 : JNI_ABORT);
    if (st != SPEC_OK) throw_status(env, ctx, st);
}
