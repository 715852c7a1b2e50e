/*
 * harness.c -- TEST INFRASTRUCTURE: calls the JNI entry points of integration/jni/specgpu_jni.c
 * (compiled against tests/jni_stub/jni.h) through a fake JNIEnv and compares what they deliver with
 * the same requests made directly through the C ABI of include/specgpu.h.
 *
 * The fake arrays behave like a copying JVM: Get<T>ArrayElements hands out a private copy,
 * Release<T>ArrayElements writes it back unless the mode is JNI_ABORT -- so a shim that released an
 * output with JNI_ABORT, or wrote through a stale pointer, is caught.
 *
 * Exit code 0: every comparison passed (needs a GPU).  Exit code 2: nativeCreate threw (no GPU): the
 * exception class and text are printed, which is what the CPU-side test checks.
 */
#include <jni.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "specgpu.h"

#define SS(name) Java_net_kcundercover_spectral_1analyzer_services_SpectralService_##name
#define EDC(name) Java_net_kcundercover_spectral_1analyzer_services_ExtractDownConvertService_##name

/* the shim's entry points (signatures as javac -h would emit them for the replacement classes) */
jlong SS(nativeCreate)(JNIEnv *, jclass, jint, jint);
void SS(nativeDestroy)(JNIEnv *, jclass, jlong);
void SS(nativeComputeMagnitudes)(JNIEnv *, jclass, jlong, jobject, jint, jint, jstring, jboolean, jdoubleArray);
void SS(nativeWaterfall)(JNIEnv *, jclass, jlong, jobject, jlong, jint, jint, jint, jlong, jint, jdouble, jfloatArray);
void SS(nativeWaterfallMulti)(JNIEnv *, jclass, jlongArray, jobject, jlong, jint, jint, jint, jlong, jint, jdouble, jfloatArray);
void SS(nativeSetOption)(JNIEnv *, jclass, jlong, jstring, jlong);
jlong SS(nativeGetOption)(JNIEnv *, jclass, jlong, jstring);
void SS(nativeWelch)(JNIEnv *, jclass, jlong, jobject, jlong, jint, jint, jint, jint, jint, jint, jdouble, jboolean,
                     jdoubleArray, jfloatArray);
void SS(nativeWelchMulti)(JNIEnv *, jclass, jlongArray, jobject, jlong, jlong, jint, jint, jint, jint, jint, jint, jint, jdouble,
                          jboolean, jdoubleArray, jfloatArray);
void SS(nativeWaterfallRender)(JNIEnv *, jclass, jlong, jobject, jlong, jint, jint, jint, jint, jint, jint, jdouble,
                               jdouble, jdouble, jint, jintArray);
void SS(nativeWelchPlanar)(JNIEnv *, jclass, jlong, jdoubleArray, jdoubleArray, jint, jint, jint, jint, jdouble,
                           jboolean, jdoubleArray, jdoubleArray);
jint SS(nativeDtype)(JNIEnv *, jclass, jstring);
jlong SS(nativeOpenRecording)(JNIEnv *, jclass, jlong, jstring, jlong);
jlong SS(nativeRecordingBytes)(JNIEnv *, jclass, jlong);
void SS(nativeCloseRecording)(JNIEnv *, jclass, jlong);
void SS(nativeWaterfallRecording)(JNIEnv *, jclass, jlong, jlong, jlong, jint, jint, jint, jlong, jint, jdouble, jfloatArray);
void SS(nativeComputeMagnitudesRecording)(JNIEnv *, jclass, jlong, jlong, jlong, jint, jstring, jboolean, jdoubleArray);
void SS(nativeTrace)(JNIEnv *, jclass, jlong, jint, jdoubleArray, jdoubleArray, jdouble, jdouble, jdouble, jdoubleArray);
jlong EDC(nativeCreate)(JNIEnv *, jclass, jint, jint);
void EDC(nativeDestroy)(JNIEnv *, jclass, jlong);
void EDC(nativeExtractAndDownConvert)(JNIEnv *, jclass, jlong, jobject, jlong, jint, jstring, jboolean, jdouble, jint,
                                      jboolean, jdoubleArray, jdoubleArray);

/* ---- fake JVM objects ---------------------------------------------------------------------- */
enum { O_CLASS = 1, O_STRING, O_ARRAY, O_BUFFER };
typedef struct {
    int kind;
    void *data;      /* array elements / string bytes / direct buffer address (NULL: not direct) */
    jsize len;       /* array length */
    size_t esz;      /* element size */
    jlong cap;       /* buffer capacity */
    const char *name;
} fake_obj;

static char thrown_class[128], thrown_msg[600];
static int n_thrown, live_copies;

static jclass f_FindClass(JNIEnv *env, const char *name) {
    (void)env;
    fake_obj *o = (fake_obj *)calloc(1, sizeof *o);
    o->kind = O_CLASS;
    o->name = name;
    return o;
}
static jint f_ThrowNew(JNIEnv *env, jclass c, const char *msg) {
    (void)env;
    snprintf(thrown_class, sizeof thrown_class, "%s", ((fake_obj *)c)->name);
    snprintf(thrown_msg, sizeof thrown_msg, "%s", msg ? msg : "");
    ++n_thrown;
    free(c);
    return 0;
}
static const char *f_GetStringUTFChars(JNIEnv *env, jstring s, jboolean *is_copy) {
    (void)env;
    if (is_copy) *is_copy = 1;
    ++live_copies;
    return strdup((const char *)((fake_obj *)s)->data);
}
static void f_ReleaseStringUTFChars(JNIEnv *env, jstring s, const char *chars) {
    (void)env; (void)s;
    --live_copies;
    free((void *)chars);
}
static jsize f_GetArrayLength(JNIEnv *env, jarray a) { (void)env; return ((fake_obj *)a)->len; }
static void *get_elems(jarray a, jboolean *is_copy) {
    fake_obj *o = (fake_obj *)a;
    if (is_copy) *is_copy = 1;
    void *p = malloc((size_t)o->len * o->esz + 1);
    memcpy(p, o->data, (size_t)o->len * o->esz);
    ++live_copies;
    return p;
}
static void release_elems(jarray a, void *elems, jint mode) {
    fake_obj *o = (fake_obj *)a;
    if (mode != JNI_ABORT) memcpy(o->data, elems, (size_t)o->len * o->esz);
    if (mode != JNI_COMMIT) { free(elems); --live_copies; }
}
static jdouble *f_GetD(JNIEnv *e, jdoubleArray a, jboolean *c) { (void)e; return (jdouble *)get_elems(a, c); }
static void f_RelD(JNIEnv *e, jdoubleArray a, jdouble *p, jint m) { (void)e; release_elems(a, p, m); }
static jfloat *f_GetF(JNIEnv *e, jfloatArray a, jboolean *c) { (void)e; return (jfloat *)get_elems(a, c); }
static void f_RelF(JNIEnv *e, jfloatArray a, jfloat *p, jint m) { (void)e; release_elems(a, p, m); }
static jint *f_GetI(JNIEnv *e, jintArray a, jboolean *c) { (void)e; return (jint *)get_elems(a, c); }
static void f_RelI(JNIEnv *e, jintArray a, jint *p, jint m) { (void)e; release_elems(a, p, m); }
static jlong *f_GetL(JNIEnv *e, jlongArray a, jboolean *c) { (void)e; return (jlong *)get_elems(a, c); }
static void f_RelL(JNIEnv *e, jlongArray a, jlong *p, jint m) { (void)e; release_elems(a, p, m); }
static void *f_BufAddr(JNIEnv *e, jobject b) { (void)e; return ((fake_obj *)b)->data; }
static jlong f_BufCap(JNIEnv *e, jobject b) { (void)e; return ((fake_obj *)b)->data ? ((fake_obj *)b)->cap : -1; }

static const struct JNINativeInterface_ table = {
    f_FindClass, f_ThrowNew, f_GetStringUTFChars, f_ReleaseStringUTFChars, f_GetArrayLength,
    f_GetD, f_RelD, f_GetF, f_RelF, f_GetI, f_RelI, f_GetL, f_RelL, f_BufAddr, f_BufCap,
};
static JNIEnv env_value = &table;
static JNIEnv *env = &env_value;

static fake_obj mk_array(void *data, jsize len, size_t esz) { fake_obj o = {O_ARRAY, data, len, esz, 0, NULL}; return o; }
static fake_obj mk_string(const char *s) { fake_obj o = {O_STRING, (void *)s, 0, 1, 0, NULL}; return o; }
static fake_obj mk_buffer(void *p, jlong cap) { fake_obj o = {O_BUFFER, p, 0, 1, cap, NULL}; return o; }

static int failures;
#define CHECK(cond, ...)                                                             \
    do {                                                                             \
        if (!(cond)) { ++failures; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
                       fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); }        \
    } while (0)
static void expect_throw(const char *cls, const char *needle, const char *what) {
    CHECK(n_thrown == 1 && strcmp(thrown_class, cls) == 0 && strstr(thrown_msg, needle) != NULL,
          "%s: expected %s(..%s..), got %d x %s(%s)", what, cls, needle, n_thrown, thrown_class, thrown_msg);
    n_thrown = 0;
}
static void expect_clean(const char *what) {
    CHECK(n_thrown == 0, "%s threw %s(%s)", what, thrown_class, thrown_msg);
    n_thrown = 0;
}

int main(void) {
    const jlong h = SS(nativeCreate)(env, NULL, 0, 0);
    if (n_thrown) {  /* no GPU: the library's text arrives as a RuntimeException */
        printf("nativeCreate threw %s: %s\n", thrown_class, thrown_msg);
        return h == 0 && strcmp(thrown_class, "java/lang/RuntimeException") == 0 ? 2 : 1;
    }
    spec_ctx *ref = NULL;  /* the same requests straight through the C ABI, on a context of their own */
    if (spec_create(0, NULL, 0, &ref) != SPEC_OK) { fprintf(stderr, "spec_create: %s\n", spec_last_error(NULL)); return 1; }

    /* a ci16_le recording (little-endian shorts from an LCG) */
    enum { SAMPLES = 70000, NFFT = 1024, LINES = 40 };
    int16_t *rec = (int16_t *)malloc(SAMPLES * 4);
    uint32_t lcg = 12345;
    for (int i = 0; i < 2 * SAMPLES; ++i) {
        lcg = lcg * 1664525u + 1013904223u;
        rec[i] = (int16_t)((int)(lcg >> 18) - 8192 + (int)(6000 * sin(0.37 * (i / 2) + (i & 1) * 1.5707963)));
    }
    fake_obj buf = mk_buffer(rec, SAMPLES * 4), dts = mk_string("ci16_le");
    CHECK(SS(nativeDtype)(env, NULL, &dts) == SPEC_DT_CI16_LE, "nativeDtype");

    /* computeMagnitudes (SS:33-85) */
    double line[NFFT], line_ref[NFFT];
    fake_obj a_line = mk_array(line, NFFT, 8);
    SS(nativeComputeMagnitudes)(env, NULL, h, &buf, 4 * 100, NFFT, &dts, 0, &a_line);
    expect_clean("computeMagnitudes");
    CHECK(spec_compute_magnitudes(ref, rec, SAMPLES * 4, 400, NFFT, "ci16_le", 0, line_ref) == SPEC_OK, "abi");
    CHECK(memcmp(line, line_ref, sizeof line) == 0, "computeMagnitudes differs from the C ABI result");
    /* its error behaviour: IllegalArgumentException (length), IndexOutOfBoundsException (range), shim checks */
    SS(nativeComputeMagnitudes)(env, NULL, h, &buf, 0, 48, &dts, 0, &a_line);
    expect_throw("java/lang/IllegalArgumentException", "power of two", "nfft = 48");
    SS(nativeComputeMagnitudes)(env, NULL, h, &buf, SAMPLES * 4 - 8, NFFT, &dts, 0, &a_line);
    expect_throw("java/lang/IndexOutOfBoundsException", "IndexOutOfBounds", "slice past the end");
    fake_obj a_short = mk_array(line, NFFT - 1, 8);
    SS(nativeComputeMagnitudes)(env, NULL, h, &buf, 0, NFFT, &dts, 0, &a_short);
    expect_throw("java/lang/IllegalArgumentException", "shorter than nfft", "short out array");
    fake_obj heap_buf = mk_buffer(NULL, 0);
    SS(nativeComputeMagnitudes)(env, NULL, h, &heap_buf, 0, NFFT, &dts, 0, &a_line);
    expect_throw("java/lang/IllegalArgumentException", "direct buffer", "heap ByteBuffer");

    /* the batched loop (MC:980-999), the last lines run past the end -> -150 */
    float *tile = (float *)calloc((size_t)LINES * NFFT, 4), *tile_ref = (float *)calloc((size_t)LINES * NFFT, 4);
    fake_obj a_tile = mk_array(tile, LINES * NFFT, 4);
    const jlong start = 4 * 30000;
    SS(nativeWaterfall)(env, NULL, h, &buf, start, SPEC_DT_CI16_LE, NFFT, NFFT, LINES, SPEC_WIN_RECT, -150.0, &a_tile);
    expect_clean("computeWaterfall");
    CHECK(spec_waterfall(ref, rec, 0, SAMPLES * 4, (uint64_t)start, SPEC_DT_CI16_LE, NFFT, NFFT, LINES, SPEC_WIN_RECT,
                         SPEC_OUT_DB20_F32, -150.0, tile_ref, 0) == SPEC_OK, "abi");
    CHECK(memcmp(tile, tile_ref, (size_t)LINES * NFFT * 4) == 0, "computeWaterfall differs from the C ABI result");
    CHECK(tile[(size_t)(LINES - 1) * NFFT + 5] == -150.0f && tile[5] != -150.0f, "EOF fill");
    fake_obj a_tile_short = mk_array(tile, LINES * NFFT - 1, 4);
    SS(nativeWaterfall)(env, NULL, h, &buf, start, SPEC_DT_CI16_LE, NFFT, NFFT, LINES, SPEC_WIN_RECT, -150.0, &a_tile_short);
    expect_throw("java/lang/IllegalArgumentException", "shorter than nLines", "short tile");

    /* the same loop sharded over three services (here: three contexts on the one GPU) -- bit for bit the single-context tile */
    {
        jlong hs[3] = {h, SS(nativeCreate)(env, NULL, 0, 0), SS(nativeCreate)(env, NULL, 0, 0)};
        expect_clean("nativeCreate x2");
        fake_obj a_hs = mk_array(hs, 3, 8);
        memset(tile, 0, (size_t)LINES * NFFT * 4);
        SS(nativeWaterfallMulti)(env, NULL, &a_hs, &buf, start, SPEC_DT_CI16_LE, NFFT, NFFT, LINES, SPEC_WIN_RECT, -150.0, &a_tile);
        expect_clean("computeWaterfallMulti");
        CHECK(memcmp(tile, tile_ref, (size_t)LINES * NFFT * 4) == 0, "computeWaterfallMulti differs from the single-context tile");
        SS(nativeWaterfallMulti)(env, NULL, &a_hs, &buf, start, SPEC_DT_CI16_LE, NFFT, NFFT, LINES, SPEC_WIN_RECT, -150.0, &a_tile_short);
        expect_throw("java/lang/IllegalArgumentException", "shorter than nLines", "short tile (multi)");
        {   /* knobs through the shim: set, read back, unknown key */
            fake_obj k_mv = mk_string("multi_verify"), k_bad = mk_string("no_such_knob");
            SS(nativeSetOption)(env, NULL, h, &k_mv, 1);
            expect_clean("setOption");
            CHECK(SS(nativeGetOption)(env, NULL, h, &k_mv) == 1, "getOption(multi_verify) after setOption");
            expect_clean("getOption");
            SS(nativeSetOption)(env, NULL, h, &k_mv, 0);
            expect_clean("setOption back");
            (void)SS(nativeGetOption)(env, NULL, h, &k_bad);
            expect_throw("java/lang/IllegalArgumentException", "unknown key", "unknown option key");
        }
        jlong twice[2] = {h, h};
        fake_obj a_twice = mk_array(twice, 2, 8);
        SS(nativeWaterfallMulti)(env, NULL, &a_twice, &buf, start, SPEC_DT_CI16_LE, NFFT, NFFT, LINES, SPEC_WIN_RECT, -150.0, &a_tile);
        expect_throw("java/lang/IllegalArgumentException", "appears twice", "same service twice");
        /* a batch of five PSDs over the same three services: what one context returns for the batch */
        {
            enum { NPSD = 5, SEG = 6, WH = 512 };
            const jlong stride = 4 * 9000;
            static float psd[NPSD * NFFT], psd_ref[NPSD * NFFT];
            double fq[NFFT], fq_ref[NFFT];
            fake_obj a_psd = mk_array(psd, NPSD * NFFT, 4), a_fq = mk_array(fq, NFFT, 8);
            SS(nativeWelchMulti)(env, NULL, &a_hs, &buf, 400, stride, NPSD, SPEC_DT_CI16_LE, NFFT, WH, SEG, SPEC_WIN_HANN,
                                 SPEC_PSD_DENSITY, 2.0e6, 1, &a_fq, &a_psd);
            expect_clean("welchPsdMulti");
            CHECK(spec_welch_psd(ref, rec, 0, SAMPLES * 4, 400, (uint64_t)stride, NPSD, SPEC_DT_CI16_LE, NFFT, WH, SEG, SPEC_WIN_HANN,
                                 SPEC_PSD_DENSITY, 2.0e6, 1, fq_ref, psd_ref, 0) == SPEC_OK, "abi");
            CHECK(memcmp(psd, psd_ref, sizeof psd) == 0 && memcmp(fq, fq_ref, sizeof fq) == 0, "welchPsdMulti differs from the single-context batch");
            fake_obj a_psd_short = mk_array(psd, NPSD * NFFT - 1, 4);
            SS(nativeWelchMulti)(env, NULL, &a_hs, &buf, 400, stride, NPSD, SPEC_DT_CI16_LE, NFFT, WH, SEG, SPEC_WIN_HANN,
                                 SPEC_PSD_DENSITY, 2.0e6, 1, &a_fq, &a_psd_short);
            expect_throw("java/lang/IllegalArgumentException", "shorter than", "short psd array (multi)");
            SS(nativeWelchMulti)(env, NULL, &a_hs, &buf, 400, stride, NPSD, SPEC_DT_CI16_LE, NFFT, WH, 4000, SPEC_WIN_HANN,
                                 SPEC_PSD_DENSITY, 2.0e6, 1, &a_fq, &a_psd);
            expect_throw("java/lang/IndexOutOfBoundsException", "needs bytes", "segments past the end (multi)");
        }
        SS(nativeDestroy)(env, NULL, hs[1]);
        SS(nativeDestroy)(env, NULL, hs[2]);
    }

    /* the same recording as a FILE with a 100-byte header, opened by path (SigMfHelper.java:69-94 replaced) */
    {
        char path[] = "/tmp/specgpu_jni_XXXXXX";
        const int fd = mkstemp(path);
        CHECK(fd >= 0, "mkstemp");
        FILE *f = fdopen(fd, "wb");
        char hdr[100];
        memset(hdr, 0x5A, sizeof hdr);
        CHECK(f && fwrite(hdr, 1, sizeof hdr, f) == sizeof hdr && fwrite(rec, 4, SAMPLES, f) == SAMPLES, "write");
        if (f) fclose(f);
        fake_obj jpath = mk_string(path);
        const jlong r = SS(nativeOpenRecording)(env, NULL, h, &jpath, 100);
        expect_clean("openRecording");
        CHECK(SS(nativeRecordingBytes)(env, NULL, r) == SAMPLES * 4, "recording length");
        memset(tile, 0, (size_t)LINES * NFFT * 4);
        SS(nativeWaterfallRecording)(env, NULL, h, r, start, SPEC_DT_CI16_LE, NFFT, NFFT, LINES, SPEC_WIN_RECT, -150.0, &a_tile);
        expect_clean("computeWaterfall(recording)");
        CHECK(memcmp(tile, tile_ref, (size_t)LINES * NFFT * 4) == 0, "computeWaterfall(recording) differs from the mapped-buffer result");
        SS(nativeComputeMagnitudesRecording)(env, NULL, h, r, 400, NFFT, &dts, 0, &a_line);
        expect_clean("computeMagnitudes(recording)");
        CHECK(memcmp(line, line_ref, sizeof line) == 0, "computeMagnitudes(recording) differs");
        SS(nativeComputeMagnitudesRecording)(env, NULL, h, r, (jlong)SAMPLES * 4 - 8, NFFT, &dts, 0, &a_line);
        expect_throw("java/lang/IndexOutOfBoundsException", "IndexOutOfBounds", "recording slice past the end");
        SS(nativeCloseRecording)(env, NULL, r);
        fake_obj nopath = mk_string("/nonexistent/specgpu.sigmf-data");
        CHECK(SS(nativeOpenRecording)(env, NULL, h, &nopath, 0) == 0, "missing file");
        expect_throw("java/lang/IllegalArgumentException", "cannot open", "missing data file");
        remove(path);
    }

    /* Welch over raw bytes + the dialog's planar call (ADC:303-313), including a non power-of-two length */
    double freq[NFFT], freq_ref[NFFT];
    float psd[NFFT], psd_ref[NFFT];
    fake_obj a_freq = mk_array(freq, NFFT, 8), a_psd = mk_array(psd, NFFT, 4);
    SS(nativeWelch)(env, NULL, h, &buf, 0, SPEC_DT_CI16_LE, NFFT, NFFT / 2, 9, SPEC_WIN_HANN, SPEC_PSD_DENSITY, 1e6, 0,
                    &a_freq, &a_psd);
    expect_clean("welchPsd");
    CHECK(spec_welch_psd(ref, rec, 0, SAMPLES * 4, 0, 0, 1, SPEC_DT_CI16_LE, NFFT, NFFT / 2, 9, SPEC_WIN_HANN,
                         SPEC_PSD_DENSITY, 1e6, 0, freq_ref, psd_ref, 0) == SPEC_OK, "abi");
    CHECK(!memcmp(freq, freq_ref, sizeof freq) && !memcmp(psd, psd_ref, sizeof psd), "welchPsd differs");

    enum { BURST = 3001 };
    double *re = (double *)malloc(BURST * 8), *im = (double *)malloc(BURST * 8);
    for (int i = 0; i < BURST; ++i) { re[i] = rec[2 * i] / 32768.0; im[i] = rec[2 * i + 1] / 32768.0; }
    fake_obj a_re = mk_array(re, BURST, 8), a_im = mk_array(im, BURST, 8);
    const int lens[2] = {1024, BURST};  /* 8192-rule: nfft = data length when the burst is short (ADC:303-307) */
    for (int c = 0; c < 2; ++c) {
        const int n = lens[c];
        double *f = (double *)calloc(n, 8), *p = (double *)calloc(n, 8), *f2 = (double *)calloc(n, 8), *p2 = (double *)calloc(n, 8);
        fake_obj af = mk_array(f, n, 8), ap = mk_array(p, n, 8);
        /* decibel = JNI_TRUE: what SpectralService.calculatePsdWelch(data, fs, nfft) passes -- the dialog reads
         * row 1 as dB (ADC:319-328, 612, 626, 675, 751) */
        SS(nativeWelchPlanar)(env, NULL, h, &a_re, &a_im, n, n / 2, SPEC_WIN_HANN, SPEC_PSD_DENSITY, 250e3, 1, &af, &ap);
        expect_clean("calculatePsdWelch");
        CHECK(spec_welch_psd_planar_f64(ref, re, im, 0, BURST, (uint32_t)n, (uint32_t)n / 2, SPEC_WIN_HANN, SPEC_PSD_DENSITY,
                                        250e3, 1, f2, p2) == SPEC_OK, "abi: %s", spec_last_error(ref));
        CHECK(!memcmp(f, f2, (size_t)n * 8) && !memcmp(p, p2, (size_t)n * 8), "calculatePsdWelch(nfft = %d) differs", n);
        /* and it IS 10 log10 of the linear row */
        CHECK(spec_welch_psd_planar_f64(ref, re, im, 0, BURST, (uint32_t)n, (uint32_t)n / 2, SPEC_WIN_HANN, SPEC_PSD_DENSITY,
                                        250e3, 0, f2, p2) == SPEC_OK, "abi: %s", spec_last_error(ref));
        double peak = 0, worst = 0;
        for (int i = 0; i < n; ++i) {
            peak = p2[i] > peak ? p2[i] : peak;
            const double d = fabs(p[i] - 10.0 * log10(p2[i] + 1e-20));
            worst = d > worst ? d : worst;
        }
        CHECK(peak > 0, "calculatePsdWelch(nfft = %d) returned nothing", n);
        CHECK(worst <= 1e-9, "calculatePsdWelch(nfft = %d): row 1 is not 10 log10(P + 1e-20) (off by %g dB)", n, worst);
        free(f); free(p); free(f2); free(p2);
    }
    double first_re = re[0];
    CHECK(first_re == rec[0] / 32768.0, "input arrays must come back untouched");

    /* one redraw as pixels (MC:962-1049 + MC:1261-1291) */
    enum { W = 48, H = 200 };
    jint *argb = (jint *)calloc(W * H, 4), *argb_ref = (jint *)calloc(W * H, 4);
    fake_obj a_px = mk_array(argb, W * H, 4);
    SS(nativeWaterfallRender)(env, NULL, h, &buf, 0, SPEC_DT_CI16_LE, NFFT, NFFT, W, SPEC_WIN_RECT, H, 1e6, -120.0, -20.0,
                              SPEC_CMAP_HEATMAP, &a_px);
    expect_clean("renderWaterfall");
    CHECK(spec_waterfall_render(ref, rec, 0, SAMPLES * 4, 0, SPEC_DT_CI16_LE, NFFT, NFFT, W, SPEC_WIN_RECT, H, 1e6, -120.0,
                                -20.0, SPEC_CMAP_HEATMAP, argb_ref, 0) == SPEC_OK, "abi");
    CHECK(memcmp(argb, argb_ref, W * H * 4) == 0, "renderWaterfall differs");

    /* traces (ADC:219-284) */
    double *tr = (double *)calloc(BURST, 8), *tr_ref = (double *)calloc(BURST, 8);
    fake_obj a_tr = mk_array(tr, BURST, 8), a_tr1 = mk_array(tr, BURST - 1, 8);
    SS(nativeTrace)(env, NULL, h, 0, &a_re, &a_im, 0.2, 0.0, 0.0, &a_tr);
    expect_clean("magnitudeTrace");
    CHECK(spec_magnitude_trace(ref, re, im, 0, BURST, 0.2, tr_ref, 0) == SPEC_OK, "abi");
    CHECK(memcmp(tr, tr_ref, BURST * 8) == 0, "magnitudeTrace differs");
    SS(nativeTrace)(env, NULL, h, 1, &a_re, &a_im, 0.2, 250e3, 1e9, &a_tr1);
    expect_clean("instFreqTrace");
    CHECK(spec_inst_freq_trace(ref, re, im, 0, BURST, 0.2, 250e3, 1e9, tr_ref, 0) == SPEC_OK, "abi");
    CHECK(memcmp(tr, tr_ref, (BURST - 1) * 8) == 0, "instFreqTrace differs");

    /* ExtractDownConvertService.extractAndDownConvert (EDC:54-117), both filters */
    const jlong he = EDC(nativeCreate)(env, NULL, 0, 0);
    expect_clean("EDC nativeCreate");
    enum { COUNT = 40000, DOWN = 8 };
    double *dr = (double *)calloc(COUNT / DOWN, 8), *di = (double *)calloc(COUNT / DOWN, 8);
    double *dr2 = (double *)calloc(COUNT / DOWN, 8), *di2 = (double *)calloc(COUNT / DOWN, 8);
    fake_obj a_dr = mk_array(dr, COUNT / DOWN, 8), a_di = mk_array(di, COUNT / DOWN, 8);
    for (int fast = 0; fast < 2; ++fast) {
        EDC(nativeExtractAndDownConvert)(env, NULL, he, &buf, 1234, COUNT, &dts, 0, 0.0625, DOWN, (jboolean)fast, &a_dr, &a_di);
        expect_clean("extractAndDownConvert");
        CHECK(spec_down_convert(ref, rec, 0, SAMPLES * 4, 1234, COUNT, SPEC_DT_CI16_LE, 0.0625, DOWN,
                                fast ? SPEC_DC_FAST : SPEC_DC_LPF, dr2, di2, 0) == SPEC_OK, "abi");
        CHECK(!memcmp(dr, dr2, COUNT / DOWN * 8) && !memcmp(di, di2, COUNT / DOWN * 8), "extractAndDownConvert(fast=%d) differs", fast);
    }
    EDC(nativeExtractAndDownConvert)(env, NULL, he, &buf, SAMPLES - 10, COUNT, &dts, 0, 0.0, DOWN, 1, &a_dr, &a_di);
    expect_throw("java/lang/IndexOutOfBoundsException", "leave", "burst past the end");
    EDC(nativeDestroy)(env, NULL, he);

    SS(nativeDestroy)(env, NULL, h);
    spec_destroy(ref);
    CHECK(live_copies == 0, "%d Get...Elements / GetStringUTFChars copies were never released", live_copies);
    if (failures) { fprintf(stderr, "%d failure(s)\n", failures); return 1; }
    printf("jni harness ok\n");
    return 0;
}
