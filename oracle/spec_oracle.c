/*
 * spec_oracle.c -- fp64 CPU restatement of the reference hot path.
 * TEST INFRASTRUCTURE ONLY; see spec_oracle.h for scope, citations and the
 * "parity unpinned" statement.  Plain C11, no dependencies beyond libm/pthread.
 */
#define _GNU_SOURCE
#include "spec_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static int starts_with(const char *s, const char *p) {
    return strncmp(s, p, strlen(p)) == 0;
}

/* Global.java:67-79 */
int so_bytes_per_sample(const char *dt) {
    if (starts_with(dt, "cf32")) return 8;
    if (starts_with(dt, "ci16")) return 4;
    if (starts_with(dt, "cu8") || starts_with(dt, "ci8")) return 2;
    if (starts_with(dt, "cf64")) return 16;
    return 8; /* Global.java:78 fallback */
}

/* SigMfHelper.java:87-91: "_le" suffix -> LITTLE_ENDIAN, anything else BIG */
int so_is_big_endian(const char *dt) {
    size_t n = strlen(dt);
    return !(n >= 3 && strcmp(dt + n - 3, "_le") == 0);
}

/* MappedByteBuffer absolute getters honouring the buffer's byte order */
static uint16_t get_u16(const uint8_t *p, int be) {
    return be ? (uint16_t)((p[0] << 8) | p[1]) : (uint16_t)((p[1] << 8) | p[0]);
}
static uint32_t get_u32(const uint8_t *p, int be) {
    return be ? ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]
              : ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0];
}
static uint64_t get_u64(const uint8_t *p, int be) {
    uint64_t hi = get_u32(be ? p : p + 4, be), lo = get_u32(be ? p + 4 : p, be);
    return (hi << 32) | lo;
}
static double get_f32(const uint8_t *p, int be) {
    uint32_t u = get_u32(p, be);
    float f;
    memcpy(&f, &u, 4);
    return (double)f;
}
static double get_f64(const uint8_t *p, int be) {
    uint64_t u = get_u64(p, be);
    double d;
    memcpy(&d, &u, 8);
    return d;
}

/* the reference resolves the datatype to four booleans once per call
 * (SS:35-38); same here, as a small enum */
enum { K_ZERO = 0, K_CI16, K_CF32, K_CU8, K_CI8, K_CF64 };
static int dtype_kind(const char *dt, int cf64_decode) {
    if (starts_with(dt, "ci16")) return K_CI16;
    if (starts_with(dt, "cf32")) return K_CF32;
    if (starts_with(dt, "cu8")) return K_CU8;
    if (starts_with(dt, "ci8")) return K_CI8;
    if (cf64_decode && starts_with(dt, "cf64")) return K_CF64;
    return K_ZERO; /* SS:60-63 (includes cf64 in the reference's own service) */
}

/* SpectralService.java:40-65; cf64 per ExtractDownConvertService.java:79-81 */
static inline void decode_kind(const uint8_t *buf, uint64_t start_byte, uint64_t i,
                               int kind, int be, double *re, double *im) {
    const uint8_t *p;
    switch (kind) {
    case K_CI16: /* SS:42-45 */
        p = buf + start_byte + i * 4;
        *re = (double)(int16_t)get_u16(p, be) / 32768.0;
        *im = (double)(int16_t)get_u16(p + 2, be) / 32768.0;
        break;
    case K_CF32: /* SS:46-49 */
        p = buf + start_byte + i * 8;
        *re = get_f32(p, be);
        *im = get_f32(p + 4, be);
        break;
    case K_CU8: /* SS:50-54 */
        p = buf + start_byte + i * 2;
        *re = ((double)p[0] - 127.5) / 128;
        *im = ((double)p[1] - 127.5) / 128;
        break;
    case K_CI8: /* SS:55-59 */
        p = buf + start_byte + i * 2;
        *re = (double)(int8_t)p[0] / 128;
        *im = (double)(int8_t)p[1] / 128;
        break;
    case K_CF64: /* EDC:79-81 */
        p = buf + start_byte + i * 16;
        *re = get_f64(p, be);
        *im = get_f64(p + 8, be);
        break;
    default: /* SS:60-63 */
        *re = 0.0;
        *im = 0.0;
    }
}

void so_decode_sample(const uint8_t *buf, uint64_t start_byte, uint64_t i,
                      const char *dt, int cf64_decode, double *re, double *im) {
    decode_kind(buf, start_byte, i, dtype_kind(dt, cf64_decode), so_is_big_endian(dt), re, im);
}

/* ---------------------------------------------------------------------------------------------
 * The transform the reference calls (SpectralService.java:23,68, build.gradle:148):
 *   new FastFourierTransformer(DftNormalization.STANDARD).transform(Complex[], TransformType.FORWARD)
 * of org.apache.commons:commons-math3:3.6.1.  The jar and its source are not in the build image; what
 * follows restates the PUBLISHED algorithm of FastFourierTransformer.transformInPlace operation by
 * operation, in the published order of the floating-point operations (the file is compiled with
 * -ffp-contract=off: Java never fuses a multiply into an add):
 *   1. n == 1: nothing; n == 2: the two-term butterfly;
 *   2. bitReversalShuffle2 on the real and imaginary arrays;
 *   3. a dedicated 4-term first stage (no multiplications);
 *   4. for n0 = 8, 16, ... n: the butterflies of every block, with the twiddle kept as a running product
 *      wSubN0ToR *= wSubN0, seeded with 1 and multiplied by the stage's root of unity W_SUB_N_R/I[log2 n0]
 *      (two hexadecimal double tables of 63 entries in the library: cos(2 pi / 2^k), -sin(2 pi / 2^k)
 *      evaluated at the double 2 pi / 2^k; regenerated by tools/gen_cm3_roots.py, which also checks the
 *      entries known from the published source).  The recurrence makes twiddle r of a stage carry an error
 *      of about r eps -- the reference's lines are LESS accurate than an exact-twiddle transform, and the
 *      oracle reproduces that (so_fft_forward_exact below is kept as the accuracy yardstick);
 *   5. STANDARD + FORWARD: no scaling.
 * Complex.abs() (SS:80) is the scaled form |a| sqrt(1 + (b/a)^2) with a the component of larger magnitude,
 * not hypot(); cm3_abs restates it including its NaN / infinity / zero branches.
 * ------------------------------------------------------------------------------------------- */
static const double CM3_W_SUB_N_R[63] = {
    0x1.0000000000000p+0, -0x1.0000000000000p+0, 0x1.1a62633145c07p-54,
    0x1.6a09e667f3bcdp-1, 0x1.d906bcf328d46p-1, 0x1.f6297cff75cb0p-1,
    0x1.fd88da3d12526p-1, 0x1.ff621e3796d7ep-1, 0x1.ffd886084cd0dp-1,
    0x1.fff62169b92dbp-1, 0x1.fffd8858e8a92p-1, 0x1.ffff621621d02p-1,
    0x1.ffffd88586ee6p-1, 0x1.fffff62161a34p-1, 0x1.fffffd8858675p-1,
    0x1.ffffff621619cp-1, 0x1.ffffffd885867p-1, 0x1.fffffff62161ap-1,
    0x1.fffffffd88586p-1, 0x1.ffffffff62162p-1, 0x1.ffffffffd8858p-1,
    0x1.fffffffff6216p-1, 0x1.fffffffffd886p-1, 0x1.ffffffffff621p-1,
    0x1.ffffffffffd88p-1, 0x1.fffffffffff62p-1, 0x1.fffffffffffd9p-1,
    0x1.ffffffffffff6p-1, 0x1.ffffffffffffep-1, 0x1.fffffffffffffp-1,
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p+0,
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p+0,
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p+0,
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p+0,
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p+0,
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p+0,
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p+0,
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p+0,
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p+0,
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p+0,
    0x1.0000000000000p+0, 0x1.0000000000000p+0, 0x1.0000000000000p+0,
};
static const double CM3_W_SUB_N_I[63] = {
    0x1.1a62633145c07p-52, -0x1.1a62633145c07p-53, -0x1.0000000000000p+0,
    -0x1.6a09e667f3bccp-1, -0x1.87de2a6aea963p-2, -0x1.8f8b83c69a60ap-3,
    -0x1.917a6bc29b42cp-4, -0x1.91f65f10dd814p-5, -0x1.92155f7a3667ep-6,
    -0x1.921d1fcdec784p-7, -0x1.921f0fe670071p-8, -0x1.921f8becca4bap-9,
    -0x1.921faaee6472dp-10, -0x1.921fb2aecb360p-11, -0x1.921fb49ee4ea6p-12,
    -0x1.921fb51aeb57bp-13, -0x1.921fb539ecf31p-14, -0x1.921fb541ad59ep-15,
    -0x1.921fb5439d73ap-16, -0x1.921fb544197a0p-17, -0x1.921fb544387bap-18,
    -0x1.921fb544403c1p-19, -0x1.921fb544422c2p-20, -0x1.921fb54442a83p-21,
    -0x1.921fb54442c73p-22, -0x1.921fb54442cefp-23, -0x1.921fb54442d0ep-24,
    -0x1.921fb54442d15p-25, -0x1.921fb54442d17p-26, -0x1.921fb54442d18p-27,
    -0x1.921fb54442d18p-28, -0x1.921fb54442d18p-29, -0x1.921fb54442d18p-30,
    -0x1.921fb54442d18p-31, -0x1.921fb54442d18p-32, -0x1.921fb54442d18p-33,
    -0x1.921fb54442d18p-34, -0x1.921fb54442d18p-35, -0x1.921fb54442d18p-36,
    -0x1.921fb54442d18p-37, -0x1.921fb54442d18p-38, -0x1.921fb54442d18p-39,
    -0x1.921fb54442d18p-40, -0x1.921fb54442d18p-41, -0x1.921fb54442d18p-42,
    -0x1.921fb54442d18p-43, -0x1.921fb54442d18p-44, -0x1.921fb54442d18p-45,
    -0x1.921fb54442d18p-46, -0x1.921fb54442d18p-47, -0x1.921fb54442d18p-48,
    -0x1.921fb54442d18p-49, -0x1.921fb54442d18p-50, -0x1.921fb54442d18p-51,
    -0x1.921fb54442d18p-52, -0x1.921fb54442d18p-53, -0x1.921fb54442d18p-54,
    -0x1.921fb54442d18p-55, -0x1.921fb54442d18p-56, -0x1.921fb54442d18p-57,
    -0x1.921fb54442d18p-58, -0x1.921fb54442d18p-59, -0x1.921fb54442d18p-60,
};

static void cm3_bit_reversal_shuffle2(double *a, double *b, uint32_t n) {
    const uint32_t half_of_n = n >> 1;
    uint32_t j = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (i < j) {
            double t = a[i]; a[i] = a[j]; a[j] = t;
            t = b[i]; b[i] = b[j]; b[j] = t;
        }
        uint32_t k = half_of_n;
        while (k <= j && k > 0) { j -= k; k >>= 1; }
        j += k;
    }
}

static void cm3_transform_in_place(double *dataR, double *dataI, uint32_t n) {
    if (n == 1) return;
    if (n == 2) {
        const double srcR0 = dataR[0], srcI0 = dataI[0], srcR1 = dataR[1], srcI1 = dataI[1];
        dataR[0] = srcR0 + srcR1; dataI[0] = srcI0 + srcI1;   /* X_0 = x_0 + x_1 */
        dataR[1] = srcR0 - srcR1; dataI[1] = srcI0 - srcI1;   /* X_1 = x_0 - x_1 */
        return;
    }
    cm3_bit_reversal_shuffle2(dataR, dataI, n);
    /* 4-term DFT, forward.  After the shuffle positions i0 .. i3 hold x_0, x_2, x_1, x_3 of the block */
    for (uint32_t i0 = 0; i0 < n; i0 += 4) {
        const uint32_t i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
        const double srcR0 = dataR[i0], srcI0 = dataI[i0];
        const double srcR1 = dataR[i2], srcI1 = dataI[i2];
        const double srcR2 = dataR[i1], srcI2 = dataI[i1];
        const double srcR3 = dataR[i3], srcI3 = dataI[i3];
        dataR[i0] = srcR0 + srcR1 + srcR2 + srcR3;            /* X_0 = x_0 + x_1 + x_2 + x_3 */
        dataI[i0] = srcI0 + srcI1 + srcI2 + srcI3;
        dataR[i1] = srcR0 - srcR2 + (srcI1 - srcI3);          /* X_1 = x_0 - x_2 + j (x_3 - x_1) */
        dataI[i1] = srcI0 - srcI2 + (srcR3 - srcR1);
        dataR[i2] = srcR0 - srcR1 + srcR2 - srcR3;            /* X_2 = x_0 - x_1 + x_2 - x_3 */
        dataI[i2] = srcI0 - srcI1 + srcI2 - srcI3;
        dataR[i3] = srcR0 - srcR2 + (srcI3 - srcI1);          /* X_3 = x_0 - x_2 + j (x_1 - x_3) */
        dataI[i3] = srcI0 - srcI2 + (srcR1 - srcR3);
    }
    uint32_t lastN0 = 4, lastLogN0 = 2;
    while (lastN0 < n) {
        const uint32_t n0 = lastN0 << 1, logN0 = lastLogN0 + 1;
        const double wSubN0R = CM3_W_SUB_N_R[logN0], wSubN0I = CM3_W_SUB_N_I[logN0];   /* FORWARD: as tabulated */
        /* combine the even / odd transforms of size lastN0 into one of size n0 */
        for (uint32_t destEvenStartIndex = 0; destEvenStartIndex < n; destEvenStartIndex += n0) {
            const uint32_t destOddStartIndex = destEvenStartIndex + lastN0;
            double wSubN0ToRR = 1, wSubN0ToRI = 0;
            for (uint32_t r = 0; r < lastN0; r++) {
                const double grR = dataR[destEvenStartIndex + r], grI = dataI[destEvenStartIndex + r];
                const double hrR = dataR[destOddStartIndex + r], hrI = dataI[destOddStartIndex + r];
                /* dest[even + r] = Gr + WsubN0ToR Hr */
                dataR[destEvenStartIndex + r] = grR + wSubN0ToRR * hrR - wSubN0ToRI * hrI;
                dataI[destEvenStartIndex + r] = grI + wSubN0ToRR * hrI + wSubN0ToRI * hrR;
                /* dest[odd + r] = Gr - WsubN0ToR Hr */
                dataR[destOddStartIndex + r] = grR - (wSubN0ToRR * hrR - wSubN0ToRI * hrI);
                dataI[destOddStartIndex + r] = grI - (wSubN0ToRR * hrI + wSubN0ToRI * hrR);
                /* WsubN0ToR *= WsubN0 */
                const double nextWsubN0ToRR = wSubN0ToRR * wSubN0R - wSubN0ToRI * wSubN0I;
                const double nextWsubN0ToRI = wSubN0ToRR * wSubN0I + wSubN0ToRI * wSubN0R;
                wSubN0ToRR = nextWsubN0ToRR;
                wSubN0ToRI = nextWsubN0ToRI;
            }
        }
        lastN0 = n0;
        lastLogN0 = logN0;
    }
    /* normalizeTransformedData(STANDARD, FORWARD): nothing */
}

/* org.apache.commons.math3.complex.Complex.abs() */
static double cm3_abs(double real, double imaginary) {
    if (isnan(real) || isnan(imaginary)) return NAN;
    if (isinf(real) || isinf(imaginary)) return INFINITY;
    if (fabs(real) < fabs(imaginary)) {
        if (imaginary == 0.0) return fabs(real);
        const double q = real / imaginary;
        return fabs(imaginary) * sqrt(1 + q * q);
    } else {
        if (real == 0.0) return fabs(imaginary);
        const double q = imaginary / real;
        return fabs(real) * sqrt(1 + q * q);
    }
}

double so_complex_abs(double re, double im) { return cm3_abs(re, im); }

int so_fft_forward_cm3(double *re, double *im, uint32_t n) {
    if (n == 0 || (n & (n - 1)) != 0) return -1;   /* the library throws MathIllegalArgumentException */
    cm3_transform_in_place(re, im, n);
    return 0;
}

/* the reference's transform */
int so_fft_forward(double *re, double *im, uint32_t n) { return so_fft_forward_cm3(re, im, n); }

/* ---- the accuracy yardstick: same radix-2 structure, every twiddle exp(-2 pi i j / len) evaluated in long
 * double and rounded to double ONCE.  Not what the reference computes; used to measure the cm3 transform's
 * own error (tests/test_oracle.py prints it per N) and by the build-defined Welch (JDSP's transform is unknown). */
static int make_twiddles(uint32_t n, double **wr_out, double **wi_out) {
    double *wr = (double *)malloc(sizeof(double) * (n / 2 + 1));
    double *wi = (double *)malloc(sizeof(double) * (n / 2 + 1));
    if (!wr || !wi) { free(wr); free(wi); return -1; }
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (uint32_t j = 0; j < n / 2; j++) {
        long double a = -two_pi * (long double)j / (long double)n;
        wr[j] = (double)cosl(a);
        wi[j] = (double)sinl(a);
    }
    *wr_out = wr;
    *wi_out = wi;
    return 0;
}

static void fft_forward_tw(double *re, double *im, uint32_t n,
                           const double *wr, const double *wi) {
    for (uint32_t i = 1, j = 0; i < n; i++) {
        uint32_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (uint32_t len = 2; len <= n; len <<= 1) {
        uint32_t half = len >> 1, step = n / len;
        for (uint32_t base = 0; base < n; base += len) {
            for (uint32_t j = 0; j < half; j++) {
                double cr = wr[j * step], ci = wi[j * step];
                uint32_t a = base + j, b = a + half;
                double tr = re[b] * cr - im[b] * ci;
                double ti = re[b] * ci + im[b] * cr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] += tr;        im[a] += ti;
            }
        }
    }
}

int so_fft_forward_exact(double *re, double *im, uint32_t n) {
    if (n == 0 || (n & (n - 1)) != 0) return -1;
    if (n == 1) return 0;
    double *wr, *wi;
    if (make_twiddles(n, &wr, &wi)) return -1;
    fft_forward_tw(re, im, n, wr, wi);
    free(wr);
    free(wi);
    return 0;
}

static void make_window(double *w, uint32_t n, int window) {
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (uint32_t i = 0; i < n; i++)
        w[i] = (window == SO_WIN_HANN)
                   ? (double)(0.5L - 0.5L * cosl(two_pi * (long double)i / (long double)n))
                   : 1.0;
}

/* one line: decode (SS:40-65) -> optional window -> FFT (SS:68); leaves the unshifted spectrum in re/im.
 * wr == NULL: the reference's transform (cm3); otherwise the exact-twiddle yardstick with that table */
static void line_spectrum(const uint8_t *buf, uint64_t start_byte, uint32_t nfft,
                          const char *dt, int cf64_decode, const double *win,
                          const double *wr, const double *wi,
                          double *re, double *im) {
    int kind = dtype_kind(dt, cf64_decode), be = so_is_big_endian(dt);
    for (uint32_t i = 0; i < nfft; i++) {
        decode_kind(buf, start_byte, i, kind, be, &re[i], &im[i]);
        if (win) { re[i] *= win[i]; im[i] *= win[i]; }
    }
    if (nfft > 1 && (nfft & (nfft - 1)) == 0) {
        if (wr) fft_forward_tw(re, im, nfft, wr, wi);
        else cm3_transform_in_place(re, im, nfft);
    }
}

/* SpectralService.java:33-85 */
int so_compute_magnitudes(const uint8_t *buf, uint64_t start_byte, uint32_t nfft,
                          const char *dt, int cf64_decode, double *out) {
    if (nfft == 0 || (nfft & (nfft - 1)) != 0) return -1;
    double *re = (double *)malloc(sizeof(double) * nfft * 2);
    if (!re) return -1;
    double *im = re + nfft;
    line_spectrum(buf, start_byte, nfft, dt, cf64_decode, NULL, NULL, NULL, re, im);   /* SS:40-68 */
    uint32_t half = nfft / 2;
    for (uint32_t i = 0; i < nfft; i++) {
        uint32_t s = (i + half) % nfft;            /* SS:78 */
        double a = cm3_abs(re[i], im[i]);          /* Complex.abs(), SS:80 */
        out[s] = 20 * log10(a + 1e-10);            /* SS:81 */
    }
    free(re);
    return 0;
}

uint64_t so_count_lines(uint64_t capacity, uint64_t start_byte, const char *dt,
                        uint32_t nfft, uint32_t hop) {
    uint64_t bps = (uint64_t)so_bytes_per_sample(dt);
    if (start_byte >= capacity || hop == 0) return 0;
    uint64_t s = (capacity - start_byte) / bps;
    if (s < nfft) return 0;
    return (s - nfft) / hop + 1;
}

/* MainController.java:980-999 (hop generalised; reference hop == nfft).
 * out_stride is the distance in doubles between consecutive output lines
 * (nfft for the public entry; 0 lets the timing driver reuse one line). */
static int waterfall_impl(const uint8_t *buf, uint64_t capacity, uint64_t start_byte,
                          const char *dt, int cf64_decode, uint32_t nfft, uint32_t hop,
                          uint64_t n_lines, int window, double eof_fill, int power_out, int fft,
                          double *out, uint64_t out_stride, double *checksum) {
    if (nfft == 0 || (nfft & (nfft - 1)) != 0 || hop == 0) return -1;
    uint64_t bps = (uint64_t)so_bytes_per_sample(dt);
    double *re = (double *)malloc(sizeof(double) * nfft * 3), *wr = NULL, *wi = NULL;
    if (!re) return -1;
    if (fft == SO_FFT_EXACT && make_twiddles(nfft, &wr, &wi)) { free(re); return -1; }
    double *im = re + nfft, *win = im + nfft;
    make_window(win, nfft, window);
    uint32_t half = nfft / 2;
    double cs = 0;
    for (uint64_t t = 0; t < n_lines; t++) {
        uint64_t byte_off = start_byte + t * (uint64_t)hop * bps;   /* MC:984-985 */
        double *o = out + t * out_stride;
        if (byte_off + (uint64_t)nfft * bps <= capacity) {           /* MC:987 */
            line_spectrum(buf, byte_off, nfft, dt, cf64_decode,
                          window == SO_WIN_RECT ? NULL : win, wr, wi, re, im);
            for (uint32_t i = 0; i < nfft; i++) {
                uint32_t s = (i + half) % nfft;                      /* SS:78 */
                if (power_out) o[s] = re[i] * re[i] + im[i] * im[i];
                else o[s] = 20 * log10((fft == SO_FFT_EXACT ? hypot(re[i], im[i]) : cm3_abs(re[i], im[i])) + 1e-10); /* SS:80-81 */
            }
        } else {
            for (uint32_t i = 0; i < nfft; i++) o[i] = eof_fill;    /* MC:994-998 */
        }
        cs += o[(t * 2654435761u) % nfft];
    }
    if (checksum) *checksum = cs;
    free(re); free(wr); free(wi);
    return 0;
}

int so_waterfall(const uint8_t *buf, uint64_t capacity, uint64_t start_byte,
                 const char *dt, int cf64_decode, uint32_t nfft, uint32_t hop,
                 uint64_t n_lines, int window, double eof_fill, int power_out,
                 double *out) {
    return waterfall_impl(buf, capacity, start_byte, dt, cf64_decode, nfft, hop,
                          n_lines, window, eof_fill, power_out, SO_FFT_CM3, out, nfft, NULL);
}

int so_waterfall_fft(const uint8_t *buf, uint64_t capacity, uint64_t start_byte,
                     const char *dt, int cf64_decode, uint32_t nfft, uint32_t hop,
                     uint64_t n_lines, int window, double eof_fill, int power_out, int fft,
                     double *out) {
    if (fft != SO_FFT_CM3 && fft != SO_FFT_EXACT) return -1;
    return waterfall_impl(buf, capacity, start_byte, dt, cf64_decode, nfft, hop,
                          n_lines, window, eof_fill, power_out, fft, out, nfft, NULL);
}

int so_welch_psd(const uint8_t *buf, uint64_t capacity, uint64_t start_byte,
                 const char *dt, int cf64_decode, uint32_t nfft, uint32_t hop,
                 uint32_t n_seg, int window, int scaling, double fs, int psd_db,
                 double *freq_out, double *psd_out) {
    if (nfft == 0 || hop == 0 || n_seg == 0) return -1;
    const int pow2 = (nfft & (nfft - 1)) == 0;
    uint64_t bps = (uint64_t)so_bytes_per_sample(dt);
    if (start_byte + ((uint64_t)(n_seg - 1) * hop + nfft) * bps > capacity) return -1;
    double *re = (double *)malloc(sizeof(double) * nfft * 6), *wr = NULL, *wi = NULL;
    if (!re) return -1;
    double *im = re + nfft, *win = im + nfft, *acc = win + nfft, *xr = acc + nfft, *xi = xr + nfft;
    if (pow2) {
        if (make_twiddles(nfft, &wr, &wi)) { free(re); return -1; }
    } else {
        /* AnalysisDialogController.java:303-307 hands JDSP the burst length itself when the burst is
         * shorter than 8192 samples: any integer.  Plain O(N^2) DFT, X[k] = sum x[n] W^(nk mod N),
         * with the full table W^m = exp(-2 pi i m / N) evaluated in long double and rounded once. */
        wr = (double *)malloc(sizeof(double) * nfft);
        wi = (double *)malloc(sizeof(double) * nfft);
        if (!wr || !wi) { free(re); free(wr); free(wi); return -1; }
        const long double two_pi = 6.283185307179586476925286766559005768L;
        for (uint32_t m = 0; m < nfft; m++) {
            long double a = -two_pi * (long double)m / (long double)nfft;
            wr[m] = (double)cosl(a);
            wi[m] = (double)sinl(a);
        }
    }
    make_window(win, nfft, window);
    double s1 = 0, s2 = 0;
    for (uint32_t i = 0; i < nfft; i++) { s1 += win[i]; s2 += win[i] * win[i]; acc[i] = 0; }
    if (!(s2 > 0)) { free(re); free(wr); free(wi); return -1; }  /* Hann of one point */
    for (uint32_t s = 0; s < n_seg; s++) {
        if (pow2) {
            line_spectrum(buf, start_byte + (uint64_t)s * hop * bps, nfft, dt,
                          cf64_decode, win, wr, wi, re, im);
        } else {
            line_spectrum(buf, start_byte + (uint64_t)s * hop * bps, nfft, dt,
                          cf64_decode, win, wr, wi, xr, xi);  /* decode + window only (see below) */
            for (uint32_t k = 0; k < nfft; k++) {
                double ar = 0, ai = 0;
                uint32_t m = 0;
                for (uint32_t n = 0; n < nfft; n++) {
                    ar += xr[n] * wr[m] - xi[n] * wi[m];
                    ai += xr[n] * wi[m] + xi[n] * wr[m];
                    m += k;
                    if (m >= nfft) m -= nfft;
                }
                re[k] = ar;
                im[k] = ai;
            }
        }
        for (uint32_t i = 0; i < nfft; i++) acc[i] += re[i] * re[i] + im[i] * im[i];
    }
    double norm = (scaling == SO_PSD_DENSITY) ? 1.0 / (fs * s2) : 1.0 / (s1 * s1);
    norm /= (double)n_seg;
    uint32_t half = nfft / 2;  /* numpy.fft.fftshift for odd lengths too: out[(k + N/2) % N] = P[k] */
    for (uint32_t i = 0; i < nfft; i++) {
        uint32_t sidx = (i + half) % nfft;
        double p = acc[i] * norm;
        psd_out[sidx] = psd_db ? 10 * log10(p + 1e-20) : p;
    }
    for (uint32_t k = 0; k < nfft; k++)
        freq_out[k] = ((double)k - (double)half) * fs / (double)nfft;
    free(re); free(wr); free(wi);
    return 0;
}

/* MainController.java:1273-1274 */
double so_display_conversion(double fs, uint32_t nfft) {
    return 10 * log10(fs / nfft) + 20 * log10((double)nfft);
}

/* ---------------- renderSpectrogram (MC:1261-1291, MC:926-957) ---------------- */
static uint8_t chan8(float c) { return (uint8_t)floor((double)c * 255.0 + 0.5); } /* Math.round */

void so_render_spectrogram(const double *waterfall, uint32_t width, uint32_t nfft, uint32_t height,
                           double fs, double min_db, double max_db, int colormap, uint8_t *out) {
    const double conversion = so_display_conversion(fs, nfft);                 /* MC:1273-1274 */
    for (uint32_t t = 0; t < width; t++) {
        for (uint32_t f = 0; f < height; f++) {
            const int bin = (int)((double)f / height * nfft);                  /* MC:1280 */
            const double db = waterfall[(size_t)t * nfft + bin] - conversion;  /* MC:1283 */
            double n = (db - min_db) / (max_db - min_db);                      /* MC:929 */
            n = n < 0.0 ? 0.0 : (n > 1.0 ? 1.0 : n);                           /* MC:930 */
            float r, g, b;
            if (colormap == 1) {                                               /* MC:943-953 Heatmap */
                if (n < 0.2) { r = g = b = 0.0f; }
                else if (n < 0.5) {                                            /* BLUE -> RED */
                    const double tt = (n - 0.2) / 0.3;
                    if (tt <= 0.0) { r = 0; g = 0; b = 1; } else if (tt >= 1.0) { r = 1; g = 0; b = 0; }
                    else { const float ft = (float)tt; r = 0.0f + (1.0f - 0.0f) * ft; g = 0.0f; b = 1.0f + (0.0f - 1.0f) * ft; }
                } else {                                                       /* RED -> YELLOW */
                    const double tt = (n - 0.5) / 0.5;
                    if (tt <= 0.0) { r = 1; g = 0; b = 0; } else if (tt >= 1.0) { r = 1; g = 1; b = 0; }
                    else { const float ft = (float)tt; r = 1.0f; g = 0.0f + (1.0f - 0.0f) * ft; b = 0.0f; }
                }
            } else {                                                           /* MC:939-941 Grayscale */
                if (n <= 0.0) r = 0.0f; else if (n >= 1.0) r = 1.0f; else r = 0.0f + (1.0f - 0.0f) * (float)n;
                g = b = r;
            }
            uint8_t *px = out + ((size_t)(height - 1 - f) * width + t) * 4;    /* MC:1288 */
            px[0] = chan8(b); px[1] = chan8(g); px[2] = chan8(r); px[3] = 255;
        }
    }
}

/* ---------------- synthetic IQ ---------------- */
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
/* tone frequencies as exact 32-bit phase increments: 0.123 and -0.31 cycles/sample */
#define SO_TONE1_INC 528280977u  /* round(0.123 * 2^32) */
#define SO_TONE2_INC 2963527434u /* round((1-0.31) * 2^32) */

static void synth_sample(uint64_t seed, uint64_t n, double *re, double *im) {
    const double two_pi = 6.283185307179586476925286766559;
    uint64_t a = splitmix64(seed ^ (2 * n)), b = splitmix64(seed ^ (2 * n + 1));
    double u1 = ((double)(a >> 40) + 0.5) * (1.0 / 16777216.0);
    double u2 = ((double)(b >> 40) + 0.5) * (1.0 / 16777216.0);
    double r = 0.05 * sqrt(-2.0 * log(u1));
    uint32_t p1 = (uint32_t)(n * (uint64_t)SO_TONE1_INC);
    uint32_t p2 = (uint32_t)(n * (uint64_t)SO_TONE2_INC);
    double a1 = two_pi * (double)p1 * (1.0 / 4294967296.0);
    double a2 = two_pi * (double)p2 * (1.0 / 4294967296.0);
    *re = r * cos(two_pi * u2) + 0.5 * cos(a1) + 0.1 * cos(a2);
    *im = r * sin(two_pi * u2) + 0.5 * sin(a1) + 0.1 * sin(a2);
}
static double clip1(double x) { return x < -1.0 ? -1.0 : (x > 1.0 ? 1.0 : x); }
static void put_u16(uint8_t *p, uint16_t v, int be) {
    if (be) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
    else { p[1] = (uint8_t)(v >> 8); p[0] = (uint8_t)v; }
}
static void put_u32(uint8_t *p, uint32_t v, int be) {
    for (int k = 0; k < 4; k++) p[be ? 3 - k : k] = (uint8_t)(v >> (8 * k));
}
static void put_u64(uint8_t *p, uint64_t v, int be) {
    for (int k = 0; k < 8; k++) p[be ? 7 - k : k] = (uint8_t)(v >> (8 * k));
}

int so_synth_iq(uint8_t *out, const char *dt, uint64_t seed,
                uint64_t first_sample, uint64_t n_samples) {
    int be = so_is_big_endian(dt);
    for (uint64_t i = 0; i < n_samples; i++) {
        double v[2];
        synth_sample(seed, first_sample + i, &v[0], &v[1]);
        for (int c = 0; c < 2; c++) {
            if (starts_with(dt, "cf32")) {
                float f = (float)v[c]; uint32_t u; memcpy(&u, &f, 4);
                put_u32(out + i * 8 + c * 4, u, be);
            } else if (starts_with(dt, "cf64")) {
                uint64_t u; memcpy(&u, &v[c], 8);
                put_u64(out + i * 16 + c * 8, u, be);
            } else if (starts_with(dt, "ci16")) {
                int16_t q = (int16_t)lrint(32767.0 * clip1(v[c]));
                put_u16(out + i * 4 + c * 2, (uint16_t)q, be);
            } else if (starts_with(dt, "ci8")) {
                out[i * 2 + c] = (uint8_t)(int8_t)lrint(127.0 * clip1(v[c]));
            } else if (starts_with(dt, "cu8")) {
                out[i * 2 + c] = (uint8_t)lrint(127.5 + 127.0 * clip1(v[c]));
            } else {
                return -1;
            }
        }
    }
    return 0;
}

/* ==========================================================================
 * the burst-analysis chain behind the Analysis dialog
 * ========================================================================== */

/* ExtractDownConvertService.java:60-97, the reader: `count` samples from sample
 * start_sample into planar doubles.  Differences from SS:40-65, restated as they are:
 * any datatype that is not ci16/cu8/ci8/cf64 reads as float pairs (EDC:94-96, no
 * zero branch), and the reference strides cf64 by 8 bytes (EDC:60-67 has no cf64
 * case: sample i's Q is sample i+1's I) -- ref_cf64_stride8 != 0 reproduces that,
 * 0 uses the 16 bytes of Global.java:67-79.  Offsets are 64-bit here (the
 * reference casts to int, EDC:80-96).  -1: span outside the buffer. */
int so_extract_iq(const uint8_t *buf, uint64_t capacity, uint64_t start_sample, uint64_t count,
                  const char *dt, int ref_cf64_stride8, double *re, double *im) {
    const int be = so_is_big_endian(dt);
    int kind = K_CF32, bps = 8, width = 8;
    if (starts_with(dt, "ci16")) { kind = K_CI16; bps = 4; width = 4; }
    else if (starts_with(dt, "cu8")) { kind = K_CU8; bps = 2; width = 2; }
    else if (starts_with(dt, "ci8")) { kind = K_CI8; bps = 2; width = 2; }
    else if (starts_with(dt, "cf64")) { kind = K_CF64; bps = ref_cf64_stride8 ? 8 : 16; width = 16; }
    if (count == 0) return 0;
    const uint64_t start_byte = start_sample * (uint64_t)bps;
    if (start_byte + (count - 1) * (uint64_t)bps + (uint64_t)width > capacity) return -1;
    for (uint64_t i = 0; i < count; ++i) {
        const uint8_t *p = buf + start_byte + i * (uint64_t)bps;
        switch (kind) {
        case K_CF64: re[i] = get_f64(p, be); im[i] = get_f64(p + 8, be); break;           /* EDC:79-81 */
        case K_CI16: re[i] = (double)(int16_t)get_u16(p, be) / 32768.0;                   /* EDC:83-85 */
                     im[i] = (double)(int16_t)get_u16(p + 2, be) / 32768.0; break;
        case K_CU8:  re[i] = ((double)p[0] - 127.5) / 128; im[i] = ((double)p[1] - 127.5) / 128; break; /* EDC:86-90 */
        case K_CI8:  re[i] = (double)(int8_t)p[0] / 128.0; im[i] = (double)(int8_t)p[1] / 128.0; break; /* EDC:91-93 */
        default:     re[i] = get_f32(p, be); im[i] = get_f32(p + 4, be);                  /* EDC:94-96 */
        }
    }
    return 0;
}

/* Down-converter taps of the build's own specification (JDSP v1.3.1 Resampler is
 * absent from the reference tree: parity unpinned).  mode 0 ("fast", EDC:104-107: "moving
 * average filter prior to decimation"): K = down boxcar taps 1/down, alignment c = down-1.
 * mode 1 ("conventional", EDC:108-113: low-pass then decimate): K = 8 down + 1 Hamming-
 * windowed sinc taps, cut-off 0.5/down cycles per sample, unit DC gain, c = 4 down. */
uint32_t so_down_convert_taps(uint32_t down, int mode, double *h /* may be NULL */, uint32_t *centre) {
    const uint32_t K = mode == 0 ? down : 8 * down + 1;
    if (centre) *centre = mode == 0 ? down - 1 : 4 * down;
    if (!h) return K;
    if (mode == 0) {
        for (uint32_t k = 0; k < K; ++k) h[k] = 1.0 / (double)down;
        return K;
    }
    const double c = 4.0 * down;
    double sum = 0.0;
    for (uint32_t k = 0; k < K; ++k) {
        const double x = ((double)k - c) / (double)down;
        const double sinc = x == 0.0 ? 1.0 : sin(M_PI * x) / (M_PI * x);
        h[k] = sinc * (0.54 - 0.46 * cos(2.0 * M_PI * (double)k / (double)(K - 1)));
        sum += h[k];
    }
    for (uint32_t k = 0; k < K; ++k) h[k] /= sum;
    return K;
}

uint64_t so_down_convert_len(uint64_t n, uint32_t down) { return down ? n / down : 0; }

/* xm[n] = x[n] exp(-2 pi i frac(freq_off n))  (freq_off in cycles per input sample: the
 * reference passes a sample rate of 1.0, EDC:106,112);  y[m] = sum_k h[k] xm[m down + c - k],
 * samples outside [0, n) are zero;  m < floor(n / down). */
int so_down_convert(const double *re, const double *im, uint64_t n, double freq_off, uint32_t down,
                    int mode, double *ore, double *oim) {
    if (down == 0 || (mode != 0 && mode != 1)) return -1;
    uint32_t c;
    const uint32_t K = so_down_convert_taps(down, mode, NULL, &c);
    double *h = (double *)malloc(sizeof(double) * K);
    double *mr = (double *)malloc(sizeof(double) * (n ? n : 1)), *mi = (double *)malloc(sizeof(double) * (n ? n : 1));
    if (!h || !mr || !mi) { free(h); free(mr); free(mi); return -1; }
    so_down_convert_taps(down, mode, h, &c);
    for (uint64_t i = 0; i < n; ++i) {
        double t = freq_off * (double)i;
        t -= floor(t);
        const double a = 2.0 * M_PI * t, cs = cos(a), sn = sin(a);
        mr[i] = re[i] * cs + im[i] * sn;  /* (re + i im)(cs - i sn) */
        mi[i] = im[i] * cs - re[i] * sn;
    }
    const uint64_t n_out = n / down;
    for (uint64_t m = 0; m < n_out; ++m) {
        double ar = 0.0, ai = 0.0;
        for (uint32_t k = 0; k < K; ++k) {
            const int64_t idx = (int64_t)(m * down) + (int64_t)c - (int64_t)k;
            if (idx < 0 || (uint64_t)idx >= n) continue;
            ar += h[k] * mr[idx];
            ai += h[k] * mi[idx];
        }
        ore[m] = ar;
        oim[m] = ai;
    }
    free(h); free(mr); free(mi);
    return 0;
}

/* AnalysisDialogController.java:219-246 updateMagnitudeChart: hypot, exponential moving
 * average seeded with the first value, 20 log10.  Every point is returned; the reference
 * plots only the finite ones (ADC:239-242). */
void so_magnitude_trace(const double *re, const double *im, uint64_t n, double alpha, double *db) {
    double v = 0.0;
    for (uint64_t i = 0; i < n; ++i) {
        const double a = hypot(re[i], im[i]);
        v = i == 0 ? a : alpha * a + (1 - alpha) * v;
        db[i] = 20 * log10(v);
    }
}

/* AnalysisDialogController.java:256-284 updateFrequencyChart: phase difference of
 * neighbouring samples wrapped to [-pi, pi], scaled to Hz, EMA seeded at i == 1, plus the
 * centre frequency.  out has n - 1 values (i = 1 .. n-1). */
void so_inst_freq_trace(const double *re, const double *im, uint64_t n, double alpha, double fs,
                        double center_freq, double *out) {
    double v = 0.0;
    for (uint64_t i = 1; i < n; ++i) {
        double d = atan2(im[i], re[i]) - atan2(im[i - 1], re[i - 1]);
        if (d > M_PI) d -= 2 * M_PI;
        else if (d < -M_PI) d += 2 * M_PI;
        const double f = (d / (2 * M_PI)) * fs;
        v = i == 1 ? f : (alpha * f) + (1 - alpha) * v;
        out[i - 1] = v + center_freq;
    }
}


/* ---------------- timed driver for the CPU baseline ---------------- */
typedef struct {
    const uint8_t *buf; uint64_t capacity; const char *dt;
    uint32_t nfft, hop; uint64_t l0, l1; int window; double checksum;
} job_t;

static void *job_main(void *p) {
    job_t *j = (job_t *)p;
    uint64_t bps = (uint64_t)so_bytes_per_sample(j->dt);
    double *line = (double *)malloc(sizeof(double) * j->nfft);
    /* the reference's own transform (commons-math3: twiddles by recurrence, no tables), SS:23,68 */
    waterfall_impl(j->buf, j->capacity, j->l0 * (uint64_t)j->hop * bps, j->dt, 1,
                   j->nfft, j->hop, j->l1 - j->l0, j->window, -150.0, 0, SO_FFT_CM3, line, 0,
                   &j->checksum);
    free(line);
    return NULL;
}

double so_time_waterfall(const uint8_t *buf, uint64_t capacity, const char *dt,
                         uint32_t nfft, uint32_t hop, uint64_t n_lines, int window,
                         int threads, double *out_checksum) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    job_t jobs[256];
    pthread_t th[256];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < threads; i++) {
        jobs[i] = (job_t){buf, capacity, dt, nfft, hop,
                          n_lines * (uint64_t)i / threads,
                          n_lines * (uint64_t)(i + 1) / threads, window, 0.0};
        pthread_create(&th[i], NULL, job_main, &jobs[i]);
    }
    double cs = 0;
    for (int i = 0; i < threads; i++) { pthread_join(th[i], NULL); cs += jobs[i].checksum; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (out_checksum) *out_checksum = cs;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
