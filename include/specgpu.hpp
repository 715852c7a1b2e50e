// specgpu.hpp -- header-only C++ mirror of the reference service on top of the C ABI
// (include/specgpu.h).  Same method name and argument meaning as
//   double[] SpectralService.computeMagnitudes(MappedByteBuffer, int startByte, int nfft, String datatype)
// (services/SpectralService.java:33); errors become the exceptions the Java code path raises:
// SPEC_EINVAL -> std::invalid_argument (IllegalArgumentException), SPEC_ERANGE -> std::out_of_range
// (IndexOutOfBoundsException), everything else -> std::runtime_error.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "specgpu.h"

namespace specgpu {

class SpectralService {
public:
    explicit SpectralService(int device = 0, void *hip_stream = nullptr, uint32_t flags = 0) {
        check(spec_create(device, hip_stream, flags, &ctx_), nullptr);
    }
    ~SpectralService() { spec_destroy(ctx_); }
    SpectralService(const SpectralService &) = delete;
    SpectralService &operator=(const SpectralService &) = delete;

    // tuning / testing knobs and read-only state of specgpu.h's spec_set_option / spec_get_option ("multi_verify", ...)
    void setOption(const std::string &key, int64_t value) { check(spec_set_option(ctx_, key.c_str(), value), ctx_); }
    int64_t getOption(const std::string &key) const {
        int64_t v = 0;
        check(spec_get_option(ctx_, key.c_str(), &v), ctx_);
        return v;
    }

    // SpectralService.java:33-85; `big_endian` is the buffer's byte order (SigMfHelper.java:87-91)
    std::vector<double> computeMagnitudes(const void *buffer, uint64_t capacity, int64_t startByte, uint32_t nfft,
                                          const std::string &datatype, bool big_endian) const {
        std::vector<double> out(nfft);
        check(spec_compute_magnitudes(ctx_, buffer, capacity, startByte, nfft, datatype.c_str(), big_endian, out.data()), ctx_);
        return out;
    }

    // the slice loop of MainController.updateDisplay (MainController.java:980-999), host buffers
    std::vector<float> computeWaterfall(const void *buffer, uint64_t capacity, uint64_t startByte, uint32_t nfft,
                                        uint32_t hop, uint64_t nLines, const std::string &datatype,
                                        spec_window window = SPEC_WIN_RECT) const {
        std::vector<float> out(nLines * nfft);
        check(spec_waterfall(ctx_, buffer, 0, capacity, startByte, spec_dtype_from_sigmf(datatype.c_str()), nfft, hop,
                             nLines, window, SPEC_OUT_DB20_F32, -150.0, out.data(), 0), ctx_);
        return out;
    }

    // the same loop sharded over several services (one per device), host buffer in, host tile out: spec_waterfall_multi
    static std::vector<float> computeWaterfallMulti(const std::vector<const SpectralService *> &services, const void *buffer,
                                                    uint64_t capacity, uint64_t startByte, uint32_t nfft, uint32_t hop,
                                                    uint64_t nLines, const std::string &datatype,
                                                    spec_window window = SPEC_WIN_RECT) {
        if (services.empty()) throw std::invalid_argument("no services");
        std::vector<spec_ctx *> ctx;
        for (const SpectralService *s : services) ctx.push_back(s->ctx_);
        std::vector<float> out(nLines * nfft);
        const void *iq[1] = {buffer};
        check(spec_waterfall_multi(ctx.data(), (uint32_t)ctx.size(), iq, 0, capacity, startByte,
                                   spec_dtype_from_sigmf(datatype.c_str()), nfft, hop, nLines, window, SPEC_OUT_DB20_F32,
                                   -150.0, out.data(), 0, 0), ctx[0]);
        return out;
    }

    // a batch of Welch PSDs (nPsd spans psdStrideBytes apart) spread over several services: spec_welch_psd_multi;
    // returns nPsd x nfft values, freq (may be null) receives the frequency axis
    static std::vector<float> welchPsdMulti(const std::vector<const SpectralService *> &services, const void *buffer,
                                            uint64_t capacity, uint64_t startByte, uint64_t psdStrideBytes, uint32_t nPsd,
                                            const std::string &datatype, double fs, uint32_t nfft, uint32_t hop,
                                            uint32_t segments, spec_window window = SPEC_WIN_HANN,
                                            spec_psd_scaling scaling = SPEC_PSD_DENSITY, bool decibel = false,
                                            std::vector<double> *freq = nullptr) {
        if (services.empty()) throw std::invalid_argument("no services");
        std::vector<spec_ctx *> ctx;
        for (const SpectralService *s : services) ctx.push_back(s->ctx_);
        std::vector<float> out((size_t)nPsd * nfft);
        if (freq) freq->resize(nfft);
        const void *iq[1] = {buffer};
        const uint64_t sizes[1] = {capacity};
        check(spec_welch_psd_multi(ctx.data(), (uint32_t)ctx.size(), iq, 0, sizes, startByte, psdStrideBytes, nPsd,
                                   spec_dtype_from_sigmf(datatype.c_str()), nfft, hop, segments, window, scaling, fs,
                                   decibel ? 1 : 0, freq ? freq->data() : nullptr, out.data(), 0), ctx[0]);
        return out;
    }

    // PowerSpectralDensity.calculatePsdWelch(data, fs, nfft) (AnalysisDialogController.java:308-312):
    // returns {freq, psd}.  psd is in dB/Hz (10 log10(P + 1e-20)) by default: the caller adds a dB offset to it
    // (AnalysisDialogController.java:319-328), labels clicked values "dB" (:612, :626), takes the SNR as their
    // difference (:675, :757) and reports "dB/Hz" (:751).  decibel = false gives the linear density.
    std::vector<std::vector<double>> calculatePsdWelch(const double *re, const double *im, uint64_t n, double fs,
                                                       uint32_t nfft, bool decibel = true) const {
        std::vector<double> f(nfft), p(nfft);
        check(spec_welch_psd_planar_f64(ctx_, re, im, 0, n, nfft, nfft > 1 ? nfft / 2 : 1, SPEC_WIN_HANN, SPEC_PSD_DENSITY,
                                        fs, decibel ? 1 : 0, f.data(), p.data()), ctx_);
        return {f, p};
    }

    // ExtractDownConvertService.extractAndDownConvert (ExtractDownConvertService.java:54-117): {I, Q} of
    // count / down samples; freqOff in cycles per input sample, `fast` as in the reference
    std::vector<std::vector<double>> extractAndDownConvert(const void *buffer, uint64_t capacity, uint64_t startSample,
                                                           uint64_t count, const std::string &datatype, double freqOff,
                                                           uint32_t down, bool fast) const {
        if (down == 0) throw std::invalid_argument("down must be >= 1");
        std::vector<double> re(count / down), im(count / down);
        check(spec_down_convert(ctx_, buffer, 0, capacity, startSample, count, spec_dtype_from_sigmf(datatype.c_str()),
                                freqOff, down, fast ? SPEC_DC_FAST : SPEC_DC_LPF, re.data(), im.data(), 0), ctx_);
        return {re, im};
    }

    // AnalysisDialogController.updateMagnitudeChart / updateFrequencyChart (ADC:219-284)
    std::vector<double> magnitudeTrace(const double *re, const double *im, uint64_t n, double alpha) const {
        std::vector<double> out(n);
        check(spec_magnitude_trace(ctx_, re, im, 0, n, alpha, out.data(), 0), ctx_);
        return out;
    }
    std::vector<double> instFreqTrace(const double *re, const double *im, uint64_t n, double alpha, double fs,
                                      double centerFreq) const {
        std::vector<double> out(n ? n - 1 : 0);
        check(spec_inst_freq_trace(ctx_, re, im, 0, n, alpha, fs, centerFreq, out.data(), 0), ctx_);
        return out;
    }

    spec_ctx *handle() const { return ctx_; }

private:
    static void check(spec_status st, const spec_ctx *ctx) {
        if (st == SPEC_OK) return;
        const std::string msg = std::string(spec_status_string(st)) + ": " + spec_last_error(ctx);
        if (st == SPEC_EINVAL) throw std::invalid_argument(msg);
        if (st == SPEC_ERANGE) throw std::out_of_range(msg);
        throw std::runtime_error(msg);
    }
    spec_ctx *ctx_ = nullptr;
};

}  // namespace specgpu
