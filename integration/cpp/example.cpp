// Plain C++ host (no HIP headers, no torch): a tone in a ci16 buffer through the C++ mirror.
// Build: g++ -std=c++17 -Iinclude integration/cpp/example.cpp -Lspectral_analyzer_amd/lib -lspecgpu
//        -Wl,-rpath,$PWD/spectral_analyzer_amd/lib -o example
#include <cmath>
#include <cstdio>
#include <vector>

#include "specgpu.hpp"

int main() {
    const uint32_t nfft = 1024;
    const int k = 100;
    std::vector<int16_t> iq(2 * nfft);
    for (uint32_t n = 0; n < nfft; ++n) {
        iq[2 * n] = (int16_t)std::lrint(16384 * std::cos(2 * M_PI * k * n / nfft));
        iq[2 * n + 1] = (int16_t)std::lrint(16384 * std::sin(2 * M_PI * k * n / nfft));
    }
    try {
        specgpu::SpectralService svc(0);
        auto line = svc.computeMagnitudes(iq.data(), iq.size() * 2, 0, nfft, "ci16_le", false);
        uint32_t arg = 0;
        for (uint32_t i = 1; i < nfft; ++i) if (line[i] > line[arg]) arg = i;
        const double expect = 20 * std::log10(0.5 * nfft);   // amplitude 16384/32768, unnormalised FFT
        std::printf("peak bin %u (expected %u), %.4f dB (expected %.4f)\n", arg, k + nfft / 2, line[arg], expect);
        bool ok = arg == k + nfft / 2 && std::fabs(line[arg] - expect) < 1e-2;
        try { svc.computeMagnitudes(iq.data(), iq.size() * 2, 8, nfft, "ci16_le", false); ok = false; }
        catch (const std::out_of_range &) {}
        try { svc.computeMagnitudes(iq.data(), iq.size() * 2, 0, 1000, "ci16_le", false); ok = false; }
        catch (const std::invalid_argument &) {}
        svc.setOption("multi_verify", 1);                   // knobs of specgpu.h through the mirror
        ok = ok && svc.getOption("multi_verify") == 1;
        try { svc.getOption("no_such_knob"); ok = false; }
        catch (const std::invalid_argument &) {}
        std::puts(ok ? "example ok" : "example FAILED");
        return ok ? 0 : 1;
    } catch (const std::exception &e) {
        std::printf("error: %s\n", e.what());
        return 2;
    }
}
