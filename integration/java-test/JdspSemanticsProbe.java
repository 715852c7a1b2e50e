package net.kcundercover.spectral_analyzer;

import com.fasterxml.jackson.databind.JsonNode;
import com.fasterxml.jackson.databind.ObjectMapper;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.file.Files;
import java.nio.file.Path;
import net.kcundercover.jdsp.signal.PowerSpectralDensity;
import net.kcundercover.jdsp.signal.Resampler;
import org.junit.jupiter.api.Test;

/**
 * Records what JDSP v1.3.1 -- the library behind the reference's PSD view and down-converter, whose source is not in the
 * reference tree -- returns for a few probe signals, so that the MI355X build's stated specification (window, overlap,
 * scaling, dB convention, filter) can be aligned with it.  Source only: never compiled or run where it was written
 * (no JDK there).  It asserts nothing about JDSP; it writes files.
 *
 * <p>Calls, exactly as the reference makes them:
 * {@code PowerSpectralDensity.calculatePsdWelch(double[][] data, double fs, int nfft)}
 * ({@code AnalysisDialogController.java:308-312}; {@code nfft} = 8192 or the burst length if shorter, {@code :303-307}),
 * {@code Resampler.downConvertPolyphase(re, im, freqOff, 1.0, down)} and
 * {@code new Resampler(1, down).downConvert(re, im, freqOff, 1.0)} ({@code ExtractDownConvertService.java:106-112}).
 *
 * <p>Inputs: {@code specgpu.jdspprobe} (default {@code src/test/resources/specgpu-jdsp-probe}: the files of
 * {@code integration/java-test/jdsp-probe/}).  Outputs under {@code specgpu.jdspout} (default
 * {@code build/specgpu-jdsp-observed}): {@code <name>.psd.f64} = freq row then psd row, {@code <name>.dc_fast.f64} /
 * {@code <name>.dc_conv.f64} = re row then im row, little-endian doubles, plus {@code observed.json} with the lengths.
 * Send the directory back; {@code python tools/fit_jdsp.py <directory>} names the hypothesis that reproduces it.
 */
class JdspSemanticsProbe {

    private static final Path IN = Path.of(System.getProperty("specgpu.jdspprobe", "src/test/resources/specgpu-jdsp-probe"));
    private static final Path OUT = Path.of(System.getProperty("specgpu.jdspout", "build/specgpu-jdsp-observed"));

    private static double[] read(Path p) throws Exception {
        ByteBuffer b = ByteBuffer.wrap(Files.readAllBytes(p)).order(ByteOrder.LITTLE_ENDIAN);
        double[] v = new double[b.capacity() / 8];
        for (int i = 0; i < v.length; i++) {
            v[i] = b.getDouble(8 * i);
        }
        return v;
    }

    private static void write(Path p, double[]... rows) throws Exception {
        int n = 0;
        for (double[] r : rows) {
            n += r.length;
        }
        ByteBuffer b = ByteBuffer.allocate(8 * n).order(ByteOrder.LITTLE_ENDIAN);
        for (double[] r : rows) {
            for (double x : r) {
                b.putDouble(x);
            }
        }
        Files.write(p, b.array());
    }

    @Test
    void recordWhatJdspReturns() throws Exception {
        JsonNode probe = new ObjectMapper().readTree(IN.resolve("probe.json").toFile());
        final double fs = probe.get("fs").asDouble(), freqOff = probe.get("freq_off").asDouble();
        final int down = probe.get("down").asInt();
        Files.createDirectories(OUT);
        StringBuilder json = new StringBuilder("{\n \"fs\": " + fs + ", \"down\": " + down + ", \"freq_off\": " + freqOff + ",\n \"observed\": [\n");
        boolean first = true;
        for (JsonNode s : probe.get("signals")) {
            final String name = s.get("name").asText();
            double[] re = read(IN.resolve(name + ".re.f64")), im = read(IN.resolve(name + ".im.f64"));
            int nfft = 8192;                                   // AnalysisDialogController.java:303-307
            if (re.length < nfft) {
                nfft = re.length;
            }
            double[][] freqAndPsd = PowerSpectralDensity.calculatePsdWelch(new double[][] {re, im}, fs, nfft);
            write(OUT.resolve(name + ".psd.f64"), freqAndPsd[0], freqAndPsd[1]);
            double[][] fast = Resampler.downConvertPolyphase(re, im, freqOff, 1.0, down);
            write(OUT.resolve(name + ".dc_fast.f64"), fast[0], fast[1]);
            double[][] conv = new Resampler(1, down).downConvert(re, im, freqOff, 1.0);
            write(OUT.resolve(name + ".dc_conv.f64"), conv[0], conv[1]);
            double lo = Double.POSITIVE_INFINITY, hi = Double.NEGATIVE_INFINITY, sum = 0;
            for (double v : freqAndPsd[1]) {
                lo = Math.min(lo, v);
                hi = Math.max(hi, v);
                sum += v;
            }
            System.out.println(name + ": nfft " + nfft + " -> " + freqAndPsd[1].length + " bins, freq " + freqAndPsd[0][0] + " .. "
                    + freqAndPsd[0][freqAndPsd[0].length - 1] + ", psd min " + lo + " max " + hi + " sum " + sum
                    + "; down-converter " + fast[0].length + " (fast) / " + conv[0].length + " (conventional) samples out of " + re.length);
            json.append(first ? "" : ",\n").append("  {\"name\": \"").append(name).append("\", \"nfft\": ").append(nfft)
                    .append(", \"psd_bins\": ").append(freqAndPsd[1].length).append(", \"dc_fast\": ").append(fast[0].length)
                    .append(", \"dc_conv\": ").append(conv[0].length).append("}");
            first = false;
        }
        json.append("\n ]\n}\n");
        Files.writeString(OUT.resolve("observed.json"), json.toString());
    }
}
