package net.kcundercover.spectral_analyzer;

import static org.junit.jupiter.api.Assertions.assertEquals;
import static org.junit.jupiter.api.Assertions.assertTrue;

import com.fasterxml.jackson.databind.JsonNode;
import com.fasterxml.jackson.databind.ObjectMapper;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.MappedByteBuffer;
import java.nio.file.Files;
import java.nio.file.Path;
import java.util.ArrayList;
import java.util.List;
import java.util.stream.Stream;
import net.kcundercover.spectral_analyzer.services.SpectralService;
import net.kcundercover.spectral_analyzer.sigmf.SigMfHelper;
import org.junit.jupiter.api.DynamicTest;
import org.junit.jupiter.api.TestFactory;

/**
 * Pins the parity claim of the MI355X build against an EXECUTION of the reference (INTEGRATION.md, "Pinning
 * parity").  Source only: the container this repository was written in has no JDK; nothing here was compiled or run
 * there.
 *
 * <p>Every fixture under {@code specgpu.fixtures} (default {@code src/test/resources/specgpu-fixtures}; the files of
 * {@code integration/java-test/fixtures/}) is a SigMF pair plus the expected lines, little-endian float64
 * {@code [lines][nfft]}, written by the build's CPU oracle ({@code oracle/spec_oracle.c}: commons-math3 3.6.1's
 * transform and {@code Complex.abs()} restated operation by operation).  The test
 *
 * <ol>
 *   <li>loads each pair with whatever {@code sigmf.SigMfHelper.load} is on the class path (header skip, data-file
 *       resolution and byte order are the loader's: {@code SigMfHelper.java:43-94}),</li>
 *   <li>calls whatever {@code services.SpectralService.computeMagnitudes} is on the class path line by line, line
 *       {@code l} at byte {@code l * hop * bytesPerSample} ({@code MainController.java:984-985} with the fixture's hop),
 *       </li>
 *   <li>and compares with the expectation.</li>
 * </ol>
 *
 * <p>Run it twice:
 * <ul>
 *   <li><b>in the UNMODIFIED reference tree</b> ({@code ./gradlew test --tests '*SpectralServiceParityTest'}): the
 *       tolerance is {@code tolerance_ulp} (4) units in the last place of the expected double -- the oracle claims
 *       operation-for-operation identity up to {@code Math.log10}.  Green here turns "parity unpinned" into "the
 *       oracle IS the reference"; red shows the first differing bin.  cf64 fixtures expect a flat -200.0 from the
 *       unmodified reference ({@code SpectralService.java:35-63} has no cf64 branch); the windowed fixture is skipped
 *       (the reference applies no window).</li>
 *   <li><b>with the drop-in classes of {@code integration/java/} installed</b> and
 *       {@code -Dspecgpu.dropin=true -Djava.library.path=...}: the same lines through JNI and the GPU, held to the
 *       fp64 contract of {@code include/specgpu.h} (every bin {@code >= 1e-9 M}:
 *       {@code ||X|_gpu - |X|_ref| <= max(8e-15 log2 N + 5e-14, 4e-17 N) M}; every bin {@code >= 1e-5 M}:
 *       {@code |dB_gpu - dB_ref| <= max(1e-9, 3e-12 N)}); cf64 fixtures are then compared with
 *       their decoded lines ({@code ExtractDownConvertService.java:79-81}).</li>
 * </ul>
 */
class SpectralServiceParityTest {

    private static final Path FIXTURES =
            Path.of(System.getProperty("specgpu.fixtures", "src/test/resources/specgpu-fixtures"));
    private static final boolean DROP_IN = Boolean.getBoolean("specgpu.dropin");

    @TestFactory
    Stream<DynamicTest> linesMatchTheCommittedExpectations() throws Exception {
        JsonNode manifest = new ObjectMapper().readTree(FIXTURES.resolve("manifest.json").toFile());
        final int ulps = manifest.get("tolerance_ulp").asInt();
        List<DynamicTest> tests = new ArrayList<>();
        for (JsonNode f : manifest.get("fixtures")) {
            final String name = f.get("name").asText();
            final String mode = f.get("reference").asText();
            if (mode.equals("no-window")) {
                continue;   // computeMagnitudes has no window argument (neither class): covered by the C-ABI tests
            }
            tests.add(DynamicTest.dynamicTest(name, () -> check(f, name, mode, ulps)));
        }
        assertTrue(tests.size() >= 20, "fixtures missing under " + FIXTURES.toAbsolutePath());
        return tests.stream();
    }

    private static void check(JsonNode f, String name, String mode, int ulps) throws Exception {
        final int nfft = f.get("nfft").asInt(), hop = f.get("hop").asInt(), lines = f.get("lines").asInt();
        SigMfHelper helper = new SigMfHelper();
        helper.load(FIXTURES.resolve(name + ".sigmf-meta"));
        MappedByteBuffer buffer = helper.getDataBuffer();
        final String datatype = helper.getMetadata().global().datatype();
        final int bps = helper.getMetadata().global().getBytesPerSample();
        assertEquals(f.get("datatype").asText(), datatype);
        assertEquals(f.get("bytes_per_sample").asInt(), bps);
        // the loader skipped the header and resolved core:dataset: the buffer holds exactly the samples
        assertEquals((long) ((lines - 1) * (long) hop + nfft) * bps, buffer.capacity(), name + ": mapped bytes");

        ByteBuffer exp = ByteBuffer.wrap(Files.readAllBytes(FIXTURES.resolve(name + ".expected.f64")))
                .order(ByteOrder.LITTLE_ENDIAN);
        assertEquals((long) lines * nfft * 8, exp.capacity());

        SpectralService service = new SpectralService();
        double worstUlps = 0, worstLin = 0, worstDb = 0;
        for (int l = 0; l < lines; l++) {
            double[] got = service.computeMagnitudes(buffer, l * hop * bps, nfft, datatype);
            assertEquals(nfft, got.length);
            double peak = 0;
            for (int i = 0; i < nfft; i++) {
                peak = Math.max(peak, Math.pow(10.0, exp.getDouble(8 * (l * nfft + i)) / 20.0));
            }
            for (int i = 0; i < nfft; i++) {
                final double want = exp.getDouble(8 * (l * nfft + i));
                if (!DROP_IN && mode.equals("flat-200")) {
                    assertEquals(-200.0, got[i], 0.0, name + ": the reference has no cf64 branch (SpectralService.java:35-63)");
                    continue;
                }
                if (!DROP_IN) {
                    final double u = Math.abs(got[i] - want) / Math.ulp(want);
                    worstUlps = Math.max(worstUlps, u);
                    assertTrue(u <= ulps, name + " line " + l + " bin " + i + ": " + got[i] + " vs " + want + " (" + u + " ulp)");
                } else {
                    final double mg = Math.pow(10.0, got[i] / 20.0), mw = Math.pow(10.0, want / 20.0);
                    if (mw < 1e-9 * peak) {
                        continue;   // below that the + 1e-10 of SpectralService.java:81 takes over
                    }
                    // the fp64 contract of include/specgpu.h ("Numerical contract"), both statements
                    final double tol = Math.max(8e-15 * (Math.log(nfft) / Math.log(2)) + 5e-14, 4e-17 * nfft);
                    final double lin = Math.abs(mg - mw) / peak;
                    worstLin = Math.max(worstLin, lin);
                    assertTrue(lin <= tol, name + " line " + l + " bin " + i + ": " + lin + " M > " + tol + " M");
                    if (mw >= 1e-5 * peak) {
                        final double dbTol = Math.max(1e-9, 3e-12 * nfft);
                        final double db = Math.abs(got[i] - want);
                        worstDb = Math.max(worstDb, db);
                        assertTrue(db <= dbTol, name + " line " + l + " bin " + i + ": " + db + " dB > " + dbTol + " dB");
                    }
                }
            }
        }
        System.out.println(name + (DROP_IN ? ": max linear error " + worstLin + " M, max dB error " + worstDb : ": max " + worstUlps + " ulp"));
    }
}
