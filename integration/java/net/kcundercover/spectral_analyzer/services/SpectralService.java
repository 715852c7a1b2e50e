package net.kcundercover.spectral_analyzer.services;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.MappedByteBuffer;
import org.springframework.stereotype.Service;

/**
 * GPU-backed replacement for the reference service of the same name: same
 * package, class and {@code computeMagnitudes} signature, so MainController's
 * {@code @Autowired} field keeps working unchanged.  All numeric work is done by
 * libspecgpu.so (HIP kernels for MI355X) through the JNI shim
 * integration/jni/specgpu_jni.c; this class holds no FFT code.
 *
 * <p>Thread safety: the reference's service is stateless and may be called from any thread.  So
 * may this one -- the native context takes its own lock for the length of every call (see
 * include/specgpu.h), concurrent callers are serialised inside the library.
 *
 * Added on top of the reference API: {@link #computeWaterfall} (the whole slice
 * loop of MainController.updateDisplay in one call) and {@link #welchPsd}
 * (the PSD dialog's Welch estimate).
 */
@Service
public class SpectralService implements AutoCloseable {

    /** window ids of include/specgpu.h */
    public static final int WINDOW_RECT = 0, WINDOW_HANN = 1;
    /** PSD scalings of include/specgpu.h */
    public static final int PSD_DENSITY = 0, PSD_SPECTRUM = 1;

    static {
        System.loadLibrary("specgpu_jni");
    }

    private static final java.lang.ref.Cleaner CLEANER = java.lang.ref.Cleaner.create();

    private final long handle;

    /** Binds GPU 0 (override with -Dspecgpu.device=N). */
    public SpectralService() {
        this.handle = nativeCreate(Integer.getInteger("specgpu.device", 0), 0);
    }

    /**
     * One spectrogram line, exactly as the reference returns it:
     * {@code out[(i + nfft/2) % nfft] = 20 log10(|X_i| + 1e-10)}.
     *
     * @throws IllegalArgumentException  nfft is not a power of two
     * @throws IndexOutOfBoundsException the slice leaves the buffer
     */
    public double[] computeMagnitudes(MappedByteBuffer buffer, int startByte, int nfft, String datatype) {
        double[] line = new double[Math.max(nfft, 0)];
        nativeComputeMagnitudes(handle, buffer, startByte, nfft, datatype,
                buffer.order() == ByteOrder.BIG_ENDIAN, line);
        return line;
    }

    /**
     * All {@code nLines} lines of a redraw in one device call; line t starts
     * {@code t * hop} samples after {@code startByte}.  Lines that run past the end of
     * the buffer come back as -150.0, as the caller's loop used to fill them.
     * Returned row-major, {@code nLines x nfft}.
     */
    public float[] computeWaterfall(ByteBuffer buffer, long startByte, int nfft, int hop, int nLines,
                                    String datatype, int window) {
        float[] tile = new float[Math.multiplyExact(nLines, nfft)];
        nativeWaterfall(handle, buffer, startByte, nativeDtype(datatype), nfft, hop, nLines, window, -150.0, tile);
        return tile;
    }

    /**
     * {@link #computeWaterfall} with the lines sharded over several services, one per GPU of the node
     * ({@code new SpectralService(0) ... new SpectralService(7)}): lines are independent
     * (MainController.updateDisplay computes every {@code waterfall[t]} from its own span), so shard
     * {@code r} of {@code n} takes the contiguous lines {@code [r L / n, (r + 1) L / n)}; inside the
     * library every device stages and transforms its own span on a host thread of its own and copies
     * its rows into {@code tile}.  {@code services[0]} reports failures.  One JVM, many GPUs.
     */
    public static float[] computeWaterfallMulti(SpectralService[] services, ByteBuffer buffer, long startByte,
                                                int nfft, int hop, int nLines, String datatype, int window) {
        long[] handles = new long[services.length];
        for (int i = 0; i < services.length; i++) {
            handles[i] = services[i].handle;
        }
        float[] tile = new float[Math.multiplyExact(nLines, nfft)];
        nativeWaterfallMulti(handles, buffer, startByte, nativeDtype(datatype), nfft, hop, nLines, window, -150.0, tile);
        return tile;
    }

    /**
     * A batch of Welch PSDs -- {@code nPsd} spans of the buffer, {@code psdStrideBytes} apart, e.g. one per annotation --
     * spread over several services, one per GPU: the PSDs are independent, service {@code r} of {@code n} takes the
     * contiguous range {@code [r nPsd / n, (r + 1) nPsd / n)} on a host thread of its own inside the library.
     * Returns {@code nPsd * nfft} values, PSD {@code b} at {@code [b * nfft, (b + 1) * nfft)}; {@code freqOut}
     * (may be null) receives the frequency axis.  {@code services[0]} reports failures.
     */
    public static float[] welchPsdMulti(SpectralService[] services, ByteBuffer buffer, long startByte, long psdStrideBytes,
                                        int nPsd, String datatype, double sampleRate, int nfft, int hop, int segments,
                                        int window, int scaling, boolean decibel, double[] freqOut) {
        long[] handles = new long[services.length];
        for (int i = 0; i < services.length; i++) {
            handles[i] = services[i].handle;
        }
        double[] freq = freqOut != null ? freqOut : new double[nfft];
        float[] psd = new float[Math.multiplyExact(nPsd, nfft)];
        nativeWelchMulti(handles, buffer, startByte, psdStrideBytes, nPsd, nativeDtype(datatype), nfft, hop, segments,
                window, scaling, sampleRate, decibel, freq, psd);
        return psd;
    }

    /** Welch PSD of the samples from {@code startByte}; returns {frequency axis, psd}. */
    public double[][] welchPsd(ByteBuffer buffer, long startByte, String datatype, double sampleRate,
                               int nfft, int hop, int segments, int window, int scaling, boolean decibel) {
        double[] freq = new double[nfft];
        float[] psd = new float[nfft];
        nativeWelch(handle, buffer, startByte, nativeDtype(datatype), nfft, hop, segments, window, scaling,
                sampleRate, decibel, freq, psd);
        double[] wide = new double[nfft];
        for (int i = 0; i < nfft; i++) {
            wide[i] = psd[i];
        }
        return new double[][] {freq, wide};
    }

    /**
     * One redraw in one device call: the {@code canvasW} lines of {@code MainController.updateDisplay()}
     * followed by {@code renderSpectrogram} / {@code getColorForMagnitude}.  Returns {@code height * width}
     * IntArgb pixels, row 0 on top, for
     * {@code PixelWriter.setPixels(0, 0, width, height, PixelFormat.getIntArgbInstance(), argb, 0, width)}.
     *
     * @param colorMap 0 = "Grayscale", 1 = "Heatmap"
     */
    public int[] renderWaterfall(ByteBuffer buffer, long startByte, int nfft, int hop, String datatype,
                                 int width, int height, double sampleRate, double minDecibel,
                                 double maxDecibel, int colorMap) {
        int[] argb = new int[Math.multiplyExact(width, height)];
        nativeWaterfallRender(handle, buffer, startByte, nativeDtype(datatype), nfft, hop, width, WINDOW_RECT,
                height, sampleRate, minDecibel, maxDecibel, colorMap, argb);
        return argb;
    }

    /**
     * Same argument shape as {@code PowerSpectralDensity.calculatePsdWelch(data, fs, nfft)} in the
     * PSD dialog: {@code data[0]} = I, {@code data[1]} = Q.  {@code nfft} is any positive length --
     * the dialog passes {@code data[0].length} for bursts shorter than 8192 samples.  Hann window,
     * 50 % overlap, density scaling, fp64 throughout; returns {frequency axis, psd}.
     *
     * <p>Row 1 is in DECIBELS ({@code 10 log10(P + 1e-20)}, dB/Hz): that is what the caller reads.
     * {@code AnalysisDialogController.updatePSDChart} adds a dB offset to it (lines 319-328), the
     * clicked levels are labelled {@code "%.1f dB"} (612, 626), the SNR is their difference (675, 757)
     * and the report prints "dB/Hz" (751).  {@link #calculatePsdWelch(double[][], double, int, boolean)}
     * with {@code decibel = false} returns the linear power density.
     */
    public double[][] calculatePsdWelch(double[][] data, double sampleRate, int nfft) {
        return calculatePsdWelch(data, sampleRate, nfft, true);
    }

    /** {@link #calculatePsdWelch(double[][], double, int)} with the unit of row 1 chosen by the caller. */
    public double[][] calculatePsdWelch(double[][] data, double sampleRate, int nfft, boolean decibel) {
        double[] freq = new double[Math.max(nfft, 0)];
        double[] psd = new double[Math.max(nfft, 0)];
        nativeWelchPlanar(handle, data[0], data[1], nfft, Math.max(nfft / 2, 1), WINDOW_HANN, PSD_DENSITY,
                sampleRate, decibel, freq, psd);
        return new double[][] {freq, psd};
    }

    /**
     * The loop of {@code AnalysisDialogController.updateMagnitudeChart}: {@code 20 log10} of the
     * exponential moving average (weight {@code alpha}) of {@code hypot(data[0][i], data[1][i])},
     * one value per sample; the caller skips non-finite values as the dialog does.
     */
    public double[] magnitudeTrace(double[][] data, double alpha) {
        double[] out = new double[data[0].length];
        nativeTrace(handle, 0, data[0], data[1], alpha, 0.0, 0.0, out);
        return out;
    }

    /**
     * The loop of {@code AnalysisDialogController.updateFrequencyChart}: smoothed instantaneous
     * frequency in Hz plus {@code centerFreq}, for samples 1 .. N-1.
     */
    public double[] instFreqTrace(double[][] data, double alpha, double sampleRate, double centerFreq) {
        double[] out = new double[Math.max(data[0].length - 1, 0)];
        nativeTrace(handle, 1, data[0], data[1], alpha, sampleRate, centerFreq, out);
        return out;
    }

    /**
     * A recording on disk, opened by the native library itself: path, header size and 64-bit offsets
     * instead of the at most 2 GiB {@code MappedByteBuffer} of the reference's SigMfHelper.  Obtain the
     * three values from the replacement {@code sigmf.SigMfHelper}:
     * {@code openRecording(helper.getDataPath(), helper.getHeaderBytes())}.
     */
    public final class Recording implements AutoCloseable {
        private long rec;
        /** closes the native handle (descriptor + whole-file mapping) should the owner forget to */
        private final java.lang.ref.Cleaner.Cleanable cleanable;

        private Recording(long rec) {
            this.rec = rec;
            final long[] box = {rec};
            this.cleanable = CLEANER.register(this, () -> {
                if (box[0] != 0) {
                    nativeCloseRecording(box[0]);
                    box[0] = 0;
                }
            });
        }

        /** Payload bytes after the header (the mapped buffer's capacity, without the cap). */
        public long length() {
            return nativeRecordingBytes(rec);
        }

        /** {@code computeMagnitudes} with a 64-bit {@code startByte}. */
        public double[] computeMagnitudes(long startByte, int nfft, String datatype, boolean bigEndian) {
            double[] line = new double[Math.max(nfft, 0)];
            nativeComputeMagnitudesRecording(handle, rec, startByte, nfft, datatype, bigEndian, line);
            return line;
        }

        /** The whole slice loop of a redraw, read straight from the file; EOF lines are -150.0. */
        public float[] computeWaterfall(long startByte, int nfft, int hop, int nLines, String datatype, int window) {
            float[] tile = new float[Math.multiplyExact(nLines, nfft)];
            nativeWaterfallRecording(handle, rec, startByte, nativeDtype(datatype), nfft, hop, nLines, window, -150.0, tile);
            return tile;
        }

        @Override
        public void close() {
            rec = 0;
            cleanable.clean();  // runs the action once: closes the handle now
        }
    }

    /** Opens the data file of a recording; {@code headerBytes} = {@code core:header_bytes} of the first capture. */
    public Recording openRecording(java.nio.file.Path dataFile, long headerBytes) {
        return new Recording(nativeOpenRecording(handle, dataFile.toString(), headerBytes));
    }

    @Override
    public void close() {
        nativeDestroy(handle);
    }

    /**
     * A tuning / testing knob of the native library (include/specgpu.h, spec_set_option): e.g.
     * {@code setOption("multi_verify", 1)} on {@code services[0]} makes {@link #computeWaterfallMulti} checksum every piece a
     * peer GPU sends before it leaves and where it landed.
     *
     * @throws IllegalArgumentException unknown key
     */
    public void setOption(String key, long value) {
        nativeSetOption(handle, key, value);
    }

    /** Current value of a knob, or of a read-only state such as {@code "multi_peer_access"} / {@code "multi_verified"}. */
    public long getOption(String key) {
        return nativeGetOption(handle, key);
    }

    private static native void nativeSetOption(long handle, String key, long value);
    private static native long nativeGetOption(long handle, String key);
    private static native long nativeCreate(int device, int flags);
    private static native void nativeDestroy(long handle);
    private static native void nativeComputeMagnitudes(long handle, ByteBuffer buffer, int startByte, int nfft,
                                                       String datatype, boolean bigEndian, double[] out);
    private static native void nativeWaterfall(long handle, ByteBuffer buffer, long startByte, int dtype, int nfft,
                                               int hop, long nLines, int window, double eofFill, float[] out);
    private static native void nativeWaterfallMulti(long[] handles, ByteBuffer buffer, long startByte, int dtype, int nfft,
                                                    int hop, long nLines, int window, double eofFill, float[] out);

    private static native void nativeWelch(long handle, ByteBuffer buffer, long startByte, int dtype, int nfft,
                                           int hop, int segments, int window, int scaling, double sampleRate,
                                           boolean decibel, double[] freq, float[] psd);
    private static native void nativeWelchMulti(long[] handles, ByteBuffer buffer, long startByte, long psdStrideBytes,
                                                int nPsd, int dtype, int nfft, int hop, int segments, int window,
                                                int scaling, double sampleRate, boolean decibel, double[] freq, float[] psd);
    private static native void nativeWaterfallRender(long handle, ByteBuffer buffer, long startByte, int dtype,
                                                     int nfft, int hop, int width, int window, int height,
                                                     double sampleRate, double minDb, double maxDb, int colorMap,
                                                     int[] argb);
    private static native void nativeWelchPlanar(long handle, double[] re, double[] im, int nfft, int hop,
                                                 int window, int scaling, double sampleRate, boolean decibel,
                                                 double[] freq, double[] psd);
    private static native void nativeTrace(long handle, int which, double[] re, double[] im, double alpha,
                                           double sampleRate, double centerFreq, double[] out);
    private static native int nativeDtype(String datatype);
    private static native long nativeOpenRecording(long handle, String path, long headerBytes);
    private static native long nativeRecordingBytes(long recording);
    private static native void nativeCloseRecording(long recording);
    private static native void nativeWaterfallRecording(long handle, long recording, long startByte, int dtype,
                                                        int nfft, int hop, long nLines, int window, double eofFill,
                                                        float[] out);
    private static native void nativeComputeMagnitudesRecording(long handle, long recording, long startByte,
                                                                int nfft, String datatype, boolean bigEndian,
                                                                double[] out);
}
