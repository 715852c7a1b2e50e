package net.kcundercover.spectral_analyzer.services;

import java.nio.ByteOrder;
import java.nio.MappedByteBuffer;
import org.springframework.stereotype.Service;

/**
 * GPU-backed replacement for the reference service of the same name (same package, class and
 * method signatures, so the controllers that inject it compile unchanged).  The burst reader,
 * the frequency shift and the decimating filter run in libspecgpu.so through
 * integration/jni/specgpu_jni.c.
 *
 * The reader reproduces the reference arithmetic exactly.  The filter is the library's own
 * stated specification (include/specgpu.h, spec_down_convert): JDSP's Resampler source is not
 * part of the reference tree, so its taps cannot be matched -- a moving average of {@code down}
 * taps in fast mode, a Hamming-windowed sinc of {@code 8 * down + 1} taps otherwise.
 */
@Service
public class ExtractDownConvertService implements AutoCloseable {

    static {
        System.loadLibrary("specgpu_jni");
    }

    private final long handle;

    /** Binds GPU 0 (override with -Dspecgpu.device=N). */
    public ExtractDownConvertService() {
        this.handle = nativeCreate(Integer.getInteger("specgpu.device", 0), 0);
    }

    /** Six-argument form of the reference: it selects the conventional (non-fast) filter. */
    public double[][] extractAndDownConvert(MappedByteBuffer buffer, long startSample, int count,
                                            String datatype, double freqOff, int down) {
        return extractAndDownConvert(buffer, startSample, count, datatype, freqOff, down, false);
    }

    /**
     * @param freqOff frequency shift in cycles per input sample
     * @param down    decimation factor; the result holds {@code count / down} samples
     * @param fast    moving-average filter (true) or low-pass FIR (false)
     * @return {@code [0]} = I, {@code [1]} = Q
     * @throws IndexOutOfBoundsException the burst leaves the buffer
     */
    public double[][] extractAndDownConvert(MappedByteBuffer buffer, long startSample, int count,
                                            String datatype, double freqOff, int down, boolean fast) {
        double[][] result = new double[2][Math.max(count, 0) / Math.max(down, 1)];
        nativeExtractAndDownConvert(handle, buffer, startSample, count, datatype,
                buffer.order() == ByteOrder.BIG_ENDIAN, freqOff, down, fast, result[0], result[1]);
        return result;
    }

    @Override
    public void close() {
        nativeDestroy(handle);
    }

    private static native long nativeCreate(int device, int flags);
    private static native void nativeDestroy(long handle);
    private static native void nativeExtractAndDownConvert(long handle, MappedByteBuffer buffer, long startSample,
                                                           int count, String datatype, boolean bigEndian,
                                                           double freqOff, int down, boolean fast,
                                                           double[] re, double[] im);
}
