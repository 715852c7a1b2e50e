package net.kcundercover.spectral_analyzer.sigmf;

import com.fasterxml.jackson.databind.DeserializationFeature;
import com.fasterxml.jackson.databind.ObjectMapper;
import com.fasterxml.jackson.databind.SerializationFeature;
import java.io.File;
import java.io.IOException;
import java.nio.ByteOrder;
import java.nio.MappedByteBuffer;
import java.nio.channels.FileChannel;
import java.nio.file.Files;
import java.nio.file.Path;
import java.nio.file.StandardOpenOption;
import java.util.List;

/**
 * Replacement for the reference class of the same name (same package, same public methods, so
 * MainController compiles unchanged) that does NOT stop at 2 GiB.
 *
 * <p>The reference maps {@code min(fileSize - headerBytes, Integer.MAX_VALUE)} bytes of the data file
 * and every caller addresses them with an {@code int} (MainController.updateDisplay casts the byte
 * offset, SpectralService.computeMagnitudes takes {@code int startByte}); the rest of a long recording
 * cannot be shown.  This class keeps the loader's rules -- which file holds the samples, the header
 * skip, the byte order -- and describes the payload by <em>path, header size and 64-bit length</em>
 * ({@link #getDataPath()}, {@link #getHeaderBytes()}, {@link #getDataLength()}).  The GPU service opens
 * the file itself from those three values ({@code SpectralService.openRecording}) and reads the slices
 * it needs into pinned memory; nothing is mapped for it.
 *
 * <p>{@link #getDataBuffer()} is still there for callers that have not been touched: it maps the first
 * window of the payload (at most 2 GiB - 1 bytes), exactly what the reference handed out, and
 * {@link #mapWindow(long, long)} maps any other window on request.
 */
public class SigMfHelper {
    /** Largest window a single MappedByteBuffer can cover. */
    public static final long MAX_WINDOW = Integer.MAX_VALUE;

    private final ObjectMapper json = new ObjectMapper()
        .configure(DeserializationFeature.FAIL_ON_UNKNOWN_PROPERTIES, false);

    private SigMfMetadata metadata;
    private Path metaPath;
    private Path dataPath;
    private long headerBytes;
    private long dataLength;
    private ByteOrder order = ByteOrder.BIG_ENDIAN;
    private MappedByteBuffer firstWindow;

    public SigMfHelper() {
    }

    /**
     * Reads the meta file and locates the samples; maps nothing but the first window.
     *
     * @param metaFile path of the {@code .sigmf-meta} file
     * @throws Exception I/O or JSON failure
     */
    public void load(Path metaFile) throws Exception {
        SigMfMetadata parsed = json.readValue(metaFile.toFile(), SigMfMetadata.class);
        Path data = locateData(metaFile, parsed);
        long skip = headerOf(parsed);
        long size = Files.size(data);

        this.metadata = parsed;
        this.metaPath = metaFile;
        this.dataPath = data;
        this.headerBytes = skip;
        this.dataLength = Math.max(0L, size - skip);
        String datatype = parsed.global() == null ? null : parsed.global().datatype();
        this.order = datatype != null && datatype.endsWith("_le") ? ByteOrder.LITTLE_ENDIAN : ByteOrder.BIG_ENDIAN;
        this.firstWindow = mapWindow(0L, Math.min(dataLength, MAX_WINDOW));
    }

    /** {@code core:dataset} next to the meta file when present, else the {@code .sigmf-data} sibling. */
    private static Path locateData(Path metaFile, SigMfMetadata meta) {
        Path dir = metaFile.getParent();
        boolean named = meta.global() != null && meta.global().dataset() != null;
        if (named && dir != null) {
            return dir.resolve(meta.global().dataset());
        }
        return Path.of(metaFile.toString().replace(".sigmf-meta", ".sigmf-data"));
    }

    /** {@code core:header_bytes} of the first capture, 0 when absent. */
    private static long headerOf(SigMfMetadata meta) {
        if (meta.captures() == null || meta.captures().isEmpty()) {
            return 0L;
        }
        Long h = meta.captures().get(0).headerBytes();
        return h == null ? 0L : h;
    }

    /**
     * Maps {@code length} payload bytes starting {@code offset} bytes after the header, with the
     * recording's byte order.  {@code length} must not exceed {@link #MAX_WINDOW}.
     */
    public MappedByteBuffer mapWindow(long offset, long length) throws IOException {
        if (offset < 0 || length < 0 || length > MAX_WINDOW || offset > dataLength - length) {
            throw new IndexOutOfBoundsException("window [" + offset + ", +" + length + ") of " + dataLength);
        }
        try (FileChannel ch = FileChannel.open(dataPath, StandardOpenOption.READ)) {
            MappedByteBuffer window = ch.map(FileChannel.MapMode.READ_ONLY, headerBytes + offset, length);
            window.order(order);
            return window;
        }
    }

    /** The data file ({@code core:dataset} or the {@code .sigmf-data} sibling). */
    public Path getDataPath() {
        return dataPath;
    }

    /** Bytes in front of the first sample. */
    public long getHeaderBytes() {
        return headerBytes;
    }

    /** Payload bytes after the header -- the whole recording, not the first 2 GiB. */
    public long getDataLength() {
        return dataLength;
    }

    /** Byte order of the samples: little endian for datatypes ending in {@code _le}, else big endian. */
    public ByteOrder getByteOrder() {
        return order;
    }

    public File getCurrentMetaFile() {
        if (metaPath == null) {
            return null;
        }
        File f = metaPath.toFile();
        if (f.getName().endsWith(".sigmf-data")) {
            return new File(f.getAbsolutePath().replace(".sigmf-data", ".sigmf-meta"));
        }
        return f;
    }

    public SigMfMetadata getMetadata() {
        return metadata;
    }

    /** First window of the payload (at most 2 GiB - 1 bytes): what the reference class returns. */
    public MappedByteBuffer getDataBuffer() {
        return firstWindow;
    }

    public List<SigMfAnnotation> getParsedAnnotations() {
        return metadata.annotations();
    }

    /** Writes the meta file back with {@code annotationList} in place of the stored annotations. */
    public void saveSigMF(List<SigMfAnnotation> annotationList) {
        this.metadata = new SigMfMetadata(metadata.global(), metadata.captures(), annotationList);
        try {
            json.enable(SerializationFeature.INDENT_OUTPUT);
            json.writeValue(getCurrentMetaFile(), this.metadata);
        } catch (IOException e) {
            throw new java.io.UncheckedIOException("cannot save " + getCurrentMetaFile(), e);
        }
    }
}
