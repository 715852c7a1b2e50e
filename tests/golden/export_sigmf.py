#!/usr/bin/env python3
"""Exports the spectrogram fixtures tests/golden/wf_*.npz as files a JVM can read -- the parity-pinning kit of
INTEGRATION.md ("Pinning parity"):

    integration/java-test/fixtures/<name>.sigmf-meta     SigMF metadata (core:datatype, core:header_bytes, core:dataset)
    integration/java-test/fixtures/<name>.sigmf-data     the recording's bytes, exactly as in the .npz (+ header where stated)
    integration/java-test/fixtures/<name>.expected.f64   little-endian float64 [lines][nfft]: 20 log10(|X| + 1e-10), fftshifted
    integration/java-test/fixtures/manifest.json         one entry per fixture: nfft, hop, lines, window, what the
                                                         UNMODIFIED reference is expected to return for it

integration/java-test/SpectralServiceParityTest.java loads every pair with the reference's own SigMfHelper.load
(sigmf/SigMfHelper.java:43-94) and calls the reference's own SpectralService.computeMagnitudes
(services/SpectralService.java:33-85) line by line; tests/test_parity_kit.py checks here, without a JVM, that the exported
files reproduce the .npz fixtures bit for bit through spectral_analyzer_amd/sigmf.py.

A few fixtures carry a header (core:header_bytes, SigMfHelper.java:60-67) or name their data file through core:dataset
(SigMfHelper.java:49-53), so that the loader rules are pinned together with the arithmetic.

    python tests/golden/export_sigmf.py
"""
import glob
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(os.path.dirname(HERE)), "integration", "java-test", "fixtures")

HEADER = bytes((37 * i + 11) & 0xFF for i in range(24))        # 24 bytes the loader must skip


def export(out_dir: str = OUT):
    os.makedirs(out_dir, exist_ok=True)
    manifest = []
    for path in sorted(glob.glob(os.path.join(HERE, "wf_*.npz"))):
        z = np.load(path)
        name = os.path.splitext(os.path.basename(path))[0]
        dt, nfft, hop, window = str(z["datatype"]), int(z["nfft"]), int(z["hop"]), int(z["window"])
        iq, db = z["iq"], z["db"]
        lines = db.shape[0] - 1                                  # the last line of the .npz is the caller's EOF fill (MC:994-998)
        assert np.all(db[lines] == -150.0)
        header = HEADER if nfft == 1024 else b""                 # the 1024-point fixtures carry a header
        dataset = name + ".raw" if (nfft == 64 and dt.startswith("ci16")) else None   # two name their data file
        data_name = dataset or name + ".sigmf-data"
        with open(os.path.join(out_dir, data_name), "wb") as f:
            f.write(header)
            f.write(iq.tobytes())
        meta = {"global": {"core:datatype": dt, "core:sample_rate": 1.0e6, "core:version": "1.0.0"},
                "captures": [{"core:sample_start": 0}], "annotations": []}
        if header:
            meta["captures"][0]["core:header_bytes"] = len(header)
        if dataset:
            meta["global"]["core:dataset"] = dataset
        with open(os.path.join(out_dir, name + ".sigmf-meta"), "w") as f:
            json.dump(meta, f, indent=1)
            f.write("\n")
        db[:lines].astype("<f8").tofile(os.path.join(out_dir, name + ".expected.f64"))
        if dt.startswith("cf64"):
            ref = "flat-200"      # SS:35-63 has no cf64 branch: zeros in, 20 log10(1e-10) = -200.0 out (the drop-in decodes cf64, EDC:79-81)
        elif window != 0:
            ref = "no-window"     # the reference applies no window: drop-in only
        else:
            ref = "lines"
        manifest.append({"name": name, "datatype": dt, "nfft": nfft, "hop": hop, "window": window, "lines": lines,
                         "bytes_per_sample": int(iq.size // ((lines - 1) * hop + nfft)), "header_bytes": len(header),
                         "data_file": data_name, "reference": ref})
    with open(os.path.join(out_dir, "manifest.json"), "w") as f:
        json.dump({"generated_by": "tests/golden/export_sigmf.py", "tolerance_ulp": 4, "fixtures": manifest}, f, indent=1)
        f.write("\n")
    return manifest


if __name__ == "__main__":
    m = export()
    print("wrote %d fixtures to %s" % (len(m), OUT))
