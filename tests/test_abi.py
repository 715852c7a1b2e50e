"""CPU checks of the C-ABI boundary: the shared library loads, exports every symbol that
include/specgpu.h declares, the pure-host entry points behave like the reference's tables,
and a context cannot be created without a GPU (there is no CPU fallback)."""
import ctypes
import os
import re

import pytest

import spectral_analyzer_amd as sa
from spectral_analyzer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "specgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spec_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libspecgpu.so does not export " + n
    assert set(names) == set(_lib.SIGNATURES), "ctypes table and header disagree"


def test_no_torch_in_the_abi():
    # the boundary is plain C: the library must not link libtorch / libc10
    import subprocess
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    # library NAMES only: the load addresses ldd prints may contain "c10" by chance (ASLR)
    names = [line.split()[0] for line in out.splitlines() if line.strip()]
    assert not any("torch" in n or "c10" in n for n in names), names
    assert any("amdhip64" in n for n in names), names


def test_dtype_table_matches_reference_rules():
    # SS:35-38 startsWith + SMH:87-91 "_le" suffix rule
    f = sa.dtype_from_sigmf
    assert f("cf32_le") == sa.DT_CF32_LE and f("cf32_be") == sa.DT_CF32_BE and f("cf32") == sa.DT_CF32_BE
    assert f("ci16_le") == sa.DT_CI16_LE and f("ci16_be") == sa.DT_CI16_BE
    assert f("cu8") == sa.DT_CU8 and f("ci8") == sa.DT_CI8 and f("cu8_le") == sa.DT_CU8
    assert f("cf64_le") == sa.DT_CF64_LE and f("cf64_be") == sa.DT_CF64_BE
    assert f("ri16_le") == sa.DT_UNKNOWN and f("") == sa.DT_UNKNOWN
    # Global.java:67-79 incl. fallback 8
    assert [sa.bytes_per_sample(d) for d in ("cf32_le", "ci16_le", "cu8", "ci8", "cf64_le", "zzz")] == [8, 4, 2, 2, 16, 8]


def test_count_lines_is_the_range_test():
    lib = _lib.load()
    # MC:987: byteOffset + nfft*bps <= capacity
    assert lib.spec_count_lines(8 * 4096, 0, sa.DT_CF32_LE, 4096, 2048) == 1
    assert lib.spec_count_lines(8 * 4096 - 1, 0, sa.DT_CF32_LE, 4096, 2048) == 0
    assert lib.spec_count_lines(8 << 30, 0, sa.DT_CF32_LE, 4096, 2048) == 524287
    assert lib.spec_count_lines(4 * 1000, 4 * 300, sa.DT_CI16_LE, 256, 256) == 2
    assert lib.spec_count_lines(100, 200, sa.DT_CU8, 16, 16) == 0
    assert lib.spec_count_lines(100, 0, sa.DT_CU8, 16, 0) == 0


def test_status_strings():
    lib = _lib.load()
    assert lib.spec_status_string(0) == b"SPEC_OK" and lib.spec_status_string(2) == b"SPEC_ERANGE"


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU backend"):
        sa.SpectralService(0)
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    assert lib.spec_create(0, None, 0, ctypes.byref(ctx)) == _lib.SPEC_EDEVICE and not ctx.value
    assert lib.spec_create(0, None, 0, None) == _lib.SPEC_EINVAL
    # NULL-context calls do not crash
    assert lib.spec_sync(None) == _lib.SPEC_EINVAL
    lib.spec_destroy(None)


def test_missing_library_is_an_import_error(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_header_is_plain_c_and_cpp(tmp_path):
    """include/specgpu.h compiles as C99 and as C++17 with nothing but the standard headers; the C++
    mirror include/specgpu.hpp and the example host build against the library without HIP or torch."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    c = tmp_path / "t.c"
    c.write_text('#include "specgpu.h"\nint main(void) { return (int)spec_bytes_per_sample(SPEC_DT_CI16_LE) - 4; }\n')
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I" + inc, str(c), "-L" + libdir,
                           "-lspecgpu", "-Wl,-rpath," + libdir, "-o", str(tmp_path / "t_c")])
    assert subprocess.call([str(tmp_path / "t_c")]) == 0
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + inc,
                           os.path.join(ROOT, "integration", "cpp", "example.cpp"), "-L" + libdir, "-lspecgpu",
                           "-Wl,-rpath," + libdir, "-o", str(tmp_path / "example")])
    # without a GPU the example reports the create failure and exits 2; with one it exits 0
    rc = subprocess.call([str(tmp_path / "example")])
    import torch
    assert rc == (0 if torch.cuda.is_available() else 2)


def test_jni_shim_covers_every_java_native_method():
    """No JDK here, so at least keep the two sides of the JNI boundary in step: every `native`
    method of the replacement Java classes has its mangled function in the shim, and vice versa."""
    import re
    shim = open(os.path.join(ROOT, "integration", "jni", "specgpu_jni.c")).read()
    jdir = os.path.join(ROOT, "integration", "java", "net", "kcundercover", "spectral_analyzer", "services")
    for cls, macro in (("SpectralService", "JNI_FN"), ("ExtractDownConvertService", "EDC_FN")):
        java = open(os.path.join(jdir, cls + ".java")).read()
        declared = set(re.findall(r"private static native [\w\[\]]+ (native\w+)\(", java))
        defined = set(re.findall(r"JNICALL %s\((native\w+)\)" % macro, shim))
        assert declared and declared == defined, (cls, declared ^ defined)
    assert "Java_net_kcundercover_spectral_1analyzer_services_ExtractDownConvertService_" in shim


# ---- the build: variant names, the ISA lint that runs inside it, the headline kernels' registers (round 5) ------------------
def test_build_variant_names_are_unique():
    """Round 4 lost a variant to a dictionary key defined twice; the table is built by a function that refuses that."""
    from spectral_analyzer_amd import build
    assert len(build.VARIANTS) >= 10
    with pytest.raises(ValueError, match="defined twice"):
        build._unique([("a", ([], [])), ("b", ([], [])), ("a", ([], []))])
    for name, (flags, units) in build.VARIANTS.items():
        assert units and all(u in build.SOURCES for u in units), name


def test_every_library_build_ran_the_isa_lint():
    """The store-data hazard lint (build.py store_hazard_findings) runs on the device assembly of every translation unit of
    every library build.py produces and fails the build on a finding; the stamp beside the library says so and names the
    toolchain the measurement behind the lint applies to."""
    from spectral_analyzer_amd import build
    lib = build.build()
    stamp = open(lib + ".flags").read()
    assert "isa-lint: store hazard" in stamp and "HIP version" in stamp
    bad = "_Zk:\n\tbuffer_store_dwordx4 v[4:7], v1, s[8:11], s0 offen\n\tv_mov_b32_e32 v5, v9\n\ts_endpgm\n"
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(bad)
    try:
        assert len(build.store_hazard_findings(f.name)) == 1
    finally:
        os.unlink(f.name)


def test_headline_kernels_do_not_spill():
    """The kernels behind the bench workloads keep everything in registers (a spill inside a line loop costs a scratch round trip
    per line: the first builds of the 65536-point pair kernel lost 10 % to eleven spilled registers).  Read from the spill report the
    build writes beside the library; a compiler or source change that makes one of them spill fails here, not on the GPU box."""
    import subprocess
    from spectral_analyzer_amd import build
    lib = build.build()
    rows = [l.split() for l in open(lib + ".spills").read().splitlines()]
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    spills = {n.replace("specgpu::(anonymous namespace)::", ""): int(r[1]) for n, r in zip(names, rows)}
    headline = {
        "v2_kernel<12, 4, 8, false, 0, false,": 0,     # cfg2: 4096 points, cf32, 50 % overlap
        "v2_kernel<12, 3, 8, false, 0, false,": 0,     # cfg3: ci16
        "v2_kernel<10, 4, 8, false, 0, false,": 0,     # cfg1
        "v2_kernel<14, 4, 8, true, 1, false,": 0,      # cfg4: 16384-point Welch segments, 75 % overlap
        "v2_kernel<14, 4, 16, false, 0, false,": 0,    # n16384
        "v2h_kernel<14, 4, false, false, false>": 0,   # n32768f
        "v2q_kernel<4, false, false>": 1,              # n65536f (one register, outside the line loop)
        "v2q_kernel<3, false, false>": 0,              # 65536 points, ci16
        "v3h_kernel<5, false, false, false>": 4,       # n16384d (round 4: 36 bytes of scratch per lane)
    }
    for pat, allowed in headline.items():
        hits = {k: v for k, v in spills.items() if pat in k}
        assert hits, "no kernel matches %r" % pat
        for k, v in hits.items():
            assert v <= allowed, "%s spills %d vector registers (allowed %d)" % (k, v, allowed)


def test_dispatch_table_is_generated_and_complete():
    """The 8192- / 16384-point choice between the family's kernel and the half-line kernel is a GENERATED table
    (tools/tune_dispatch.py -> csrc/spec_dispatch_table.h): every cell of size x format x hop class x
    window is there, carries the fractions it was decided from, both verdicts occur at both sizes (so "mid_single" /
    "small_single" = 2 really exercises both kernels; tests/test_gpu_v2h.py checks 0 / 1 / 2 on the GPU), and the C ABI reads it."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import tune_dispatch as td
    cells = td.parse_header()
    assert len(cells) == 2 * 6 * 4 * 2
    for n in td.SIZES:
        vals = {v for (size, *_), v in cells.items() if size == n}
        assert vals == {0, 1}, "size %d: the table never takes one of the two kernels" % n
    text = open(td.HEADER).read()
    assert "GENERATED by tools/tune_dispatch.py" in text and len(re.findall(r"rect [\d.]+ / [\d.]+\s+Hann [\d.]+ / [\d.]+", text)) == 48
    # emit(parse) round trip: the committed header is what the generator writes for the fractions in its own comments
    frac = {}
    rows = re.findall(r"rect ([\d.]+) / ([\d.]+)\s+Hann ([\d.]+) / ([\d.]+)", text)
    it = iter(rows)
    for n in td.SIZES:
        for f in td.FORMATS:
            for h in td.HOPS:
                r = next(it)
                frac[(n, f, h, 0)], frac[(n, f, h, 1)] = (float(r[0]), float(r[1])), (float(r[2]), float(r[3]))
    regenerated = td.emit(frac, "x")
    assert regenerated.split("#pragma once")[1] == text.split("#pragma once")[1]
    capi = open(os.path.join(ROOT, "spectral_analyzer_amd", "csrc", "spec_capi.hip")).read()
    assert "dispatch_half_line(" in capi and '#include "spec_dispatch_table.h"' in capi
