"""The JNI shim (integration/jni/specgpu_jni.c) is compiled with -Wall -Wextra -Werror against the JNI
subset in tests/jni_stub/jni.h (no JDK in the image) and its entry points are CALLED through a fake
JNIEnv by tests/jni_stub/harness.c, which compares every result with the same request made through the
C ABI.  Without a GPU the harness still runs: nativeCreate must come back as a RuntimeException that
carries the library's "no CPU backend" text."""
import os
import subprocess

import pytest

from spectral_analyzer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "jni_stub")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("jni")
    libdir = os.path.dirname(_lib.LIB_PATH)
    _lib.load()
    shim = str(out / "libspecgpu_jni.so")
    warn = ["-std=c99", "-Wall", "-Wextra", "-Werror", "-O1"]
    subprocess.check_call(["gcc", *warn, "-shared", "-fPIC", "-I" + STUB, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "integration", "jni", "specgpu_jni.c"), "-L" + libdir, "-lspecgpu",
                           "-Wl,-rpath," + libdir, "-o", shim])
    exe = str(out / "harness")
    subprocess.check_call(["gcc", *warn, "-D_POSIX_C_SOURCE=200809L", "-I" + STUB, "-I" + os.path.join(ROOT, "include"),
                           os.path.join(STUB, "harness.c"), shim, "-L" + libdir, "-lspecgpu", "-lm",
                           "-Wl,-rpath," + libdir + ":" + str(out), "-o", exe])
    return exe


def test_shim_exports_the_mangled_names(harness):
    shim = os.path.join(os.path.dirname(harness), "libspecgpu_jni.so")
    syms = subprocess.run(["nm", "-D", "--defined-only", shim], capture_output=True, text=True, check=True).stdout
    for n in ("SpectralService_nativeComputeMagnitudes", "SpectralService_nativeWaterfall",
              "SpectralService_nativeWelchPlanar", "SpectralService_nativeSetOption", "SpectralService_nativeGetOption",
              "ExtractDownConvertService_nativeExtractAndDownConvert"):
        assert "Java_net_kcundercover_spectral_1analyzer_services_" + n in syms


def test_shim_reports_a_missing_gpu_as_an_exception(harness):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([harness], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2, r.stdout + r.stderr
    assert "java/lang/RuntimeException" in r.stdout and "no CPU backend" in r.stdout


@pytest.mark.gpu
def test_shim_called_through_a_fake_jnienv_matches_the_c_abi(harness):
    r = subprocess.run([harness], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "jni harness ok" in r.stdout
