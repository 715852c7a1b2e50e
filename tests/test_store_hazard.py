"""ISA lint of the kernels that store 16 bytes per lane with buffer instructions (tools/check_store_hazard.py): no
vector-ALU write to a store's data registers within two instruction slots behind it.  Found on the GPU in round 4
(spec_k_v3h.hip: the last lanes of a wave left with a restored register's value); hipcc's hazard recogniser covers the
pattern only for stores without a scalar offset register.  Compiles the translation unit to assembly (no GPU needed) and
also shows that the lint sees the pattern when the padding is taken out."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spectral_analyzer_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def sources_with_wide_buffer_stores():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") and re.search(r"raw_buffer_store_b(96|128)", open(os.path.join(CSRC, f)).read()):
            out.append(f)
    return out


def test_the_lint_covers_every_source_with_16_byte_buffer_stores():
    assert sources_with_wide_buffer_stores() == ["spec_k_v3h.hip"]


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    import shutil
    if not shutil.which("hipcc") and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = str(tmp_path_factory.mktemp("v3h") / "v3h.s")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only",
                           os.path.join(CSRC, "spec_k_v3h.hip"), "-o", out], stderr=subprocess.DEVNULL)
    return out


def test_no_valu_write_behind_a_16_byte_buffer_store(asm):
    import check_store_hazard
    text = open(asm).read()
    assert text.count("buffer_store_dwordx4") >= 16 * 4      # the fp64 output path of every cf64 variant at least
    assert check_store_hazard.check(asm) == []


def test_the_lint_sees_the_unpadded_pattern(asm, tmp_path):
    """The same assembly with the wait states removed: wherever hipcc restored a borrowed half of the data tuple behind the
    store (it does in the cf64 variants of this build), the lint reports it; and a hand-written instance always."""
    import check_store_hazard
    stripped = str(tmp_path / "stripped.s")
    open(stripped, "w").write(re.sub(r"\ts_nop 1\n", "", open(asm).read()))
    hand = str(tmp_path / "hand.s")
    open(hand, "w").write("_Zk:\n\tbuffer_store_dwordx4 v[4:7], v1, s[8:11], s0 offen\n\tv_mov_b32_e32 v5, v9\n\ts_endpgm\n"
                          "_Zok:\n\tbuffer_store_dwordx4 v[4:7], v1, s[8:11], s0 offen\n\ts_nop 1\n\tv_mov_b32_e32 v5, v9\n\ts_endpgm\n")
    found = check_store_hazard.check(hand)
    assert len(found) == 1 and found[0].startswith("_Zk")
    assert len(check_store_hazard.check(stripped)) >= len(check_store_hazard.check(asm))


# ---- waterfall loops (build.py waterfall_findings) -------------------------------------------------------------------------
WATERFALL = """_Zwf:
\tv_add_u32_e32 v34, 0x4000, v46
\ts_mov_b64 s[0:1], exec
.LBB1_8:
\tv_readfirstlane_b32 s2, v34
\ts_nop 1
\tv_cmp_eq_u32_e32 vcc, s2, v34
\ts_and_saveexec_b64 vcc, vcc
\tbuffer_load_dwordx2 v[32:33], v113, s[4:7], s2 offen nt
\ts_xor_b64 exec, exec, vcc
\ts_cbranch_execnz .LBB1_8
\ts_mov_b64 exec, s[0:1]
\ts_endpgm
_Zplain:
.LBB2_1:
\tbuffer_load_dwordx2 v[32:33], v113, s[4:7], s2 offen nt
\tv_readfirstlane_b32 s3, v1
\ts_cbranch_scc1 .LBB2_1
\ts_endpgm
"""


def test_the_build_lint_sees_a_waterfall_loop(tmp_path):
    """The headline kernel shipped with eight of these per line for five rounds (spec_v2.h, `iters`): the loop around a memory
    instruction whose scalar operand hipcc could not prove uniform.  The product build fails on one; here the detector itself."""
    from spectral_analyzer_amd import build as hip_build
    p = str(tmp_path / "wf.s")
    open(p, "w").write(WATERFALL)
    found = hip_build.waterfall_findings(p)
    assert len(found) == 1 and found[0].startswith("_Zwf: 1 waterfall")


def test_no_waterfall_loop_in_the_16_byte_store_kernels(asm):
    from spectral_analyzer_amd import build as hip_build
    assert hip_build.waterfall_findings(asm) == []


def test_the_product_library_was_built_with_both_lints():
    from spectral_analyzer_amd import build as hip_build
    stamp = hip_build.LIB + ".flags"
    if not os.path.exists(stamp):
        pytest.skip("library not built")
    assert "isa-lint: store hazard, waterfall loops" in open(stamp).read()
