"""Build libspecgpu.so (HIP kernels + C ABI) for gfx950 with hipcc.

    python -m spectral_analyzer_amd.build [--force]

The library is built in-tree (spectral_analyzer_amd/lib/libspecgpu.so) so that
it travels with the repository snapshot; hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libspecgpu.so")
SOURCES = ["spec_capi.hip", "spec_k_f32.hip", "spec_k_f64.hip", "spec_k_large.hip", "spec_k_team.hip", "spec_k_v2s.hip", "spec_k_v2w.hip", "spec_k_v2r.hip", "spec_k_v2n.hip", "spec_k_v2h.hip", "spec_k_v2q.hip", "spec_k_v3d.hip", "spec_k_v3h.hip", "spec_misc.hip", "spec_burst.hip"]
ARCH = "gfx950"
# --offload-compress: the code objects are stored zstd-compressed in the library (about 360 kernel instantiations: 16 MB -> 2.5 MB);
# the HIP runtime unpacks them when the library is loaded
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
         "-ffp-contract=fast", "--offload-compress"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _newest_dep() -> float:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not os.path.isdir(os.path.join(CSRC, f))]
    deps += [os.path.join(CSRC, "experiments", f) for f in os.listdir(os.path.join(CSRC, "experiments"))]
    deps.append(os.path.join(HERE, "..", "include", "specgpu.h"))
    return max(os.path.getmtime(d) for d in deps)


# ---- ISA lint run on EVERY translation unit of EVERY library this module builds -------------------------------
# A vector-ALU write to a data register of a buffer store wider than 8 bytes within the next two instruction slots (s_nop N
# counts N + 1) reaches the store on gfx950 when the store has a scalar offset register: measured in spec_k_v3h.hip (round 4;
# the last lanes of a wave left with the restored value, one run in three), not padded by LLVM's hazard recogniser, which
# exempts stores with an SGPR offset.  The measurement applies to the toolchain named in the library's .flags stamp
# (ROCm 7.2.0's hipcc); tools/check_store_hazard.py is the command-line form, tests/test_store_hazard.py the mutation test.
_STORE_WAIT = 2


def _vregs(tok):
    import re
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def store_hazard_findings(path: str):
    """Findings (one line each) in the device assembly `path` (hipcc -S / -save-temps)."""
    import re
    findings, kern, ins = [], None, []
    for line in open(path):
        s = line.strip()
        if line.startswith("_Z") and ":" in line:
            kern = line.split(":")[0]
        if not kern or not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        ins.append((kern, s))
    for i, (k, s) in enumerate(ins):
        if not re.match(r"buffer_store_(dwordx[34]|format_xyzw?)\b", s):
            continue
        data = _vregs(re.split(r"[ ,]+", s)[1])
        slots, j = 0, i + 1
        while slots < _STORE_WAIT and j < len(ins) and ins[j][0] == k:
            toks = re.split(r"[ ,]+", ins[j][1])
            if toks[0] == "s_nop":
                slots += int(toks[1], 0) + 1
            else:
                if toks[0].startswith("v_") and not toks[0].startswith("v_cmp") and len(toks) > 1 and _vregs(toks[1]) & data:
                    findings.append("%s: `%s` then, %d slot(s) later, `%s`" % (k[:80], s, slots + 1, ins[j][1]))
                slots += 1
            j += 1
    return findings


def waterfall_findings(path: str):
    """Waterfall loops in the device assembly `path`: a memory instruction whose scalar operand the compiler could not prove
    uniform is wrapped in `v_readfirstlane / v_cmp_eq / s_and_saveexec / <access> / s_xor exec / s_cbranch_execnz` -- three vector
    and five scalar instructions and a branch per access.  Rounds 1-5 shipped the headline kernel with eight of them per line
    (the ping-pong line loop looked divergent to hipcc: spec_v2.h, `iters`); no kernel of the product needs one, so the product
    build fails on a finding (experiment variants only report)."""
    import re
    lines, kern = [], None
    for line in open(path):
        if line.startswith("_Z") and ":" in line:
            kern = line.split(":")[0]
        lines.append((kern, line.strip()))
    label_at = {}
    for i, (k, s) in enumerate(lines):
        m = re.match(r"(\.LBB\d+_\d+):", s)
        if m:
            label_at[(k, m.group(1))] = i
    count = {}
    for i, (k, s) in enumerate(lines):
        m = re.match(r"s_cbranch_execnz\s+(\.LBB\d+_\d+)", s)
        if not m:
            continue
        j = label_at.get((k, m.group(1)))
        if j is None or j > i or i - j > 40:
            continue
        body = [t for _, t in lines[j:i]]
        if any(t.startswith("v_readfirstlane") for t in body) and any(re.match(r"(buffer|global|flat|ds)_", t) for t in body):
            count[k] = count.get(k, 0) + 1
    return ["%s: %d waterfall loop(s)" % (k[:100], n) for k, n in sorted(count.items())]


def kernel_spills(path: str):
    """{mangled kernel name: spilled vector registers} from the metadata of a device assembly file."""
    import re
    out, name = {}, None
    for line in open(path):
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.vgpr_spill_count:\s+(\d+)", line)
        if m and name:
            out[name] = int(m.group(1))
            name = None
    return out


# experiment variants whose flags concern the team kernel compile the frozen experiment copy of it (csrc/experiments/): the
# product file no longer carries that code
EXPERIMENT_SOURCE = {"spec_k_team.hip": os.path.join("experiments", "spec_k_team_exp.hip")}
_TEAM_EXPERIMENT_FLAGS = ("-DSPEC_TEAM_WA", "-DSPEC_ABL_TEAM", "-DSPEC_ABL_WA", "-DSPEC_TEAM_PROF", "-DSPEC_TEAM_STRICT_WAITS",
                          "-DSPEC_TEAM_SINGLE_STORES", "-DSPEC_TEAM_NO_STRIPX")


def _source_for(src: str, extra) -> str:
    if src in EXPERIMENT_SOURCE and any(f.startswith(_TEAM_EXPERIMENT_FLAGS) for f in extra):
        return EXPERIMENT_SOURCE[src]
    return src


def _compile(src: str, objdir: str, extra) -> str:
    """Compile one translation unit; the device assembly hipcc leaves beside the object (-save-temps: the SAME compilation,
    not a second one) goes through the ISA lint, and a finding fails the build."""
    stem = os.path.splitext(src)[0]
    obj = os.path.join(objdir, stem + ".o")
    real = _source_for(src, extra)
    if real != src:  # (same object name; the temporaries carry the real file's stem)
        stem = os.path.splitext(os.path.basename(real))[0]
    cmd = [_hipcc(), *FLAGS, *extra, "-I" + CSRC, "-save-temps=obj", "-c", os.path.join(CSRC, real), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))
    asm = os.path.join(objdir, "%s-hip-amdgcn-amd-amdhsa-%s.s" % (stem, ARCH))
    if not os.path.exists(asm):
        raise RuntimeError("no device assembly for %s (expected %s): the ISA lint cannot run" % (src, asm))
    bad = store_hazard_findings(asm)
    wf = waterfall_findings(asm)
    spills = kernel_spills(asm)
    for f in os.listdir(objdir):  # the other temporaries (bitcode, preprocessed source: hundreds of MB over all units)
        if f.startswith(stem + "-") or f.startswith(stem + ".hip-"):
            os.remove(os.path.join(objdir, f))   # (the device assembly too, once linted below: 120 MB over all units)
    if bad:
        raise RuntimeError("ISA lint (store-data hazard, build.py store_hazard_findings) on %s:\n%s" % (src, "\n".join(bad[:20])))
    if wf and not extra:
        raise RuntimeError("ISA lint (waterfall loops, build.py waterfall_findings) on %s:\n%s" % (src, "\n".join(wf[:20])))
    if wf:
        print("note: %s (%s): %s" % (src, " ".join(extra), "; ".join(wf[:4])))
    with open(os.path.splitext(obj)[0] + ".spills", "w") as f:  # which kernels of this unit spill (tests/test_abi.py reads the product's)
        for k, n in sorted(spills.items()):
            f.write("%s %d\n" % (k, n))
    return obj


_TOOLCHAIN = None


def _toolchain() -> str:
    """First line of `hipcc --version` (the measurement behind the ISA lint applies to it)."""
    global _TOOLCHAIN
    if _TOOLCHAIN is None:
        try:
            out = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout.strip().splitlines()
            _TOOLCHAIN = next((l.strip() for l in out if "HIP version" in l), out[0].strip() if out else "unknown")
        except Exception:  # noqa: BLE001
            _TOOLCHAIN = "unknown"
    return _TOOLCHAIN


def _stamp(extra) -> str:
    """What a library was built with: the product flags plus any experiment flags, and the toolchain."""
    return " ".join([*FLAGS, *extra]) + " | " + _toolchain() + " | isa-lint: store hazard, waterfall loops"


# named variants: what they are compiled with, and which translation units the flags touch (the others are taken from
# the product build as they are)
def _unique(pairs):
    """dict(pairs), refusing a name defined twice (round 4 lost a variant that way)"""
    out = {}
    for name, spec in pairs:
        if name in out:
            raise ValueError("build variant %r is defined twice" % name)
        out[name] = spec
    return out


VARIANTS = _unique([
    # the team kernel's two experiment geometries (large_wg = 256 / 1024): measured slower, kept
    # under test (tests/test_gpu_large.py), not carried by the product library
    ("teamvar", (["-DSPEC_TEAM_VARIANTS"], ["spec_k_team.hip"])),
    # development aid: lane-0 cycle counters inside the team kernel (tools/team_prof.py)
    ("tPROF", (["-DSPEC_TEAM_PROF"], ["spec_k_team.hip", "spec_capi.hip"])),
    # experiments on the single-workgroup 32768-point kernel: how many cf32 samples of the next line are requested
    # beside the second transform (tools/bench_v2h.py)
    # round-4 experiment: wave-autonomous sides of the team kernel (fp64 lines)
    ("tWA", (["-DSPEC_TEAM_WA"], ["spec_k_team.hip"])),
    ("tWAnt", (["-DSPEC_TEAM_WA", "-DSPEC_TEAM_WA_LD=0"], ["spec_k_team.hip"])),
    # ablations of it (results wrong by construction): column side alone; without its slot stores; without its arithmetic
    ("tWAa", (["-DSPEC_TEAM_WA", "-DSPEC_ABL_TEAM_NOB", "-DSPEC_ABL_TEAM_NOWAIT"], ["spec_k_team.hip"])),
    ("tWAb", (["-DSPEC_TEAM_WA", "-DSPEC_ABL_TEAM_NOB", "-DSPEC_ABL_TEAM_NOWAIT", "-DSPEC_ABL_TEAM_NOSLOT"], ["spec_k_team.hip"])),
    ("tWAnf", (["-DSPEC_TEAM_WA", "-DSPEC_ABL_TEAM_NOB", "-DSPEC_ABL_TEAM_NOWAIT", "-DSPEC_ABL_TEAM_NOSLOT", "-DSPEC_ABL_TEAM_NOFFT"], ["spec_k_team.hip"])),
    ("tWAs", (["-DSPEC_TEAM_WA", "-DSPEC_TEAM_WA_DEEP=0"], ["spec_k_team.hip"])),
    ("tWA1", (["-DSPEC_TEAM_WA", "-DSPEC_TEAM_WA_ANN=1"], ["spec_k_team.hip"])),
    ("tWA2", (["-DSPEC_TEAM_WA", "-DSPEC_TEAM_WA_ANN=2"], ["spec_k_team.hip"])),
    ("tWA1a", (["-DSPEC_TEAM_WA", "-DSPEC_TEAM_WA_ANN=1", "-DSPEC_ABL_TEAM_NOB", "-DSPEC_ABL_TEAM_NOWAIT"], ["spec_k_team.hip"])),
    ("tWA2a", (["-DSPEC_TEAM_WA", "-DSPEC_TEAM_WA_ANN=2", "-DSPEC_ABL_TEAM_NOB", "-DSPEC_ABL_TEAM_NOWAIT"], ["spec_k_team.hip"])),
    ("tWAca", (["-DSPEC_TEAM_WA", "-DSPEC_TEAM_WA_DEEP=0", "-DSPEC_ABL_WA_COALESCED", "-DSPEC_ABL_TEAM_NOB", "-DSPEC_ABL_TEAM_NOWAIT"], ["spec_k_team.hip"])),
    ("tWAcb", (["-DSPEC_TEAM_WA", "-DSPEC_TEAM_WA_DEEP=0", "-DSPEC_ABL_WA_COALESCED", "-DSPEC_ABL_TEAM_NOB", "-DSPEC_ABL_TEAM_NOWAIT", "-DSPEC_ABL_TEAM_NOSLOT"], ["spec_k_team.hip"])),
    ("tWAsa", (["-DSPEC_TEAM_WA", "-DSPEC_TEAM_WA_DEEP=0", "-DSPEC_ABL_TEAM_NOB", "-DSPEC_ABL_TEAM_NOWAIT"], ["spec_k_team.hip"])),
    ("tWAc", (["-DSPEC_TEAM_WA", "-DSPEC_TEAM_WA_DEEP=0", "-DSPEC_ABL_WA_COALESCED"], ["spec_k_team.hip"])),
    ("tOLDa", (["-DSPEC_ABL_TEAM_NOB", "-DSPEC_ABL_TEAM_NOWAIT"], ["spec_k_team.hip"])),
    ("v2hpf0", (["-DV2H_PF=0"], ["spec_k_v2h.hip"])),
    ("v2hpf8", (["-DV2H_PF=8"], ["spec_k_v2h.hip"])),
    ("v2hpf20", (["-DV2H_PF=20"], ["spec_k_v2h.hip"])),
    ("v2hw4", (["-DV2H_WAVES_12=4"], ["spec_k_v2h.hip"])),
    ("v2hw2", (["-DV2H_WAVES_12=2"], ["spec_k_v2h.hip"])),
    # fp64 family: last-pass twiddles in registers (rounds 1-3) / in LDS without the cf64 / cf32 prefetch
    ("v3dreg", (["-DSPEC_V3D_LDS_TWL=0"], ["spec_k_v3d.hip"])),
    ("v3dnp", (["-DSPEC_V3D_PREFETCH_ALL=0"], ["spec_k_v3d.hip"])),
    # 32-point-per-thread plans: the window re-read from the L2-resident table every line (rounds 1-3) instead of a quarter Hann table in LDS
    ("v2wg", (["-DSPEC_V2_WIN_LDS=0"], ["spec_k_v2s.hip", "spec_k_v2w.hip"])),
    # round 5, A/B: the write-after-read barrier of an exchange in front of its stores (rounds 1-4) instead of right behind the
    # previous exchange's loads
    ("v2late", (["-DSPEC_V2_LATE_WAR=1"], ["spec_k_v2s.hip", "spec_k_v2w.hip", "spec_k_v2r.hip", "spec_k_v2h.hip"])),
    # round 5 experiment: 16384-point Welch segments through the plan 16 x (32 x 32) ("welch_rows" = 1; csrc/experiments/spec_v2_exp.h)
    ("v2rows", (["-DSPEC_V2_ROWS"], ["spec_k_v2w.hip", "spec_capi.hip"])),
    # ablations of the paired 65536-point kernel (results wrong by construction): every line re-reads the workgroup's first
    # (the lines come from L1 / L2); no output stores; both
    ("v2qnl", (["-DV2Q_ABL_NOLOAD"], ["spec_k_v2q.hip"])),
    ("v2qns", (["-DV2Q_ABL_NOSTORE"], ["spec_k_v2q.hip"])),
    ("v2qnn", (["-DV2Q_ABL_NOLOAD", "-DV2Q_ABL_NOSTORE"], ["spec_k_v2q.hip"])),
    ("v2qnt", (["-DV2Q_ST_AUX=2"], ["spec_k_v2q.hip"])),
    ("v2qst0", (["-DV2Q_ST_AUX=0"], ["spec_k_v2q.hip"])),
    ("v2qst1", (["-DV2Q_ST_AUX=1"], ["spec_k_v2q.hip"])),
    ("v2qst3", (["-DV2Q_ST_AUX=3"], ["spec_k_v2q.hip"])),
    ("v2qst16", (["-DV2Q_ST_AUX=16"], ["spec_k_v2q.hip"])),
    ("v2qst18", (["-DV2Q_ST_AUX=18"], ["spec_k_v2q.hip"])),
    ("v2qld16", (["-DV2Q_LD_AUX=16"], ["spec_k_v2q.hip"])),
    ("v2qm16", (["-DV2Q_ST_AUX=2", "-DV2Q_M0=1", "-DV2Q_M1=12", "-DV2Q_M2=16"], ["spec_k_v2q.hip"])),
    ("v2qm12", (["-DV2Q_ST_AUX=2", "-DV2Q_M0=1", "-DV2Q_M1=8", "-DV2Q_M2=12"], ["spec_k_v2q.hip"])),
    ("v2qm22", (["-DV2Q_ST_AUX=2", "-DV2Q_M0=2", "-DV2Q_M1=12", "-DV2Q_M2=22"], ["spec_k_v2q.hip"])),
    ("v2qlnt", (["-DV2Q_LD_AUX=2"], ["spec_k_v2q.hip"])),
    ("v2qroll", (["-DV2Q_ROLLING=1"], ["spec_k_v2q.hip"])),  # the rolling request schedule for every variant (product: with the Hann window only)
    ("v2qnoroll", (["-DV2Q_ROLLING=0"], ["spec_k_v2q.hip"])),
    # the compiler's own scheduling strategies on the packed family (round 5; -mllvm options of this toolchain)
    ("v2sr", (["-DSPEC_V2_SINGLE_READS=1"], ["spec_k_v2s.hip", "spec_k_v2w.hip", "spec_k_v2h.hip", "spec_k_v2q.hip", "spec_k_v2n.hip", "spec_k_v2r.hip"])),
    ("schilp", (["-mllvm", "-amdgpu-sched-strategy=max-ilp"], ["spec_k_v2s.hip", "spec_k_v2w.hip", "spec_k_v2h.hip"])),
    ("schb0", (["-mllvm", "-amdgpu-schedule-metric-bias=0"], ["spec_k_v2s.hip", "spec_k_v2w.hip", "spec_k_v2h.hip"])),
    ("schtrk", (["-mllvm", "-amdgpu-use-amdgpu-trackers"], ["spec_k_v2s.hip", "spec_k_v2w.hip", "spec_k_v2h.hip"])),
    # round 5, A/B: |X|^2 left to the compiler (its SLP vectorizer pairs two bins per v_pk_mul / v_pk_fma behind four v_mov)
    # instead of the two spelled-out instructions per bin of spec_fft_pk.h
    ("slpnorm", (["-DSPEC_PK_NORM_ASM=0"], ["spec_k_v2s.hip", "spec_k_v2w.hip", "spec_k_v2r.hip", "spec_k_v2h.hip", "spec_k_v2q.hip", "spec_k_v2n.hip"])),
    # round 5, A/B: the line loop of the whole-workgroup kernels with the trip count left in a VGPR (rounds 1-5): hipcc wraps every
    # load of the cf32 ping-pong kernels (2048 ... 16384 points, 50 % overlap, no window) in a waterfall loop
    ("v2wfall", (["-DSPEC_V2_VECTOR_TRIP=1"], ["spec_k_v2s.hip", "spec_k_v2r.hip"])),
    # round 5, A/B: no wait for the first line's samples in front of the line loop (rounds 1-5): hipcc's wait counts at the loop
    # header then cover the previous line's output stores as well
    ("v2nowait", (["-DSPEC_V2_ENTRY_WAIT=0"], ["spec_k_v2s.hip", "spec_k_v2r.hip"])),
    # development aid: per-wave shader-clock stamps at the phase boundaries of the Welch kernels (tools/v2_timeline.py)
    ("v2stamp", (["-DSPEC_V2_STAMPS", "-DSPEC_V2_ROWS"], ["spec_k_v2w.hip", "spec_capi.hip"])),
    ("v3hhi", (["-DV3H_EARLY_LO_FIRST=0"], ["spec_k_v3h.hip"])),
    ("v3he48", (["-DV3H_EARLY_REGS=48"], ["spec_k_v3h.hip"])),
    ("v3he64", (["-DV3H_EARLY_REGS=64"], ["spec_k_v3h.hip"])),
])


STREAM_PROBE = os.path.join(LIBDIR, "membench_rw")


def build_tools(verbose: bool = False) -> str:
    """tools/membench_rw.hip -> lib/membench_rw: the stand-alone stream probe bench.py runs as a child process (the box's own
    ceiling for a workload's read : write mix, roofline.stream).  A measurement aid, not part of the library."""
    src = os.path.join(HERE, "..", "tools", "membench_rw.hip")
    if os.path.exists(STREAM_PROBE) and os.path.getmtime(STREAM_PROBE) >= os.path.getmtime(src):
        return STREAM_PROBE
    os.makedirs(LIBDIR, exist_ok=True)
    r = subprocess.run([_hipcc(), "-O3", "--offload-arch=" + ARCH, src, "-o", STREAM_PROBE], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on tools/membench_rw.hip:\n%s" % r.stderr[-2000:])
    if verbose:
        print("built", STREAM_PROBE)
    return STREAM_PROBE


def build(force: bool = False, verbose: bool = False, variant: str = "", extra_flags=(), only=None) -> str:
    """Build the product library, or -- ``variant`` -- an EXPERIMENT library beside it
    (lib/libspecgpu_<variant>.so, objects under build/<variant>/, compiled with ``extra_flags``, e.g. the
    SPEC_ABL_* ablation macros whose results are wrong by construction).  An experiment never overwrites the
    product: the product build takes no extra flags, and its freshness test also compares the flag stamp
    stored beside the library, so a library built any other way is rebuilt.  ``only``: the translation units the
    flags affect -- the rest is linked from the product's objects (built first when stale)."""
    if variant in VARIANTS and not extra_flags:
        extra_flags, only = VARIANTS[variant]
    extra = list(extra_flags)
    if variant:
        lib = os.path.join(LIBDIR, "libspecgpu_%s.so" % variant)
        objdir = os.path.join(OBJDIR, variant)
    else:
        if extra:
            raise ValueError("the product library takes no extra flags; name a variant")
        lib, objdir = LIB, OBJDIR
    stamp_path = lib + ".flags"
    fresh = (os.path.exists(lib) and os.path.getmtime(lib) >= _newest_dep() and os.path.exists(stamp_path)
             and open(stamp_path).read() == _stamp(extra))
    if fresh and not force:
        return lib
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    if variant and only:
        build(force=False, verbose=verbose)  # the product's objects, fresh
        with cf.ThreadPoolExecutor(max_workers=min(8, len(only))) as ex:
            mine = dict(zip(only, ex.map(lambda src: _compile(src, objdir, extra), only)))
        objs = [mine.get(src) or os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o") for src in SOURCES]
        missing = [o for o in objs if not os.path.exists(o)]
        if missing:  # the product library was built elsewhere (objects do not travel): compile everything
            with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
                objs = list(ex.map(lambda src: _compile(src, objdir, extra if src in only else []), SOURCES))
    else:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
            objs = list(ex.map(lambda src: _compile(src, objdir, extra), SOURCES))
    cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", lib, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    with open(stamp_path, "w") as f:
        f.write(_stamp(extra))
    with open(lib + ".spills", "w") as f:  # kernel -> spilled vector registers, every unit of this library
        for o in objs:
            sp = os.path.splitext(o)[0] + ".spills"
            if os.path.exists(sp):
                f.write(open(sp).read())
    if verbose:
        print("built", lib)
    return lib


def build_jni(verbose: bool = False):
    """Compile integration/jni/specgpu_jni.c when a JDK (jni.h) is present; returns the
    path or None.  The authoring container has no JDK, so this is normally skipped --
    loudly -- and the C ABI underneath is what tests and benchmarks drive."""
    root = os.path.dirname(HERE)
    jh = os.environ.get("JAVA_HOME", "")
    inc = os.path.join(jh, "include")
    if not jh or not os.path.exists(os.path.join(inc, "jni.h")):
        if verbose:
            print("JNI shim skipped: no JAVA_HOME/include/jni.h on this machine")
        return None
    out = os.path.join(LIBDIR, "libspecgpu_jni.so")
    cmd = ["gcc", "-shared", "-fPIC", "-O2", "-I" + inc, "-I" + os.path.join(inc, "linux"),
           "-I" + os.path.join(root, "include"), os.path.join(root, "integration", "jni", "specgpu_jni.c"),
           "-L" + LIBDIR, "-lspecgpu", "-Wl,-rpath,$ORIGIN", "-o", out]
    subprocess.check_call(cmd)
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    # python -m spectral_analyzer_amd.build [--force] [--variant NAME -- -DSPEC_ABL_X ...]
    if "--variant" in sys.argv:  # a named variant (VARIANTS) needs no flags
        i = sys.argv.index("--variant")
        flags = sys.argv[sys.argv.index("--") + 1:] if "--" in sys.argv else []
        build(force=True, verbose=True, variant=sys.argv[i + 1], extra_flags=flags)
        sys.exit(0)
    build(force="--force" in sys.argv, verbose=True)
    if "--jni" in sys.argv:
        build_jni(verbose=True)
