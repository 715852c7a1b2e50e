"""The persistent large-N kernel (csrc/spec_k_team.hip) keeps loads in flight across loop iterations with
inline-assembly loads and counted waits, and hands tiles between workgroups through the XCD's L2 with relaxed
counters.  Both are only sound under conditions that can be read off the ISA; the two lints under tools/ compile
the file for gfx950 (hipcc cross-compiles without a GPU; the experiment geometries of the variant library included)
and scan every instantiation:

  tools/check_inflight.py      no register is the destination of a load that is still in flight (the pipelined loads
                               are LDS-DMA: no register destination at all)
  tools/check_team_handoff.py  R1-R3: the N youngest operations in front of a hand-written vmcnt(N) are stream traffic
                               on every control-flow path (never a slot store, slot read or counter poll); no counter
                               add is reachable from a slot store without a vmcnt wait in between; slot reads and polls
                               carry sc1, recording reads nt, slot stores are plain, output stores nt
The mutation tests below show that the second lint does fail when either side is reordered."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def team_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("team_isa") / "team.s")
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-DSPEC_TEAM_VARIANTS",
                           "--cuda-device-only", "-S", os.path.join(ROOT, "spectral_analyzer_amd", "csrc", "spec_k_team.hip"),
                           "-o", out], stderr=subprocess.DEVNULL, timeout=900)
    return out


def test_pipelined_loads_are_lds_dma_and_nothing_touches_a_register_in_flight(team_asm):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight.py"), team_asm], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "in-flight registers touched 0" in r.stdout


def test_hand_off_contract_holds_in_the_isa_of_every_instantiation(team_asm):
    import check_team_handoff as h
    counts, report = h.check(open(team_asm).read())
    assert not report, "\n".join(report[:20])
    # every instantiation has both roles' counted waits, its counter adds and its slot stores under the lint
    assert counts["kernels"] >= 20 and counts["counted_waits"] >= 2 * 10 and counts["adds"] >= counts["kernels"]
    assert counts["slot_stores"] >= 8 * counts["kernels"] and counts["dma"] > 0


def _product_cfg5_kernel(text):
    """(start, end) line indices of large_team_kernel<double, 8, 8, true, true, 512, false> in the assembly text"""
    lines = text.split("\n")
    a = next(i for i, t in enumerate(lines) if re.match(r"_ZN7specgpu\S*large_team_kernelIdLi8ELi8ELb1ELb1ELi512ELb0E\S*:", t.strip()))
    b = next(i for i in range(a, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    return lines, a, b


def test_the_lint_fails_when_the_hand_off_is_reordered(team_asm):
    """Mutations of the cfg5 kernel's assembly that a compiler upgrade could produce, each of which must be reported."""
    import check_team_handoff as h
    text = open(team_asm).read()
    lines, a, b = _product_cfg5_kernel(text)
    body = lines[a:b]

    def rebuilt(new_body):
        return "\n".join(lines[:a] + new_body + lines[b:])

    def plain_store(t):
        return t.strip().startswith("global_store_dwordx4") and " nt" not in t and " sc" not in t

    # (1) a slot read loses its sc1: could be served by a stale line of the CU's L1
    k = next(i for i, t in enumerate(body) if t.strip().startswith("global_load_lds_dwordx4") and t.rstrip().endswith("sc1"))
    m = list(body); m[k] = m[k].replace(" sc1", "")
    assert any(r.startswith("R3") for r in h.check(rebuilt(m))[1])
    # (2) the column side's input requests move in front of the line's slot stores: the counted wait then leaves slot
    #     stores in flight when the line is announced
    last = max(i for i, t in enumerate(body) if plain_store(t))
    nt_after = [i for i in range(last, len(body)) if body[i].strip().startswith("global_load_lds_dwordx4") and body[i].rstrip().endswith("nt")]
    assert len(nt_after) >= 4
    m = [t for i, t in enumerate(body) if i not in set(nt_after)]
    rep = h.check(rebuilt(m))[1]
    assert any(r.startswith("R1") and "global_store_dwordx4" in r for r in rep)
    # (3) the hand-written waits disappear: a counter add is reachable from the slot stores without any wait
    m = [t for t in body if not re.match(r"\s*s_waitcnt vmcnt\(\d+\)\s*$", t)]
    assert any(r.startswith("R2") for r in h.check(rebuilt(m))[1])
    # (4) the row side's poll is issued behind the output stores: it is among the youngest at the counted wait
    first_out = next(i for i, t in enumerate(body) if t.strip().startswith("global_store_dwordx4") and t.rstrip().endswith("nt"))
    poll = next(i for i in range(first_out, 0, -1) if body[i].strip().startswith("global_load_lds_dword ") and body[i].rstrip().endswith("sc1"))
    run = [i for i in range(first_out, len(body)) if body[i].strip().startswith("global_store_dwordx4") and body[i].rstrip().endswith("nt")][:4]
    m = list(body)
    moved = m.pop(poll)
    m.insert(run[-1], moved)          # (indices behind `poll` shifted down by one: this lands behind the 4th store)
    assert any(r.startswith("R1") and "global_load_lds_dword " in r for r in h.check(rebuilt(m))[1])
