#!/usr/bin/env python3
"""Where a kernel's issue slots go, from the counters of tools/profile.sh (profiles/<round>_<cfg>_pmc.json): one table row
per workload, per LINE (Welch: per segment).

    python tools/issue_table.py <label>:<pmc.json>:<lines per launch> [...]  > profiles/rNN_issue_table.md

Two views of the same launch (MI355X_MICROARCH.md: SQ_* cycle counters count quad-cycles, i.e. x 4 shader cycles):
  * the SIMD's view -- of the cycles one SIMD has per line (kernel cycles x 1024 SIMDs / lines), how many it spent
    issuing vector-ALU, LDS and vector-memory instructions (SQ_ACTIVE_INST_{VALU,LDS,VMEM} x 4); the rest is idle:
    no wave of the SIMD had an instruction of that kind ready.  Issue cycles per instruction = ACTIVE / INSTS.
  * the wave's view -- SQ_WAVE_CYCLES split into issuing (SQ_ACTIVE_INST_ANY), parked at an s_waitcnt or s_barrier
    (SQ_WAIT_ANY) and stalled at issue (SQ_WAIT_INST_ANY; SQ_WAIT_INST_LDS is the part stalled on the LDS queue).
"""
import json
import sys


def main():
    rows = []
    for arg in sys.argv[1:]:
        label, path, lines = arg.rsplit(":", 2)
        d = json.load(open(path))
        p = d["pmc_mean_per_launch"]
        lines = float(lines)
        cyc = p["GRBM_GUI_ACTIVE"] / 8.0                     # shader cycles of the launch (counter summed over 8 XCDs)
        ghz = cyc / d["timed_avg_ns"] if d.get("timed_avg_ns") else float("nan")
        avail = cyc * 1024.0 / lines                         # SIMD-cycles per line
        valu, lds, vmem = (p[k] * 4.0 / lines for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"))
        wc = p["SQ_WAVE_CYCLES"]
        rows.append((label, __import__("re").search(r"(\w+<[^>]*>)", d["kernel"]).group(1), d["timed_avg_ns"] / 1e6, ghz, avail, valu, lds, vmem,
                     p["SQ_INSTS_VALU"] / lines, p["SQ_INSTS_LDS"] / lines, p["SQ_INSTS_VMEM"] / lines,
                     p["SQ_ACTIVE_INST_VALU"] * 4.0 / p["SQ_INSTS_VALU"], p["SQ_ACTIVE_INST_LDS"] * 4.0 / p["SQ_INSTS_LDS"],
                     p["SQ_ACTIVE_INST_ANY"] / wc, p["SQ_WAIT_ANY"] / wc, p["SQ_WAIT_INST_ANY"] / wc, p["SQ_WAIT_INST_LDS"] / wc,
                     p["SQ_WAVES"] , p.get("SQ_LDS_BANK_CONFLICT", 0) / max(p.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
    print("| workload | kernel | ms (profiled) | shader GHz | SIMD-cycles per line | VALU issue | LDS issue | VMEM issue | "
          "VALU / LDS / VMEM instructions per line (all waves) | cycles per VALU / LDS instruction | wave: issuing | parked (waitcnt, barrier) | "
          "stalled at issue (of which LDS queue) | LDS conflict cycles / active |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        (label, kern, ms, ghz, avail, valu, lds, vmem, iv, il, im, cv, cl, act, park, stall, stall_lds, waves, conf) = r
        print("| %s | `%s` | %.3f | %.2f | %.0f | %.0f = **%.2f** | %.0f = %.2f | %.0f = %.2f | %.0f / %.0f / %.0f | %.1f / %.1f | %.2f | %.2f | %.2f (%.2f) | %.2f |" % (
            label, kern, ms, ghz, avail, valu, valu / avail, lds, lds / avail, vmem, vmem / avail, iv, il, im, cv, cl, act, park, stall, stall_lds, conf))


if __name__ == "__main__":
    main()
