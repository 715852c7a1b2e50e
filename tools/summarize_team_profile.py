#!/usr/bin/env python3
"""gpurun_out/prof_<tag> (tools/prof_team.sh) -> profiles/<round>_<workload>_{summary.md,kernel_stats.csv,pmc.json}
and profiles/pmc_traffic.json (HBM-side bytes per step, read by bench.py for roofline.traffic).

    python tools/summarize_team_profile.py <tag> <round> <workload> <kernel-name-substring>
"""
import collections, csv, glob, json, os, shutil, sys

tag, rnd, workload, needle = sys.argv[1:5]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
stats = max(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
shutil.copy(stats, os.path.join(dst, "%s_%s_kernel_stats.csv" % (rnd, workload)))
rows = list(csv.DictReader(open(stats)))
main = [r for r in rows if needle in r["Name"]][0]
trace = max(glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv")), key=os.path.getmtime)
durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(trace)) if r["Kernel_Name"] == main["Name"]]
bench = json.load(open(os.path.join(src, "bench_under_prof.json")))
timed = durs[-bench["steps"]:]
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
# one file per pass: gpurun merges a call's output into what earlier calls left in gpurun_out/, keep the newest
newest = {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    d = os.path.dirname(os.path.dirname(f))
    if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
        newest[d] = f
for f in newest.values():
    for r in csv.DictReader(open(f)):
        pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
step_fetch = step_write = 0.0
per_kernel = {}
for k, d in mean.items():
    # the pmc passes repeat the bench arguments, so they make steps + warmup launches of the step as well
    launches_per_step = len(next(iter(pmc[k].values()))) / float(bench["steps"] + bench["warmup"])
    f = d.get("FETCH_SIZE", 0) * 1024 * 2    # MI355X_MICROARCH.md: KiB, and x2 on gfx950 for wide streaming reads
    w = d.get("WRITE_SIZE", 0) * 1024
    per_kernel[k] = {"launches_per_step": launches_per_step, "read_bytes_per_launch": f, "write_bytes_per_launch": w, "pmc": d}
    if "synth" in k or "rocclr" in k:
        continue
    step_fetch += f * launches_per_step
    step_write += w * launches_per_step
out = {"kernel": main["Name"], "calls": int(main["Calls"]), "avg_ns": float(main["AverageNs"]),
       "timed_launches": len(timed), "timed_avg_ns": sum(timed) / len(timed),
       "hbm_side_read_bytes_per_step(FETCH_SIZE*1024*2)": step_fetch, "hbm_side_write_bytes_per_step(WRITE_SIZE*1024)": step_write,
       "hbm_side_traffic_bytes_per_step": step_fetch + step_write, "kernels": per_kernel, "bench_line_under_profiler": bench}
json.dump(out, open(os.path.join(dst, "%s_%s_pmc.json" % (rnd, workload)), "w"), indent=1)

def _stamp(profile):
    """Where a traffic figure comes from: the commit of the tree the profile was taken on (HEAD when the summary is
    written -- summarise before committing further kernel changes), the summary file, the date."""
    import datetime, subprocess
    try:
        commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
        dirty = subprocess.call(["git", "-C", root, "diff", "--quiet", "--", "spectral_analyzer_amd/csrc"]) != 0
    except Exception:
        commit, dirty = "unknown", False
    return {"commit": commit + ("+uncommitted csrc changes" if dirty else ""), "profile": profile,
            "date": datetime.date.today().isoformat()}

tp = os.path.join(dst, "pmc_traffic.json")
t = json.load(open(tp)) if os.path.exists(tp) else {}
t[workload] = {"bytes": step_fetch + step_write, **_stamp("profiles/%s_%s_pmc.json" % (rnd, workload))}
json.dump(t, open(tp, "w"), indent=1)
d = mean[main["Name"]]
with open(os.path.join(dst, "%s_%s_summary.md" % (rnd, workload)), "w") as f:
    f.write("# %s %s -- rocprofv3 summary\n\n" % (rnd, workload))
    f.write("Command: `tools/prof_team.sh %s --workload %s --steps %d --warmup %d` = `rocprofv3 --kernel-trace --stats -- python3 "
            "bench.py --no-cpu-baseline ...` plus separate `--pmc` passes (--steps 3 --warmup 1).\n\n" % (tag, workload, bench["steps"], bench["warmup"]))
    f.write("| kernel | calls | avg ms | % of GPU time |\n|---|---|---|---|\n")
    for r in rows[:8]:
        f.write("| `%s` | %s | %.4f | %s |\n" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
    f.write("\nDominant kernel, the %d timed launches only: average %.4f ms (bench.py's HIP-event time under the profiler: %.4f ms)\n"
            % (len(timed), sum(timed) / len(timed) / 1e6, bench["roofline"]["kernel_ms"]))
    f.write("\nPMC means per launch of the dominant kernel:\n\n")
    for k in sorted(d):
        f.write("* %s = %.5g\n" % (k, d[k]))
    f.write("\nMemory-side traffic per step (all kernels of the step): read %.3f GB (FETCH_SIZE KiB x 1024 x 2, gfx950 correction) + "
            "write %.3f GB = %.3f GB\n" % (step_fetch / 1e9, step_write / 1e9, (step_fetch + step_write) / 1e9))
    if "TCC_HIT_sum" in d:
        f.write("\nL2 hit rate: %.3f\n" % (d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])))
    if "SQ_ACTIVE_INST_VALU" in d and "GRBM_GUI_ACTIVE" in d:
        cyc = d["GRBM_GUI_ACTIVE"] / 8
        f.write("VALU busy (SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)): %.3f; LDS busy: %.3f; shader clock %.2f GHz\n"
                % (d["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, d.get("SQ_ACTIVE_INST_LDS", 0) * 4 / 1024 / cyc, cyc / (sum(timed) / len(timed))))
print(json.dumps({"kernel": out["kernel"][:60], "timed_avg_ms": out["timed_avg_ns"] / 1e6, "traffic_GB": (step_fetch + step_write) / 1e9}))
