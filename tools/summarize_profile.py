#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag> (tools/profile.sh) into the tracked summaries under profiles/.

    python tools/summarize_profile.py <tag> <round-name> [workload]
writes profiles/<round>_kernel_stats.csv, profiles/<round>_pmc.json, profiles/<round>_summary.md and
updates profiles/pmc_traffic.json (HBM bytes per launch, read by bench.py for roofline.traffic).
"""
import collections, csv, glob, json, os, shutil, sys

tag, rnd = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, rnd + "_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
main = max(rows, key=lambda r: float(r["TotalDurationNs"]) if not any(x in r["Name"] for x in ("synth", "copyBuffer", "fill")) else 0)
# timed launches only: the last `steps` dispatches of the dominant kernel in the trace
trace = glob.glob(os.path.join(src, "stats", "*", "*_kernel_trace.csv"))[0]
durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(trace))
        if r["Kernel_Name"] == main["Name"]]
bj = os.path.join(src, "bench_under_prof.json")
steps = json.load(open(bj))["steps"] if os.path.exists(bj) else len(durs)
timed = durs[-steps:]
timed_avg_ns = sum(timed) / len(timed)
pmc = {}
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].split("(")[0] in main["Name"] or main["Name"].split("(")[0] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k] = sum(v) / len(v)
# MI355X_MICROARCH.md, HBM: FETCH_SIZE is in KiB and reads exactly half the bytes of a wide
# coalesced stream on gfx950 -> x2; WRITE_SIZE (KiB) is exact for streaming stores.
fetch = pmc.get("FETCH_SIZE", 0) * 1024 * 2
write = pmc.get("WRITE_SIZE", 0) * 1024
traffic = fetch + write
out = {"kernel": main["Name"], "calls": int(main["Calls"]), "avg_ns": float(main["AverageNs"]),
       "min_ns": float(main["MinNs"]), "timed_launches": len(timed), "timed_avg_ns": timed_avg_ns,
       "pmc_mean_per_launch": pmc,
       "hbm_read_bytes_per_launch(FETCH_SIZE*1024*2)": fetch, "hbm_write_bytes_per_launch(WRITE_SIZE*1024)": write,
       "hbm_traffic_bytes_per_launch": traffic}
json.dump(out, open(os.path.join(dst, rnd + "_pmc.json"), "w"), indent=1)

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
t[workload] = {"bytes": traffic, **_stamp("profiles/%s_pmc.json" % rnd)}
json.dump(t, open(tp, "w"), indent=1)
with open(os.path.join(dst, rnd + "_summary.md"), "w") as f:
    f.write("# %s -- rocprofv3 summary (%s)\n\n" % (rnd, workload))
    f.write("Command: `tools/profile.sh %s` = `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline` "
            "(default steps/warmup, as the plain bench) plus separate `--pmc` passes with --steps 5 --warmup 2.\n\n" % tag)
    f.write("| kernel | calls | avg ms | min ms |\n|---|---|---|---|\n")
    for r in rows:
        f.write("| `%s` | %s | %.4f | %.4f |\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6))
    f.write("\nDominant kernel, the %d timed launches only (warm-up launches excluded): average %.4f ms\n"
            % (len(timed), timed_avg_ns / 1e6))
    f.write("\nPMC means per launch of the dominant kernel:\n\n")
    for k in sorted(pmc):
        f.write("* %s = %.4g\n" % (k, pmc[k]))
    if pmc.get("GRBM_GUI_ACTIVE") and timed_avg_ns:
        # the shader clock the counter pass ran at (GRBM cycles / kernel time of the stats pass; the two
        # passes are separate processes on the same box, so this is approximate -- bench.py's roofline.clocks is the direct reading)
        f.write("\nGRBM_GUI_ACTIVE / kernel time = %.0f MHz (shader clock during the counter pass, approximate)\n"
                % (pmc["GRBM_GUI_ACTIVE"] / 8 / (timed_avg_ns * 1e-9) / 1e6))   # (the counter is summed over the 8 XCDs)
    f.write("\nHBM traffic per launch: read %.3f GB (FETCH_SIZE KiB x 1024 x 2, gfx950 correction) + write %.3f GB "
            "= %.3f GB\n" % (fetch / 1e9, write / 1e9, traffic / 1e9))
    if os.path.exists(bj):
        f.write("\nbench.py line under the profiler (clocks are lower under rocprofv3):\n\n```\n%s```\n" % open(bj).read())
print(json.dumps({k: out[k] for k in ("kernel", "avg_ns", "hbm_traffic_bytes_per_launch")}))
