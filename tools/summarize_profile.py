#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/profile_config.sh into the committed summaries:

    profiles/<tag>_<config>_kernel_stats.csv   the --kernel-trace --stats table (all kernels of the bench command)
    profiles/<tag>_<config>_counters.json      per-dispatch means of every collected counter for the render kernel
    profiles/counters.json                     the per-config entry bench.py reads (updated in place)

usage: tools/summarize_profile.py <tag> <config> [<config> ...]     (run here, after gpurun merged gpurun_out/)
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def git_rev():
    """HEAD (+ "-dirty" when tracked files differ from it) of the tree the profile was taken from: the summaries are
    made right after the gpurun call, from the same working tree."""
    try:
        rev = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
        dirty = subprocess.call(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", "raytracing_amd/csrc", "include"]) != 0
        return rev + ("-dirty" if dirty else "")
    except (OSError, subprocess.CalledProcessError):
        return None


def main():
    tag, cfgs = sys.argv[1], sys.argv[2:]
    sys.path.insert(0, ROOT)
    import bench
    here_hash, rev = bench.source_hash(), git_rev()
    idx_path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        index = json.load(open(idx_path))
    except (OSError, ValueError):
        index = {"_about": "Per-launch rocprofv3 counters of the render kernel per bench config, N = 1 (tools/"
                           "profile_config.sh: SQ counters in two --pmc passes, FETCH_SIZE and WRITE_SIZE in a pass "
                           "each, all of `python3 bench.py --no-cpu-baseline --config X`).  FETCH_SIZE / WRITE_SIZE "
                           "in KiB as the counters report them; bench.py applies the gfx950 x2 correction to "
                           "FETCH_SIZE.  SQ_*_CYCLES and SQ_ACTIVE / SQ_WAIT counters are quad-cycles summed over "
                           "waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs."}
    for cfg in cfgs:
        d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{cfg}")
        stats = glob.glob(os.path.join(d, "**", "trace_kernel_stats.csv"), recursive=True)
        kern = None
        if stats:
            dst = os.path.join(ROOT, "profiles", f"{tag}_{cfg}_kernel_stats.csv")
            shutil.copy(stats[0], dst)
            rows = [r for r in csv.DictReader(open(stats[0])) if r["Name"].startswith("mrt::render") or "render_" in r["Name"]]
            rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
            if rows:
                kern = {"name": rows[0]["Name"], "calls": int(rows[0]["Calls"]), "avg_ns": float(rows[0]["AverageNs"]),
                        "min_ns": float(rows[0]["MinNs"]), "max_ns": float(rows[0]["MaxNs"])}
        bj = os.path.join(d, "bench_under_rocprof.json")
        if os.path.exists(bj):
            lines = [ln for ln in open(bj).read().splitlines() if ln.startswith("{")]
            if lines:
                open(os.path.join(ROOT, "profiles", f"{tag}_{cfg}_bench_under_rocprof.json"), "w").write(lines[-1] + "\n")
        counters, ndisp = {}, {}
        for f in glob.glob(os.path.join(d, "**", "pmc_*counter_collection.csv"), recursive=True):
            tot, n = {}, {}
            for r in csv.DictReader(open(f)):
                if kern and r["Kernel_Name"] != kern["name"]:
                    continue
                if not kern and "render_" not in r["Kernel_Name"]:
                    continue
                c = r["Counter_Name"]
                tot[c] = tot.get(c, 0.0) + float(r["Counter_Value"])
                n[c] = n.get(c, 0) + 1
            for c in tot:
                counters[c] = tot[c] / n[c]
                ndisp[c] = n[c]
        prov = {}
        try:
            prov = json.load(open(os.path.join(d, "provenance.json")))
        except (OSError, ValueError):
            pass
        if prov.get("source_hash") and prov["source_hash"] != here_hash:
            print(f"WARNING {cfg}: profiled sources {prov['source_hash']} != this tree {here_hash}: the summary is "
                  f"stamped with the PROFILED hash and bench.py will call it stale")
        entry = {"kernel": kern, "counters_per_dispatch_mean": counters, "dispatches": ndisp,
                 "how": "tools/profile_config.sh " + tag + " " + cfg,
                 "source_hash": prov.get("source_hash"), "git_rev": rev}
        out = os.path.join(ROOT, "profiles", f"{tag}_{cfg}_counters.json")
        json.dump(entry, open(out, "w"), indent=1)
        e = {"source": f"profiles/{tag}_{cfg}_counters.json", "source_hash": prov.get("source_hash"), "git_rev": rev}
        if kern:
            e["rocprof_kernel_avg_ns"] = kern["avg_ns"]
            e["kernel"] = kern["name"]
        for c, v in counters.items():
            e[c + "_KB" if c in ("FETCH_SIZE", "WRITE_SIZE") else c] = round(v, 2)
        index[cfg] = e
        print(cfg, json.dumps(e))
    json.dump(index, open(idx_path, "w"), indent=1)


if __name__ == "__main__":
    main()
