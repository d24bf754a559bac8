#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the small, committed summaries under
profiles/ (kernel-trace --stats table; PMC FETCH_SIZE / WRITE_SIZE per kernel, converted to
bytes per launch with the gfx950 corrections of MI355X_MICROARCH.md: FETCH_SIZE is in KiB
and counts a wide coalesced read stream at HALF its bytes -> x2; WRITE_SIZE is in KiB).

usage: tools/summarize_profiles.py <round-tag> <stats_dir> [<fetch_dir> <write_dir>] [--key N M] [--fkey N M]
       --key: size of the log-weights workload (k_fwd_partial / k_adj), --fkey: of the forces workload (k_strip);
       traffic.json also records the sha of the kernel sources the passes were taken with (bench.kernel_source_sha).
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void ", "").replace("bioen::", "")
    return name.split("(")[0]


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    out_dir = os.environ.get("PROFILES_OUT", os.path.join(ROOT, "profiles"))
    os.makedirs(out_dir, exist_ok=True)
    stats = glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)
    lines = []
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        lines.append("# rocprofv3 --kernel-trace --stats  (%s)\n" % tag)
        lines.append("| kernel | calls | total ms | avg us | % | min us | max us |")
        lines.append("|---|---|---|---|---|---|---|")
        for r in rows:
            lines.append("| %s | %s | %.3f | %.3f | %s | %.3f | %.3f |" % (
                short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                r["Percentage"], float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    traffic = {}
    if len(sys.argv) >= 5 and not sys.argv[3].startswith("--"):
        per = collections.defaultdict(dict)
        for d, ctr in ((sys.argv[3], "FETCH_SIZE"), (sys.argv[4], "WRITE_SIZE")):
            f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
            if not f:
                continue
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f[0])):
                if r["Counter_Name"] == ctr:
                    agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                per[k][ctr] = (sum(v) / len(v), len(v))
        lines.append("\n# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per launch\n")
        lines.append("| kernel | launches | FETCH_SIZE KiB (raw) | read bytes (x1024 x2) | WRITE_SIZE KiB | write bytes |")
        lines.append("|---|---|---|---|---|---|")
        for k in sorted(per, key=lambda k: -per[k].get("FETCH_SIZE", (0, 0))[0]):
            fs, n = per[k].get("FETCH_SIZE", (0.0, 0))
            ws, _ = per[k].get("WRITE_SIZE", (0.0, 0))
            lines.append("| %s | %d | %.1f | %.4e | %.1f | %.4e |" % (k, n, fs, fs * 1024 * 2, ws, ws * 1024))
            traffic[k] = (fs * 1024 * 2 + ws * 1024, n)
    with open(os.path.join(out_dir, "%s_rocprof_summary.md" % tag), "w") as fp:
        fp.write("\n".join(lines) + "\n")
    if "--key" in sys.argv and traffic:
        i = sys.argv.index("--key")
        N, M = int(sys.argv[i + 1]), int(sys.argv[i + 2])
        tpath = os.path.join(out_dir, "traffic.json")
        tj = json.load(open(tpath)) if os.path.isfile(tpath) else {}
        for base in ("k_fwd_partial", "k_adj", "k_strip_fwd", "k_strip_adj", "k_strip2"):     # (k_strip2: the one-copy adjoint at 512 < M <= 1024)      # launch-weighted mean over the batch-width variants
            sel = [v for k, v in traffic.items() if k.split("<")[0] == base]
            if sel:
                tj["%s_N%d_M%d" % (base, N, M)] = sum(b * n for b, n in sel) / max(sum(n for _, n in sel), 1)
        if "--fkey" in sys.argv:
            j = sys.argv.index("--fkey")
            FN, FM = int(sys.argv[j + 1]), int(sys.argv[j + 2])
            sel = [v for k, v in traffic.items() if k.split("<")[0] == "k_strip"]
            if sel:
                tj["k_strip_N%d_M%d" % (FN, FM)] = sum(b * n for b, n in sel) / max(sum(n for _, n in sel), 1)
        sys.path.insert(0, ROOT)
        import bench
        tj["_source_sha"] = bench.kernel_source_sha()
        tj["_note"] = ("HBM bytes per launch = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 from separate rocprofv3 --pmc "
                       "passes (gfx950: FETCH_SIZE reads 1/2 of a wide coalesced stream), see *_rocprof_summary.md")
        json.dump(tj, open(tpath, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
