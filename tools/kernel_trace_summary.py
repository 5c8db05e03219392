"""Summarise a `rocprofv3 --kernel-trace --stats --output-format csv` run into the two artefacts kept under profiles/:
the per-kernel stats table (top rows) and the per-grid-size launch durations of the GAE scan kernel.

    python tools/kernel_trace_summary.py <rocprof output dir> <out stats csv> <out gae json> "<command that was profiled>"
"""
import csv
import glob
import json
import os
import re
import statistics
import sys


def find(d, pattern):
    hits = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    if not hits:
        raise SystemExit(f"no {pattern} under {d}")
    return max(hits, key=os.path.getsize)


def main():
    d, out_stats, out_gae, command = sys.argv[1:5]
    with open(find(d, "*kernel_stats.csv")) as f:
        rows = list(csv.reader(f))
    with open(out_stats, "w") as f:
        f.write(f"# {command}\n")
        w = csv.writer(f)
        for r in rows[:25]:
            w.writerow([c[:120] for c in r])
    per = {}
    with open(find(d, "*kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name", "")
            if "gae_kernel" not in name:
                continue
            grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1)
            m = re.search(r"gae_kernel<[^>]*>", name)
            per.setdefault((m.group(0) if m else "gae_kernel", grid), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {"command": command, "per_kernel_and_grid_size": {}}
    for (name, grid), v in sorted(per.items()):
        out["per_kernel_and_grid_size"][f"{name} grid={grid}"] = {
            "launches": len(v), "avg_us": round(sum(v) / len(v), 3), "median_us": round(statistics.median(v), 3),
            "min_us": round(min(v), 3), "max_us": round(max(v), 3)}
    out["note"] = ("grid = threads launched; the 4096 x 128 roofline launch of bench.py is 100 back-to-back launches per timed epoch "
                   "(gae_kernel<4,32,..,true>: reward and cost scans in separate lane groups, 2 x 4096 x 32 lanes = 262144 threads)")
    with open(out_gae, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["per_kernel_and_grid_size"], indent=1))


if __name__ == "__main__":
    main()
