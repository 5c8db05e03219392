"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, CSV output) of tools/gae_modes.py into
profiles/r02/gae_pmc.json: HBM bytes per launch of the GAE scan, per kernel instantiation and size, against the
algorithmic bytes.  Units and gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are
in KiB; FETCH_SIZE reports half of the bytes of a wide coalesced streaming read on gfx950 (doubled here); WRITE_SIZE as is."""
import collections
import csv
import json
import re
import sys

fetch_csv, write_csv, out = sys.argv[1], sys.argv[2], sys.argv[3]


def per_kernel(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "gae_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            m = re.search(r"gae_kernel<([^>]*)>", r["Kernel_Name"])
            d[(m.group(1) if m else "?", int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on tools/gae_modes.py",
       "corrections": "KiB -> bytes (x1024); FETCH_SIZE x2 (gfx950 wide-coalesced-read undercount); WRITE_SIZE x1",
       "template_args": "gae_kernel<VEC, LPR, BOOT, RC, F>: BOOT 0 = bootstrap arrays (predicated loads), 2 = folded; RC = reward/cost in "
                        "separate lane groups (cache-resident sizes)",
       "launches": []}
for (targs, grid) in sorted(f):
    rc = targs.replace(" ", "").split(",")[3] == "true"       # <VEC, LPR, BOOT, RC[, F]>
    folded = targs.replace(" ", "").split(",")[2] == "2"
    n_envs = grid // 64 if rc else grid // 32          # 256 threads per block; 4 rows per block with RC, 8 without (T = 128)
    algo = 33.0 * n_envs * 128 + (0.0 if folded else 8.0 * 2 * n_envs)
    fb, wb = f[(targs, grid)] * 1024 * 2, w.get((targs, grid), 0.0) * 1024
    res["launches"].append({"kernel": f"gae_kernel<{targs}>", "num_envs": n_envs, "folded": folded, "grid": grid,
                            "fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                            "algorithmic_bytes": algo, "traffic_over_algorithmic": round((fb + wb) / algo, 4)})
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
