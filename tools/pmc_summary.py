"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) for the GAE scan kernel
into profiles/r01_gae_pmc.json.  Units and gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md
(section HBM): counters are in KiB; FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced
streaming read on gfx950 -> doubled; WRITE_SIZE taken as is (matches the algorithmic 16 B/element exactly)."""
import csv, json, sys, collections
fetch_csv, write_csv, out = sys.argv[1], sys.argv[2], sys.argv[3]
def per_grid(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "gae_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            d[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {g: sum(v) / len(v) for g, v in d.items()}
f, w = per_grid(fetch_csv, "FETCH_SIZE"), per_grid(write_csv, "WRITE_SIZE")
res = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on bench.py",
       "corrections": "KiB -> bytes (x1024); FETCH_SIZE x2 (gfx950 wide-coalesced-read undercount); WRITE_SIZE x1",
       "launches": {}}
for g in sorted(f):
    # 256 threads per block; 4 rows per block for the cache-resident variant (reward / cost scans in separate lane groups:
    # bench.py's 4096-env launch), 8 rows per block otherwise (the 262 144-env streaming launch)
    n_envs = g // 64 if (g // 64) * 128 <= (4 << 20) else g // 32
    algo = 33.0 * n_envs * 128
    fb, wb = f[g] * 1024 * 2, w.get(g, 0.0) * 1024
    res["launches"][str(n_envs)] = {"grid": g, "fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes": fb + wb,
                                    "algorithmic_bytes_33B_per_elem": algo, "traffic_over_algorithmic": (fb + wb) / algo}
if "4096" in res["launches"]:
    res["hbm_bytes_per_launch_n4096"] = res["launches"]["4096"]["hbm_bytes"]
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
