# GPU box: start/end timeline of the replayed wide step's dispatches (gaps between consecutive kernels)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/wprof
cat > /tmp/w1.py <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import wide_bench
print(wide_bench.one([128, 128], 64, 256))
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/wprof -- python /tmp/w1.py > /tmp/w1.log 2>&1
tail -1 /tmp/w1.log
f=$(find /tmp/wprof -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "mlp_rows" in r["Kernel_Name"] or "wide_" in r["Kernel_Name"]]
sel = sel[-60:]
prev = None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[1][:28] if "(" in r["Kernel_Name"] else r["Kernel_Name"][:28]
    nm = r["Kernel_Name"]
    nm = nm[nm.find("::") + 2:][:24]
    print(f"{nm:26s} dur {(e - s) / 1000:7.2f} us   gap before {((s - prev) / 1000) if prev else 0:7.2f} us")
    prev = e
PY
