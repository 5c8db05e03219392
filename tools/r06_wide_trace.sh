cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/wprof
cat > /tmp/w1.py <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools"))
import wide_bench
print(wide_bench.one([128, 128], 64, 256))
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wprof -- python /tmp/w1.py > /tmp/w1.log 2>&1
tail -2 /tmp/w1.log
f=$(find /tmp/wprof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>6s} avg_ns {float(r['AverageNs']):9.0f} pct {r['Percentage']}")
PY
