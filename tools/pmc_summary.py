"""Summarise rocprofv3 --pmc output (counter_collection.csv files) per kernel:
    python tools/pmc_summary.py LABEL DIR [DIR ...] > profiles/rNN_pmc_summary.csv
Columns: source,kernel,counter,dispatches,avg_value_KB,avg_bytes  (FETCH_SIZE / WRITE_SIZE count KiB)."""
import csv, glob, os, sys
from collections import defaultdict

label, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: [0, 0.0])
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0][:80].replace(",", ";")
            a = acc[(name, r["Counter_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
print("source,kernel,counter,dispatches,avg_value_KB,avg_bytes")
for (name, ctr), (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{label},{name},{ctr},{n},{tot / n:.6g},{tot / n * 1024:.6g}")
