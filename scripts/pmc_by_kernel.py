"""Sums rocprofv3 --pmc counter values per kernel name: python scripts/pmc_by_kernel.py <counter_collection.csv>"""
import csv, sys, collections

rows = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), k)
    if key not in seen:
        seen.add(key); calls[k] += 1
names = sorted({c for v in rows.values() for c in v})
print("kernel,calls," + ",".join(names))
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print('"%s",%d,' % (k[:90], calls[k]) + ",".join("%.0f" % v.get(c, 0) for c in names))
