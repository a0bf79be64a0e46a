"""Per-kernel means of the counters in a rocprofv3 --pmc counter_collection CSV: pmc_kernel_table.py file.csv [kernel regex]"""
import csv, sys, re, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    if pat and not re.search(pat, r["Kernel_Name"]):
        continue
    rows[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in rows.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n=%3d mean %.4g" % (c, len(v), sum(v) / len(v)))
