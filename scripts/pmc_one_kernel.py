"""Average the counters of a rocprofv3 --pmc counter_collection CSV over the dispatches of kernels matching a pattern:
    python scripts/pmc_one_kernel.py <counter_collection.csv> <kernel regex>"""
import collections, csv, re, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if re.search(sys.argv[2], r["Kernel_Name"]):
            rows[(r["Kernel_Name"].split("(")[0][-70:], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in rows.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s n %4d  mean %16.1f" % (c, len(v), sum(v) / len(v)))
