"""Average per-dispatch counter values per kernel from scripts/pmc_kernel.sh output.  usage: pmc_kernel_summary.py gpurun_out/<tag> <name filter>"""
import csv, glob, collections, sys, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if flt not in n:
            continue
        key = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("occ_gemm_detail::", "").replace("void ", "")) + " grid=" + r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c in sorted(v):
        print("   %-32s %16.0f   (n=%d)" % (c, sum(v[c]) / len(v[c]), len(v[c])))
