"""Average per-dispatch counter values per GEMM kernel from scripts/pmc_gemm.sh output.  usage: pmc_gemm_summary.py gpurun_out/<tag>"""
import csv, glob, collections, sys, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm_bf16" not in n:
            continue
        key = re.sub(r"\(.*", "", n.replace("occ_gemm_detail::", "").replace("void ", ""))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c in sorted(v):
        print("   %-44s %16.0f   (n=%d)" % (c, sum(v[c]) / len(v[c]), len(v[c])))
