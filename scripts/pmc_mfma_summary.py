"""rocprofv3 --pmc MfmaUtil GRBM_GUI_ACTIVE (CSV) -> per-kernel MFMA utilisation.  MfmaUtil is rocprofv3's derived metric
100 * sum(SQ_VALU_MFMA_BUSY_CYCLES) / (max(GRBM_GUI_ACTIVE) * SIMD_NUM) per dispatch; launches of a kernel are averaged weighted by
their GRBM_GUI_ACTIVE (i.e. by duration).   usage: pmc_mfma_summary.py <counter_collection.csv> <out.json>"""
import collections, csv, json, re, sys
util, act = collections.defaultdict(dict), collections.defaultdict(dict)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = re.sub(r"^void ", "", re.sub(r"(\(anonymous namespace\)|occ_gemm_detail)::", "", r["Kernel_Name"]).split("(")[0])
        d = r.get("Dispatch_Id") or r.get("Correlation_Id")
        if r["Counter_Name"] == "MfmaUtil":
            util[k][d] = util[k].get(d, 0.0) + float(r["Counter_Value"])
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[k][d] = max(act[k].get(d, 0.0), float(r["Counter_Value"]))
out = {"note": "duration-weighted mean of rocprofv3's MfmaUtil (percent of SIMD-cycles with the MFMA pipe busy) over the launches of each kernel", "kernels": {}}
rows = []
for k in util:
    w = sum(act[k].get(d, 1.0) for d in util[k])
    m = sum(util[k][d] * act[k].get(d, 1.0) for d in util[k]) / max(w, 1e-9)
    rows.append((w, k, m, len(util[k])))
for w, k, m, n in sorted(rows, reverse=True):
    if m > 0:
        out["kernels"][k] = {"launches": n, "mfma_util_pct": round(m, 2)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in list(out["kernels"].items())[:10]:
    print("%-60s %6.2f %%  (%d launches)" % (k[:60], v["mfma_util_pct"], v["launches"]))
