"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs, CSV output) -> profiles/rNN_pmc_hbm.json.
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [workload tag printed by `bench.py --print-workload`]
HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE reports half of a wide coalesced read
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections, csv, json, re, sys


def load(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            k = re.sub(r"(\(anonymous namespace\)|occ_gemm_detail)::", "", r["Kernel_Name"]).split("(")[0]
            k = re.sub(r"^void ", "", k)
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return tot, cnt


ft, fc = load(sys.argv[1], "FETCH_SIZE")
wt, wc = load(sys.argv[2], "WRITE_SIZE")
out = {"kernels": {}, "note": "KiB per launch averaged over all launches of the run; hbm_bytes_per_launch_corrected = (2*FETCH_SIZE + WRITE_SIZE) * 1024"}
for k in sorted(set(ft) | set(wt)):
    f = ft[k] / max(fc[k], 1); w = wt[k] / max(wc[k], 1)
    out["kernels"][k] = {"FETCH_SIZE_KiB_per_launch": round(f, 1), "WRITE_SIZE_KiB_per_launch": round(w, 1),
                         "hbm_bytes_per_launch_corrected": int((2 * f + w) * 1024), "launches": max(fc[k], wc[k])}
import os
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root)
import bench                                    # ONE list of GEMM sources (bench.GEMM_SOURCES): the profile is tied to exactly the files bench.py hashes
out["gemm_src_sha16"] = bench.gemm_source_sha()
# the workload the counters were taken on (bench.py's config.workload string + dtype): bench.py reports `traffic` only for that workload
if len(sys.argv) > 4:
    out["workload"] = sys.argv[4]
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print("wrote", sys.argv[3], len(out["kernels"]), "kernels")
