"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid, workgroup) launch count, mean and total duration."""
import csv, sys, collections, re
path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = collections.defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if pat and not re.search(pat, name):
            continue
        short = re.sub(r"(\(anonymous namespace\)|occ_gemm_detail)::", "", name).split("(")[0][-60:]
        key = (short, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
        rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in rows.values())
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print("%-62s grid %8s x%-4s wg %4s  n %5d  mean %9.1f us  total %9.2f ms  %5.1f%%" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, 100.0 * sum(v) / tot))
print("total %.2f ms" % (tot / 1e6))
