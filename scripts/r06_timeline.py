"""Round 6: the kernel timeline of ONE step from a rocprofv3 --kernel-trace csv (start offset, duration, queue) — who overlaps whom.
Usage: python scripts/r06_timeline.py <kernel_trace.csv> [kernel that ends the step: default k_scatter]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
end_name = sys.argv[2] if len(sys.argv) > 2 else "k_scatter"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if end_name in r["Kernel_Name"]]
lo, hi = ends[-2] + 1, ends[-1] + 1  # the last complete step
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ykk::", "")
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  q{r.get('Queue_Id','?'):>2}  {name[:60]}  grid {r.get('Grid_Size','?')} wg {r.get('Workgroup_Size','?')} lds {r.get('LDS_Block_Size','?')} vgpr {r.get('VGPR_Count','?')}")
print(f"step: {(int(rows[hi-1]['End_Timestamp']) - t0)/1e3:.1f} us")
