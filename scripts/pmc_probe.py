"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch, grouped by kernel + grid size.

    python scripts/pmc_probe.py <dir with *counter_collection.csv files (searched recursively)> [kernel substring]
"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "igemm"
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if sub not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"]
        k = k[k.find("igemm"):][:40] if "igemm" in k else k[:40]
        key = (k, r.get("Grid_Size", "?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")))
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for key in sorted(acc):
    print(f"== {key[0]} grid={key[1]} lds={key[2]}")
    c = {n: v[0] / max(1, v[1]) for n, v in acc[key].items()}
    for n in sorted(c):
        print(f"   {n:32s} {c[n]:16.1f}")
    if "SQ_BUSY_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        print(f"   -> MFMA busy / SQ busy = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}")
