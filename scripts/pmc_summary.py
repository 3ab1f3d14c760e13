"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel (short names)."""
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for root in sys.argv[1:]:
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"]
                if "at::native" in name or "elementwise" in name.lower() and "ltxk" not in name:
                    continue
                short = re.sub(r"\(.*", "", name).replace("void ", "").replace("ltxk::", "")[:60]
                acc[short + " wg" + row["Workgroup_Size"] + " grid" + row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
