"""Short per-kernel table from rocprofv3 --kernel-trace --stats output: python3 scripts/trace_summary.py DIR [divide_calls_by]"""
import csv, glob, re, sys
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f, newline="")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
    n = r["Name"]
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"(ltxk::\w+(<[^>]*>)?)", n) or re.search(r"_ZN4ltxk\d+(\w+?)E", n) or re.search(r"at::native::(\w+)", n)
    short = (m.group(1) if m else n)[:60]
    print(f"{short:62s} calls/unit {int(r['Calls'])/div:7.2f}  avg {float(r['AverageNs'])/1e3:8.1f} us  per unit {float(r['TotalDurationNs'])/div/1e3:9.1f} us  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
