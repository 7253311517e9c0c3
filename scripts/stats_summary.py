"""Summarise a rocprofv3 kernel_stats.csv: python stats_summary.py <csv> <steps_in_profile>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / n / 1e6:.1f} ms/step")
for r in rows[:32]:
    name = r["Name"]
    m = re.search(r"(gemm_\w+_kernel)<cxrk::(\w+)<\d+>, cxrk::(\w+)<\d+>, (\d), (\d)>", name)
    short = f"{m.group(1)}<{m.group(2)},{m.group(3)},{m.group(4)}x{m.group(5)}>" if m else re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:56]
    print(f"{short:58s} {int(r['Calls']):5d} calls {float(r['TotalDurationNs']) / n / 1e6:8.2f} ms/step  avg {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):5.2f}%")
