"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel: python pmc_summary.py <csv> [<csv2> ...]"""
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        m = re.search(r"gemm_f32_kernel<cxrk::(\w+)<\d+>, cxrk::(\w+)<\d+>, (\d), (\d)>", k)
        k = f"gemm_f32_kernel<{m.group(1)},{m.group(2)},{m.group(3)},{m.group(4)}>" if m else re.sub(r"\(.*", "", k).replace("(anonymous namespace)::", "")[:50]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print("kernel, launches, FETCH_SIZE(KB)/launch, WRITE_SIZE(KB)/launch, HBM bytes/launch = (2*FETCH+WRITE)*1024 [gfx950 FETCH_SIZE counts half of wide reads]")
for k, d in sorted(agg.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))):
    n = max(cnt[k].values())
    f, w = d.get("FETCH_SIZE", 0) / max(1, cnt[k].get("FETCH_SIZE", 1)), d.get("WRITE_SIZE", 0) / max(1, cnt[k].get("WRITE_SIZE", 1))
    print(f"{k:52s} {n:5d} {f:14.0f} {w:14.0f} {(2 * f + w) * 1024 / 1e6:12.1f} MB")
