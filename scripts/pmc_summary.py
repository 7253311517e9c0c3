"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel and (optionally) write profiles/pmc_traffic.json:
    python pmc_summary.py [--json out.json] <FETCH_SIZE csv> <WRITE_SIZE csv>
Kernel keys match bench.py's launch labels: gemm_pw_kernel<Pw256|Pw128,DmaA,DmaB>, gemm_x3_kernel<A<PL>,B<PL>,WM,WN>,
gemm_f32_kernel<A,B,WM,WN>."""
import csv, sys, collections, re, json
args = sys.argv[1:]
out_json = None
if args and args[0] == "--json":
    out_json, args = args[1], args[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)

import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary_key import key  # noqa: E402

for path in args:
    for r in csv.DictReader(open(path)):
        k = key(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
print("kernel, launches, FETCH_SIZE(KB)/launch, WRITE_SIZE(KB)/launch, HBM bytes/launch = (2*FETCH+WRITE)*1024 [gfx950 FETCH_SIZE counts half of wide reads]")
js = {}
for k, d in sorted(agg.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))):
    n = max(cnt[k].values())
    f, w = d.get("FETCH_SIZE", 0) / max(1, cnt[k].get("FETCH_SIZE", 1)), d.get("WRITE_SIZE", 0) / max(1, cnt[k].get("WRITE_SIZE", 1))
    print(f"{k:58s} {n:5d} {f:14.0f} {w:14.0f} {(2 * f + w) * 1024 / 1e6:12.1f} MB")
    if k.startswith("gemm_"):
        js[k] = {"launches_in_profile": n, "FETCH_SIZE_KB_per_launch": round(f), "WRITE_SIZE_KB_per_launch": round(w),
                 "hbm_bytes_per_launch": (2 * f + w) * 1024}
if out_json:
    json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes, scripts/pmc_traffic.sh) -- python3 bench.py "
                          "--steps 1 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline",
               "correction": "gfx950: HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half "
                             "of wide coalesced reads)",
               "round": 3, "precision": "split_bf16", "kernels": js}, open(out_json, "w"), indent=1)
