"""Per-kernel MFMA utilisation and occupancy from rocprofv3 --pmc counter CSVs collected over one bench step
(scripts/pmc_sq_bench.sh):

    python pmc_sq_summary.py [--json profiles/pmc_sq.json] <pass-1 csv> [<pass-2 csv>]

Units (MI355X_MICROARCH.md, 'price list'): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles (32 per v_mfma_f32_32x32x16_bf16, 64 per
v_mfma_f32_32x32x2_f32), summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so a dispatch lasted
GRBM_GUI_ACTIVE / 8 cycles; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles.
    mfma_busy      = SQ_VALU_MFMA_BUSY_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)      fraction of SIMD-cycles with the matrix pipe busy
    waves_per_simd = 4 * SQ_WAVE_CYCLES / (1024 * GRBM_GUI_ACTIVE / 8)            average resident waves per SIMD
    valu_per_mfma  = SQ_INSTS_VALU / SQ_INSTS_MFMA (SQ_INSTS_VALU includes the MFMAs)
Kernel keys match bench.py's launch labels (scripts/pmc_summary.py's `key`)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary_key import key  # noqa: E402

args = sys.argv[1:]
out_json = None
if args and args[0] == "--json":
    out_json, args = args[1], args[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for path in args:
    for r in csv.DictReader(open(path)):
        k = key(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
rows = []
for k, d in agg.items():
    n = max(cnt[k].values())
    per = {c: v / max(1, cnt[k][c]) for c, v in d.items()}          # per launch (GRBM_GUI_ACTIVE may appear in both passes)
    cyc = per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0:
        continue
    simd_cyc = 1024.0 * cyc
    row = {"launches_in_profile": n, "cycles_per_launch": cyc,
           "mfma_busy": per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cyc,
           "waves_per_simd": 4.0 * per.get("SQ_WAVE_CYCLES", 0.0) / simd_cyc,
           "insts_mfma_per_launch": per.get("SQ_INSTS_MFMA", 0.0),
           "valu_per_mfma": per.get("SQ_INSTS_VALU", 0.0) / per["SQ_INSTS_MFMA"] if per.get("SQ_INSTS_MFMA") else None}
    if "SQ_WAIT_ANY" in per and per.get("SQ_WAVE_CYCLES"):
        row["wait_any_frac_of_wave_cycles"] = per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"]
    if per.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_frac"] = per.get("SQ_LDS_BANK_CONFLICT", 0.0) / per["SQ_LDS_IDX_ACTIVE"]
    row["total_cycles"] = cyc * n
    rows.append((k, row))
rows.sort(key=lambda kr: -kr[1]["total_cycles"])
print("kernel, launches, kcycles/launch, MFMA busy (fraction of SIMD-cycles), waves/SIMD, VALU/MFMA, wait_any/wave_cycles, LDS conflict frac")
for k, r in rows:
    v = r["valu_per_mfma"]
    print(f"{k:58s} {r['launches_in_profile']:5d} {r['cycles_per_launch'] / 1e3:10.1f} {r['mfma_busy']:8.3f} {r['waves_per_simd']:7.2f} "
          f"{(f'{v:6.2f}' if v else '     -')} {r.get('wait_any_frac_of_wave_cycles', float('nan')):7.3f} {r.get('lds_bank_conflict_frac', float('nan')):7.3f}")
tot_busy = sum(r["mfma_busy"] * r["total_cycles"] for _, r in rows)
tot_cyc = sum(r["total_cycles"] for _, r in rows)
print(f"all kernels of the step: MFMA busy {tot_busy / tot_cyc:.3f} of SIMD-cycles over {tot_cyc / 1e6:.1f} Mcycles of kernel time")
if out_json:
    json.dump({"command": "rocprofv3 --kernel-trace --pmc <SQ / GRBM counters, two passes> (scripts/pmc_sq_bench.sh) -- python3 bench.py --steps 1 "
                          "--warmup 1 --no-secondary --no-cpu-baseline --no-roofline",
               "definitions": "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); waves_per_simd = 4 * SQ_WAVE_CYCLES / "
                              "the same denominator (MI355X_MICROARCH.md units)",
               "step_mfma_busy": tot_busy / tot_cyc,
               "kernels": {k: {a: b for a, b in r.items() if a != "total_cycles"} for k, r in rows if k.startswith("gemm_")}},
              open(out_json, "w"), indent=1)
