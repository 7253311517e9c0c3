"""Summarise a rocprofv3 kernel_stats.csv: python stats_summary.py <csv> <steps_in_profile>"""
import csv, sys, re


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("cxrk::", "")
    m = re.match(r"(gemm_\w+_kernel)<(.*)>\(", name)
    if not m:
        return re.sub(r"\(.*", "", name)[:60]
    args = m.group(2)
    loaders = re.findall(r"(Dma\w+|\w+[KM]C)(?:<(\d+), (F32|PL)[^>]*>)?", args)
    tile = re.search(r">?, (\d), (\d)$", args)
    parts = [f"{a}{'<' + f + '>' if f == 'PL' else ''}" for a, _, f in loaders[:2]]
    cfg = re.search(r"PwCfg<(\d), (\d), (\d)>", args)
    if cfg:
        parts.insert(0, {"244": "Pw256", "222": "Pw128", "412": "Pw256x64", "142": "Pw64x256"}.get("".join(cfg.groups()), "Pw?"))
    return f"{m.group(1)}<{','.join(parts)}{',' + tile.group(1) + 'x' + tile.group(2) if tile else ''}>"


if __name__ == "__main__":
    rows = list(csv.DictReader(open(sys.argv[1])))
    n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"total kernel time {tot / n / 1e6:.1f} ms/step")
    for r in rows[:40]:
        print(f"{short(r['Name']):62s} {int(r['Calls']):5d} calls {float(r['TotalDurationNs']) / n / 1e6:8.2f} ms/step  avg {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):5.2f}%")
