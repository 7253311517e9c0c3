"""rocprofv3 kernel name -> bench.py launch label (shared by pmc_summary.py and pmc_sq_summary.py)."""
import re


def key(k):
    m = re.search(r"gemm_pw_kernel<cxrk::PwCfg<(\d), (\d), (\d)>, cxrk::(\w+)<[^>]*>, cxrk::(\w+)<[^>]*>\s*>", k)
    if m:
        cfg = {"244": "Pw256", "222": "Pw128", "412": "Pw256x64", "142": "Pw64x256"}.get(m.group(1) + m.group(2) + m.group(3), "Pw?")
        return f"gemm_pw_kernel<{cfg},{m.group(4)},{m.group(5)}>"
    m = re.search(r"(gemm_x3_kernel)<cxrk::(\w+)<\d+, cxrk::PL[^>]*>, cxrk::(\w+)<\d+, cxrk::PL[^>]*>\s*, (\d), (\d)\s*>", k)
    if m:
        return f"{m.group(1)}<{m.group(2)}<PL>,{m.group(3)}<PL>,{m.group(4)},{m.group(5)}>"
    m = re.search(r"(gemm_\w+_kernel)<cxrk::(\w+)<[^>]*>, cxrk::(\w+)<[^>]*>\s*(?:, (\d), (\d))?\s*>", k)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)}" + (f",{m.group(4)},{m.group(5)}>" if m.group(4) else ">")
    return re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", "")).replace("void ", "").replace("cxrk::", "")[:50]
