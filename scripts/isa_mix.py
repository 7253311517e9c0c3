"""Static instruction mix of the MFMA main loop of every gemm_* kernel in a device assembly file.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I<csrc> <file>.hip -S --cuda-device-only -o x.s && isa_mix.py x.s
A basic block belongs to the main loop when its label comment says `in Loop: Header=<h>` for the header h of the loop
that holds the MFMAs (or is that header)."""
import re, sys, collections

src = open(sys.argv[1]).read()
for f in re.split(r"\n(?=_ZN4cxrk\w+:)", src):
    m = re.match(r"(_ZN4cxrk\w+):", f)
    if not m or "gemm_" not in m.group(1):
        continue
    body = f[: f.find("s_endpgm")]
    # basic blocks: (label line or '; %bb.N:' line) starts a block
    blocks, cur, tag = [], [], ""
    for line in body.split("\n"):
        if re.match(r"\.LBB\d+_\d+:", line) or re.match(r"; %bb\.\d+:", line):
            blocks.append((tag, cur)); cur, tag = [], line
        else:
            cur.append(line)
    blocks.append((tag, cur))
    hdr = None
    for tag, ins in blocks:
        if any("v_mfma" in l for l in ins):
            mm = re.search(r"Header=BB(\d+_\d+)", tag) or re.match(r"\.LBB(\d+_\d+):.*Loop Header", tag)
            if mm:
                hdr = mm.group(1); break
    if hdr is None:
        continue
    c = collections.Counter()
    for tag, ins in blocks:
        if f"Header=BB{hdr}" in tag or tag.startswith(f".LBB{hdr}:"):
            for l in ins:
                op = l.strip().split(" ")[0].split("\t")[0]
                if not op or op.startswith(";") or op.startswith("."):
                    continue
                if op.startswith("v_mfma"): c["mfma"] += 1
                elif op.startswith("v_"): c["valu"] += 1
                elif op.startswith("s_cbranch") or op.startswith("s_branch"): c["branch"] += 1
                elif op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"): c["wait"] += 1
                elif op.startswith("s_"): c["salu"] += 1
                elif op.startswith("ds_read"): c["ds_read"] += 1
                elif op.startswith("ds_write"): c["ds_write"] += 1
                elif op.startswith("buffer_load") or op.startswith("global_load"): c["vmem"] += 1
                else: c["other"] += 1
    name = re.sub(r"_ZN4cxrk\d+gemm_(\w+?)_kernelINS_", r"\1 ", m.group(1))
    name = re.sub(r"EvNT_1PENT0_1PENS_9EpiParamsEiiiiii", "", name)[:70]
    print(f"{name:70s} " + "  ".join(f"{k} {c[k]}" for k in ("mfma", "valu", "salu", "branch", "wait", "ds_read", "ds_write", "vmem", "other")))
