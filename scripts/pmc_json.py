#!/usr/bin/env python
"""profiles/r03_pmc.json from the text summaries of the separate rocprofv3 --pmc passes (scripts/pmc_passes.sh):
    python scripts/pmc_json.py gpurun_out/pmc3 > profiles/r03_pmc.json
FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B... as reported by rocprofv3 (unit KB = 1024 B, checked in round 2 by
scripts/pmc_calibrate.py); FETCH_SIZE is doubled (MI355X_MICROARCH.md, gfx950: 16 B/lane streaming reads and LDS-DMA are tallied
at half; own calibration x1.91).  What FETCH_SIZE counts leaves the XCD L2s — Infinity-Cache hits included: L2-MISS bytes, an
upper bound on HBM reads."""
import json
import re
import sys


def parse(path):
    rows = {}
    for ln in open(path):
        m = re.match(r"^(.*?)\s+([A-Z_0-9]+)\s+(\d+)\s+([0-9.e+]+)\s+([0-9.e+]+)\s*$", ln.rstrip("\n"))
        if m:
            rows[(m.group(1).strip(), m.group(2))] = (int(m.group(3)), float(m.group(4)), float(m.group(5)))
    return rows


def main(d):
    fetch, write, sq = parse(f"{d}/fetch.txt"), parse(f"{d}/write.txt"), parse(f"{d}/sq.txt")
    lean = [k for (k, c) in fetch if re.match(r"gemm_pp_kernel<[^,]+, unsigned short, true", k) and c == "FETCH_SIZE"]
    commit = None
    try:
        import os
        commit = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".commit_stamp")).read().strip() or None
    except OSError:
        pass
    out = {"collected_on_commit": commit, "source": "rocprofv3 --pmc, separate passes (--kernel-trace only), scripts/pmc_passes.sh around: python bench.py --steps 2 "
                     "--warmup 1 --no-cpu-baseline --no-latency --no-secondary --no-recipe (the code of the round named in the file name)",
           "kernel": "gemm_pp_kernel (bf16 in; the lean NT / NN / TN instantiations)",
           "correction": "FETCH_SIZE x2 on gfx950 (guide; own calibration x1.91 in round 2); WRITE_SIZE exact; unit 1024 B",
           "what_fetch_size_counts": "requests leaving the XCD L2s, Infinity-Cache hits included: L2-miss (fabric-side) bytes, an "
                                     "upper bound on HBM bytes",
           "by_instantiation": {}}
    tot_n = tot_b = tot_mfma = tot_gui = 0.0
    for k in lean:
        n, fs, _ = fetch[(k, "FETCH_SIZE")]
        _, ws, _ = write.get((k, "WRITE_SIZE"), (n, 0.0, 0.0))
        bytes_per = (2.0 * fs + ws) * 1024.0 / n
        ent = {"launches": n, "fetch_size_kb_per_launch": round(fs / n, 1), "write_size_kb_per_launch": round(ws / n, 1),
               "l2_miss_bytes_per_launch": int(bytes_per)}
        if (k, "SQ_VALU_MFMA_BUSY_CYCLES") in sq and (k, "GRBM_GUI_ACTIVE") in sq:
            mf, gui = sq[(k, "SQ_VALU_MFMA_BUSY_CYCLES")][1], sq[(k, "GRBM_GUI_ACTIVE")][1]
            # MFMA-busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
            ent["mfma_util"] = round(mf / 1024.0 / (gui / 8.0), 4)
            tot_mfma += mf
            tot_gui += gui
        out["by_instantiation"][k] = ent
        tot_n += n
        tot_b += bytes_per * n
    out["l2_miss_bytes_per_launch"] = int(tot_b / max(tot_n, 1))
    out["hbm_bytes_per_launch"] = out["l2_miss_bytes_per_launch"]          # (field name bench.py read in round 2; same number)
    if tot_gui:
        out["mfma_util"] = round(tot_mfma / 1024.0 / (tot_gui / 8.0), 4)
    for k in [k for (k, c) in fetch if k.startswith("adamw_k") and c == "FETCH_SIZE"]:
        n, fs, _ = fetch[(k, "FETCH_SIZE")]
        ws = write.get((k, "WRITE_SIZE"), (n, 0.0, 0.0))[1]
        out["adamw_k"] = {"fetch_gb_per_launch": round(2.0 * fs * 1024 / n / 1e9, 1), "write_gb_per_launch": round(ws * 1024 / n / 1e9, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
