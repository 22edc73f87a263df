#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_passes.sh) -> the per-kernel HBM traffic JSON that
bench.py quotes as roofline.traffic.

    python tools/pmc_traffic.py gpurun_out/<pmc dir> profiles/r01_x_pmc_traffic.json "<build label>"

hbm_bytes_corrected = 2 * FETCH_SIZE + WRITE_SIZE with both counters in KB of 1024 B: the gfx950 FETCH_SIZE
correction of MI355X_MICROARCH.md (the counter reports half of the bytes fetched).
"""
import collections
import csv
import glob
import json
import os
import sys

# (template arguments as rocprofv3 prints them: round 4's backward is <S4, F16 = true, MASKS>, its forward <S4, TRACE, UNROLL2, MASKS, LEARN>)
STAGE_KERNEL_PREFIX = {"blend_bwd": ("render_bwd_rows_k<4, true", "render_bwd_rows_k<4, 0"),
                       "blend_fwd": ("render_fwd_k<4, false, true",), "preprocess": ("preprocess_fwd_k",), "emit": ("emit_k<true",)}


def main():
    src, dst, label = sys.argv[1], sys.argv[2], sys.argv[3]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(src, "p*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "goi::" not in k or r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            k = k.replace("void ", "").replace("goi::(anonymous namespace)::", "").split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    kernels = {}
    for k, cs in agg.items():
        if "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
            continue
        fs = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
        ws = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
        kernels[k] = {"FETCH_SIZE_KB": round(fs, 1), "WRITE_SIZE_KB": round(ws, 1),
                      "hbm_bytes_corrected": (2 * fs + ws) * 1024}
    stage_kernel = {}
    for stage, prefixes in STAGE_KERNEL_PREFIX.items():
        hits = [k for k in kernels if k.startswith(prefixes)]
        if hits:
            stage_kernel[stage] = max(hits, key=lambda k: kernels[k]["hbm_bytes_corrected"])
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_passes.sh) on the " + label +
                   " build, headline workload; hbm_bytes_corrected = 2*FETCH_SIZE + WRITE_SIZE (KB = 1024 B), the "
                   "gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md",
           "kernels": kernels, "stage_kernel": stage_kernel}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1))


if __name__ == "__main__":
    main()
