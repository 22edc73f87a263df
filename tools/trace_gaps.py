#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: every kernel of the last
full step with its start offset, duration and the idle gap before it, plus the busy/idle split.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o NAME -- python bench.py --steps 5 ...
    python tools/trace_gaps.py DIR/NAME_kernel_trace.csv [anchor-substring]

A step is delimited by consecutive launches of the anchor kernel (default: preprocess_fwd_k).
"""
import csv
import sys


def main():
    path = sys.argv[1]
    anchor = sys.argv[2] if len(sys.argv) > 2 else "preprocess_fwd_k"
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(marks) < 3:
        sys.exit("not enough anchor launches")
    # steps with grads are the ones followed by a backward kernel; take the last window that holds one
    pick = None
    for a, b in zip(marks[:-1], marks[1:]):
        if any("bwd" in rows[i][2] for i in range(a, b)):
            pick = (a, b)
    a, b = pick
    t0 = rows[a][0]
    busy = 0
    prev_end = t0
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>8}  kernel")
    for i in range(a, b):
        s, e, n = rows[i]
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("goi::", "").split("(")[0][:70]
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:8.1f}  {short}")
        busy += e - s
        prev_end = max(prev_end, e)
    span = rows[b][0] - t0
    print(f"step span {span / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
