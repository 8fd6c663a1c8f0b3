#!/usr/bin/env python3
"""Per-launch timeline of ONE gpca_rsvd call out of a rocprofv3 --kernel-trace CSV: start offset, duration and the idle gap before
every kernel between two consecutive k_omega launches (k_omega opens a call).  usage: call_timeline.py <kernel_trace.csv> [call index]"""
import csv
import sys


def main():
    f = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_omega" in r["Kernel_Name"]]
    a, b = idx[which], idx[which + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end, gaps, busy = t0, 0, 0
    print("| start us | dur us | gap us | kernel | grid | wg |\n|---|---|---|---|---|---|")
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = s - prev_end
        gaps += max(gap, 0); busy += e - s
        print(f"| {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap / 1e3:.1f} | `{r['Kernel_Name'][:60]}` | {r.get('Grid_Size', '')} | {r.get('Workgroup_Size', '')} |")
        prev_end = max(prev_end, e)
    print(f"\ncall span {(prev_end - t0) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, idle gaps {gaps / 1e3:.1f} us, {b - a} launches")


if __name__ == "__main__":
    main()
