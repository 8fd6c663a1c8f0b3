#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of scripts/gpu_profile.sh into profiles/<tag>_*.{csv,md} (tracked)."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
out = "gpurun_out"
os.makedirs("profiles", exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


lines = [f"# rocprofv3 summary `{tag}` (MI355X, bench.py default config: 1M SNPs x 10k samples, k=20, l=30, q=2)", ""]
b = f"{out}/bench_{tag}.json"
if os.path.exists(b) and os.path.getsize(b):
    d = json.loads(open(b).read().strip().splitlines()[-1])
    lines += ["## bench.py line", "```json", json.dumps(d, indent=1), "```", ""]
# the bench line the PROFILED process printed: its HIP-event averages are the ones the rocprofv3 table below must agree with (two
# processes on one box can differ by a few per cent in both GEMMs: where the 10 GB buffer lands, profiles/r1_kbench_summary.md 7)
pl = f"{out}/prof_{tag}.log"
if os.path.exists(pl):
    cand = [ln[ln.index('{"metric'):] for ln in open(pl, errors="replace") if '{"metric' in ln]
    if cand:
        try:
            dp = json.loads(cand[-1])
            r = dp["roofline"]
            lines += ["## the same command under `rocprofv3 --kernel-trace --stats` (the process the table below comes from): HIP-event figures",
                      "", f"ms_per_step {dp['ms_per_step']:.3f}; dominant kernel `{r['kernel']}` avg launch {r['avg_launch_ms'] * 1e3:.1f} us "
                      f"(HIP events on the engine's stream), frac {r['frac']:.3f}; per-step GEMM totals {json.dumps(r['all_kernels_ms_per_step'])}", ""]
        except Exception:
            pass
ks = glob.glob(f"{out}/prof_{tag}/*/*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks[0])))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        f.write(open(ks[0]).read())
    lines += ["## `rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline` (20 timed + 2 warm-up steps per path; default = exact-integer path, then the f32-MFMA path, 2-bit residency, three planes)", "",
              "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rows[:22]:
        lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
    lines.append("")
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write", "sq"):
    for f in glob.glob(f"{out}/pmc_{kind}_{tag}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            pm[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            pm[k]["_ms_" + kind].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
if pm:
    lines += ["## PMC passes (separate `rocprofv3 --pmc` runs, 1 step, per-dispatch averages)", "",
              "FETCH_SIZE/WRITE_SIZE are in KiB-units of 1024 B as reported; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts",
              "128-B requests at 64 B for wide coalesced streams on gfx950, so `fetch_corrected = 2 x FETCH_SIZE`.", "",
              "| kernel | FETCH_SIZE (MB) | x2 corrected (MB) | WRITE_SIZE (MB) | MFMA busy cyc / SIMD | GRBM_GUI_ACTIVE / XCD | MFMA busy % |",
              "|---|---|---|---|---|---|---|"]
    for k, v in pm.items():
        if not any(x in k for x in ("gq_f32", "gtt_f32", "snp_stats", "gq_i8", "gtt_i8", "2bit", "gq_x", "gtt_x", "gq_d", "gtt_d", "gtt_p")):
            continue
        avg = lambda n: (sum(v[n]) / len(v[n])) if v.get(n) else float("nan")
        fetch = avg("FETCH_SIZE") * 1024 / 1e6
        write = avg("WRITE_SIZE") * 1024 / 1e6
        mf = avg("SQ_VALU_MFMA_BUSY_CYCLES") / 1024
        ga = avg("GRBM_GUI_ACTIVE") / 8
        lines.append(f"| `{k}` | {fetch:.1f} | {2 * fetch:.1f} | {write:.1f} | {mf:.3e} | {ga:.3e} | {100 * mf / ga if ga == ga and ga else float('nan'):.1f} |")
    lines.append("")
    with open(f"profiles/{tag}_pmc.json", "w") as f:
        json.dump({k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in pm.items()}, f, indent=1)
open(f"profiles/{tag}_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-14:]))
