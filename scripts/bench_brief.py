#!/usr/bin/env python3
"""Print the figures of a bench.py line that matter when reading a gpurun tail.  usage: bench_brief.py <file with the JSON line>"""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
r = d["roofline"]
print(f"headline n_gpus={d['n_gpus']} {d['ms_per_step']:.3f} ms/step value={d['value']:.4g} {r['kernel']} {r['avg_launch_ms']:.4f} ms frac={r['frac']:.3f}")
print("  per step:", {k: round(v, 3) for k, v in r["all_kernels_ms_per_step"].items()})
if "step_level" in r:
    print("  step level:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r["step_level"].items() if k != "definition"})
for k in ("f32_mfma_path", "packed_2bit_residency", "packed_2bit_four_planes", "config3_per_gpu_shard", "north_star_literal", "config2_chr22_shape",
          "config5_per_gpu_shard_streamed"):
    v = d.get(k)
    if not v:
        continue
    if "error" in v:
        print(f"  {k}: ERROR {v['error']}")
        continue
    rf = v.get("roofline", {})
    fr = rf.get("frac") if "frac" in rf else {n: round(x.get("frac", 0), 3) for n, x in rf.items() if isinstance(x, dict)}
    print(f"  {k}: {v['ms_per_step']:.3f} ms/step value={v['value']:.4g} frac={fr}")
if d.get("multi_gpu"):
    print("  multi_gpu:", json.dumps(d["multi_gpu"]))
if d.get("parity"):
    p = d["parity"]
    print("  parity:", {k: p[k] for k in ("max_abs_dPC_scores", "max_rel_d_eigenvalue") if k in p} or p)
if d.get("cpu_baseline"):
    c = d["cpu_baseline"]
    print("  cpu_baseline:", c.get("value"), c.get("cores"), c.get("error"))
if d.get("extra_path_errors"):
    print("  extra_path_errors:", d["extra_path_errors"])
