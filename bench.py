#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

metric : SNPs x samples / sec through the randomized PCA (gpca_rsvd) at k = 20  (l = 30, q = 2)
step   : one gpca_rsvd over the resident int8 genotype matrix (sketch + 2 power iterations + projection +
         small SVD + scores + loadings = 4 passes over G, 12*l flop per genotype -- SURVEY.md 8d)
N = 1  : configs[1] = synthetic 1M SNPs x 10k samples int8, k = 20, fixed seed, resident in HBM
N > 1  : SNP-row shards, one rank per GPU, weak scaling: every rank holds configs[3]'s per-GPU shard (1.25M SNPs x 100k samples int8,
         125 GB; 8 ranks = configs[3] itself, 10M x 100k); the N x l sketch is all-reduced with RCCL inside libgpca.so (f64).
         `python bench.py --gpus N` starts its own N ranks (the parent touches no GPU); under torch.distributed.run the ranks are
         taken as given.  No torch in either case: genomic_pca_amd/launch.py carries the unique id, the barriers and the max.

One JSON line on rank 0.  `value` is the configuration BASELINE.json names -- int8 genotypes resident in HBM -- on the engine's default
GEMM path for them: exact-integer GEMMs, HBM-bound.  The same job with the genotypes resident as 2-bit codes (what both command lines
choose for >= 1 024 samples: less HBM, faster kernels) is `packed_2bit_residency`; `f32_mfma_path` is the same job on
v_mfma_f32_32x32x2_f32 (north_star's MFMA-fp32 roofline).  `roofline` is for the
dominant kernel, from HIP events recorded on the engine's own stream inside the timed region; `parity` is
max|dPC| against the oracle on a small seeded case; `cpu_baseline` is the oracle's f32 restatement ("port")
timed on this box's host cores on a bounded sample (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 FLOP/clk x 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec
# What this part's memory system delivers to a read-only nt stream shaped like K1's LDS-DMA fill (8 rows x 128 B per wave instruction):
# 6.94-6.96 TB/s on the best-placed of ten 10.5 GB buffers (profiles/r3_kbench_place_contig.log, scripts/kbench/kbench_place.hip).  A
# measured ceiling for the read-bound kernels, printed beside `frac` (which is always against the 8 TB/s spec), never instead of it.
HBM_STREAM_CEILING_GBS = 6950.0
HBM_STREAM_CEILING_SOURCE = ("profiles/r3_kbench_place_contig.log: read-only nt stream with K1's fill shape, best-placed 10.5 GB buffer, "
                             "6.94-6.96 TB/s (scripts/kbench/kbench_place.hip)")


def cpu_baseline(N, k, oversample, q, seed, target_s=15.0):
    """Oracle restatement (REAL=float, OpenMP) timed on a bounded sample of the same workload, on the CPUs this process really owns
    (affinity mask, cgroup quota, physical cores: oracle.usable_cpus)."""
    from oracle import oracle as O   # cpu_baseline leg only
    import genomic_pca_amd as g
    cpus = O.usable_cpus()
    threads = cpus["threads"]
    O.set_num_threads(threads, "f32")
    l = k + oversample

    def run(Ms):
        th = g.synth_thresholds(Ms, 3, seed=seed)
        G = O.synth_genotypes(Ms, N, seed, th)
        st = O.snp_stats(G, N, 0.0, 0.0, 1.0)
        r, b = O.scale_shift(st["mu"], st["sigma"], st["keep"])
        t0 = time.perf_counter()
        O.rsvd_port(G, N, r, b, k, oversample, q, seed=seed, real="f32")
        return time.perf_counter() - t0
    Ms = 25000
    t = run(Ms)
    rate = Ms * N / t
    half = None
    if threads >= 2:      # the same sample on half the threads: the baseline must scale with the cores it is given
        O.set_num_threads(threads // 2, "f32")
        half = Ms * N / run(Ms)
        O.set_num_threads(threads, "f32")
    Ms2 = int(min(max(rate * target_s / N, Ms), 1_000_000))   # ~target_s of CPU work, capped at 10 GB of genotypes
    if Ms2 > 1.5 * Ms:
        t = run(Ms2); Ms = Ms2
    flop = (2 + 4 * q + 2) * l * Ms * N
    return {"value": Ms * N / t, "unit": "SNPs*samples/s", "cores": threads, "kind": "port", "gflops": flop / t / 1e9,
            "value_on_half_the_threads": half, "cpus": cpus,
            "sample": f"oracle/gpca_oracle.c REAL=float, {Ms} SNPs x {N} samples, k={k}, l={l}, q={q}, "
                      f"{t:.1f} s on {threads} OpenMP threads = {flop / t / 1e9:.0f} GFLOP/s (CPU restatement, not the Rust/faer binary; "
                      f"threads = min(affinity {cpus['affinity']}, cgroup quota {cpus['cgroup_quota_cpus']}, physical cores {cpus['physical_cores']}))"}


def parity_check(g, precision, seed):
    """max|dPC| of the measured path against the oracle's f64 restatement (same seed) on a small seeded case --
    the second half of BASELINE.json's metric.  The oracle is only the checker here."""
    from oracle import oracle as O
    M, N, P, k = 20000, 1000, 16, 10
    th = g.synth_thresholds(M, P, seed=seed, fst=0.2)
    with g.GpcaEngine(precision=precision) as e:
        e.synth_genotypes(M, N, seed, th)
        G = e.download_genotypes_i8()
        st = e.snp_stats(g.QcConfig.none())
        e.rsvd(k, 10, 2, seed=seed)
        r, b = O.scale_shift(st["mu"], st["sigma"], st["keep"])
        R = O.rsvd(G, N, r, b, k, 10, 2, seed=seed)     # the checker: LAPACK QR / SVD, no small-dense code shared with the product
        E = O.exact_pca(G, N, r, b, k)                  # BASELINE.md's parity metric: exact f64 PCA (eigh of the Gram)
        sc2, ev2 = e.scores(f64=True), e.eigenvalues().copy()
        load2 = e.loadings().astype(np.float64)
        e.rsvd(k, 10, 4, seed=seed)                     # two more power iterations: the randomized PCA converged below the bar
        gap = lambda S, ev: {"max_abs_dPC_scores": O.max_abs_dpc(S, E["scores"]),
                             "max_rel_d_eigenvalue": float(np.max(np.abs(ev - E["eigenvalues"]) / E["eigenvalues"]))}
        return {"case": f"{M}x{N} k={k} vs oracle f64 (same sketch; Householder QR + LAPACK SVD)",
                "vs_exact_pca_eigh": {"product_q2": gap(sc2, ev2), "oracle_q2": gap(R["scores"], R["eigenvalues"]),
                                      "product_q4": gap(e.scores(f64=True), e.eigenvalues()),
                                      "note": "exact PCA of the standardised matrix (all k PCs are population structure here); at q = 2 the gap is "
                                              "the randomized PCA's own convergence, identical for the oracle; at q = 4 it is under the 1e-4 bar"},
                "max_abs_dPC_scores": O.max_abs_dpc(sc2, R["scores"]),
                "max_abs_dPC_loadings": O.max_abs_dpc(load2, R["loadings"]),
                "max_rel_d_eigenvalue": float(np.max(np.abs(ev2 - R["eigenvalues"]) / R["eigenvalues"])),
                "tolerance": 1e-4}


# An EXTRA path (never the headline) starts right after the previous path's engine was closed, and a large hipFree leaves the part
# 4 % slow on both GEMMs for a second or two of sustained load (profiles/r4_kbench_summary.md section 9): its untimed warm-up runs
# for at least this long.  The headline path is the first engine of the process and gets exactly --warmup steps.
EXTRA_PATH_MIN_WARM_S = 1.5


def weak_scaling_fields(ms_per_step, solo_ms):
    """Weak-scaling efficiency of a multi-rank run against ITS OWN reference: every rank timed the same shard as a matrix of its own
    (no exchange) in the same process just before it joined the exchange (`solo_ms`, one figure per rank).  ms_per_step is the contract's
    max over ranks.  Two readings: against the mean of the solo times (what the exchange + waiting for peers cost on average) and
    against the slowest rank's solo time (a max-over-ranks step can never beat the slowest GPU: this one isolates the exchange from
    the +-4 % spread between processes / GPUs that a max over ranks absorbs).  NOT against the --gpus 1 line, which is configs[1]."""
    solo = [float(x) for x in solo_ms if x is not None]
    if not solo or not ms_per_step:
        return {"weak_scaling_efficiency": None, "weak_scaling_efficiency_vs_slowest_rank": None}
    mean, worst = sum(solo) / len(solo), max(solo)
    return {"weak_scaling_efficiency": mean / ms_per_step, "weak_scaling_efficiency_vs_slowest_rank": worst / ms_per_step,
            "same_shard_without_exchange_ms_mean": mean, "same_shard_without_exchange_ms_slowest": worst,
            "same_shard_spread_between_ranks": (worst - min(solo)) / mean,
            "weak_scaling_reference": "the same shard on the same GPU in the same process without the exchange (three calls before the "
                                      "communicator is joined); NOT the --gpus 1 line, which is BASELINE.json configs[1] (1M x 10k)",
            "expected_band": "0.95-1.0 vs the slowest rank (3 x N x L f64 sketch + one L^2 Gram per call: < 1 % of a step over xGMI); "
                             "up to 4 % lower vs the mean, which also carries the spread between the ranks' GPUs"}


def shard_bytes_needed(M, N, storage, k, oversample, streamed=False, panel_rows=0, ring=3):
    """Device bytes a rank needs before it generates anything: the resident shard (padded pitch) or the panel ring, plus the solver's
    workspace (gpca.h: about 1 KiB per SNP row and 8 KiB per sample for l <= 32; twice that for wider sketches) and a 2 GiB margin."""
    l = k + oversample
    wide = 1 if l <= 32 else (2 if l <= 64 else 4)
    if storage == "2bit":
        ldg = -(-N // 1024) * 1024
        pitch = ldg // 4 + (256 if (ldg // 4 // 256) % 2 == 0 else 0)
    else:
        ldg = -(-N // 256) * 256
        pitch = ldg + (256 if (ldg // 256) % 2 == 0 else 0)
    rows = (min(M, (panel_rows or 131072)) * ring) if streamed else -(-M // 128) * 128
    return rows * pitch + wide * (M * 1024 + N * 8192) + (2 << 30)


def memory_preflight(eng, what, need_bytes, rank):
    """Fail with a sentence instead of an out-of-memory abort half-way through generating 125 GB."""
    free_b, total_b = eng.device_memory()
    if need_bytes > free_b:
        raise SystemExit(f"bench.py: rank {rank}: {what} needs about {need_bytes / 2**30:.1f} GiB of device memory, "
                         f"{free_b / 2**30:.1f} GiB of {total_b / 2**30:.1f} GiB are free on its GPU -- is the device shared, or is --snps / --samples "
                         f"meant for a larger part?  (--storage 2bit needs a quarter of the shard; --streamed none of it)")
    return {"needed_GiB": need_bytes / 2**30, "free_GiB": free_b / 2**30, "total_GiB": total_b / 2**30}


# HIP events around a kernel cost the stream ~5 us of idle time per pair; the per-kernel figures of the line come from every TIMING_EVERY-th
# step of the timed region (gpca_enable_timings(h, n)): still measured live inside that region, on the engine's own stream, at a quarter
# of the overhead (12 event pairs a step were ~0.06 ms of a 10 ms step).  Per-step kernel figures divide by the sampled steps.
TIMING_EVERY = 4


def sampled_steps(steps):
    return -(-steps // TIMING_EVERY)


def timed_run(eng, a, k, barrier, rdzv, min_warm_s=0.0):
    t_w = time.perf_counter()
    for _ in range(a.warmup):
        eng.rsvd(k, a.oversample, a.power_iters, a.rfit_seed)
    while time.perf_counter() - t_w < min_warm_s:
        eng.rsvd(k, a.oversample, a.power_iters, a.rfit_seed)
    eng.enable_timings(TIMING_EVERY)       # HIP events on the engine's own stream (off by default in the library), every TIMING_EVERY-th step
    eng.reset_timings()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.rsvd(k, a.oversample, a.power_iters, a.rfit_seed)
    barrier()
    dt = time.perf_counter() - t0
    timings = eng.timings()
    ns = sampled_steps(a.steps)
    if rdzv is not None:
        ar = timings.get("allreduce")
        mine = {"dt": dt, "allreduce_ms_per_step": (ar["total_ms"] / ns) if ar else None,
                "gemm_ms_per_step": sum(t["total_ms"] for n_, t in timings.items() if n_.startswith("gemm")) / ns}
        every = rdzv.allgather(mine)               # small host objects through the launcher's hub (genomic_pca_amd/launch.py)
        per_rank = [e_["dt"] for e_ in every]
        dt = max(per_rank)                         # the contract's max over ranks
        timings["_ranks"] = {"per_rank_ms_per_step": [t / a.steps * 1e3 for t in per_rank],
                             "per_rank_ms_per_step_min": min(per_rank) / a.steps * 1e3, "per_rank_ms_per_step_max": max(per_rank) / a.steps * 1e3,
                             "per_rank_gemm_ms_per_step": [e_["gemm_ms_per_step"] for e_ in every],
                             "per_rank_allreduce_ms_per_step": [e_["allreduce_ms_per_step"] for e_ in every],
                             "allreduce_ms_per_step_rank0": (ar["total_ms"] / ns) if ar else None,
                             "allreduce_launches_per_step": (ar["launches"] / ns) if ar else None,
                             "allreduce_bytes_per_step": (ar["bytes"] / ns) if ar else None,
                             "kernel_timings_sampled_every": TIMING_EVERY,
                             "note": "allreduce_ms spans include the wait for the slowest rank to arrive at the exchange"}
    return dt, timings


def pmc_traffic(kernel, planes=4):
    """HBM bytes per launch of `kernel` from the newest committed PMC pass (profiles/*_pmc.json, produced by
    scripts/gpu_profile.sh: separate rocprofv3 --pmc runs).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes
    for gfx950 wide coalesced reads (calibrated on k_snp_stats: 2 x 5120.1 MB = the 10 240 MB it streams)."""
    import glob
    key = {"gemm_GQ_2bit": f"gpca::k_gq_2bit<{planes}>", "gemm_GtT_2bit": f"gpca::k_gtt_p<{planes}>", "gemm_GQ_i8": "gpca::k_gq_d<1, 6>", "gemm_GtT_i8": "gpca::k_gtt_d<1>", "gemm_GQ_f32": "gpca::k_gq_f32<1, false>",
           "gemm_GtT_f32": "gpca::k_gtt_f32<1, false>"}.get(kernel)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")))
    if not key or not files:
        return None, None
    table = json.load(open(files[-1]))
    d = table.get(key)
    if d is None:   # template arguments changed between profiles (k_gq_d<1> -> k_gq_d<1, 6>): match on the kernel's base name
        base = key.split("<")[0]
        cands = [v for k2, v in sorted(table.items(), reverse=True) if k2.split("<")[0] == base and ("true" in k2) == ("true" in key)]
        d = cands[0] if cands else None
    if not d or "FETCH_SIZE" not in d:
        return None, None
    return (2.0 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)) * 1024.0, os.path.basename(files[-1])


DEFAULT_SHAPE = True   # set in main(): the committed PMC traffic figures belong to the default 1M x 10k workload only


# A register-resident v_mfma_i32_32x32x32_i8 loop on random bytes holds the chip at 1.70 GHz and delivers 3 400 TOP/s (profiles/r1_kbench_summary.md
# section 8): what the matrix cores sustain under this operand mix, against the 5 000 TOP/s dense peak at the 2.4 GHz they do not hold.
INT8_MFMA_SUSTAINED_TOPS = 3400.0


def roofline_of(timings, precision, steps, storage="int8", planes=4):
    gq, gt = timings.get("gemm_GQ"), timings.get("gemm_GtT")
    dom_name, dom = max((("gemm_GQ", gq), ("gemm_GtT", gt)), key=lambda kv: kv[1]["total_ms"] if kv[1] else 0.0)
    avg_ms = dom["total_ms"] / dom["launches"]
    tflops = dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12
    gbs = dom["bytes"] / dom["launches"] / (avg_ms * 1e-3) / 1e9
    common = {"kernel": dom_name + (("_2bit" if storage == "2bit" else "_i8") if precision == "i8" else "_f32"), "avg_launch_ms": avg_ms, "launches": dom["launches"],
              "algorithmic_flops_per_launch": dom["flops"] / dom["launches"],
              "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
              "all_kernels_ms_per_step": {n: t["total_ms"] / sampled_steps(steps) for n, t in timings.items()},
              "kernel_timings_sampled_every": TIMING_EVERY}
    traffic, src = pmc_traffic(common["kernel"], planes) if DEFAULT_SHAPE else (None, None)
    common["traffic_source"] = (f"profiles/{src}: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per launch, separate rocprofv3 --pmc passes"
                                if src else None)
    if precision == "i8" and storage == "2bit":
        # 0.25 B per genotype: the exact-integer kernels become matrix-core bound (4 digit planes x 32-cycle int8 MFMAs
        # per 1024 genotypes); peak = dense int8 MFMA rate, 2 x the ~2.5 PF bf16 peak (MI355X_MICROARCH.md)
        ops = 2.0 * 32 * planes * (dom["bytes"] / dom["launches"] * 4)     # executed int8 MACs x 2 per launch (`planes` digit planes, L = 32)
        tops = ops / (avg_ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": tops, "peak": 5000.0, "unit": "TOP/s (int8, executed digit-plane MFMAs)",
                "frac": tops / 5000.0, "frac_of_sustained": tops / INT8_MFMA_SUSTAINED_TOPS, "sustained_peak": INT8_MFMA_SUSTAINED_TOPS,
                "sustained_peak_source": "bare register-resident int8 MFMA loop on random bytes, 1.70 GHz under load (profiles/r1_kbench_summary.md section 8)",
                "bound_detail": "matrix-core issue under operand delivery, not the dense int8 peak: 256 accumulation registers hold 4 tiles x 4 planes, so a "
                                "loaded plane operand feeds 4 MFMAs and Q's planes are re-read by every wave (18 % of the launch, ablated); the clock under "
                                "a dense int8 MFMA stream is 1.7 GHz, not 2.4.  Measured reach of this design: 0.49 of `peak` with the plane operands "
                                "shared through LDS and nothing synchronised (profiles/r4_kbench_summary.md section 3); fp4 x fp6 scaled MFMAs run at "
                                "1.97 x the int8 rate but need 6 planes for 28 bits = 384 accumulators at 4 tiles (profiles/r5_kbench_summary.md section 4)",
                "reach_of_this_design_frac": 0.49,
                "digit_planes": planes, "traffic": traffic, "hbm_GBs_algorithmic": gbs, "algorithmic_TFLOPs_equivalent": tflops, **common}
    if precision == "i8":   # exact-integer MFMA needs ~1/10 of the matrix-core time per byte: HBM-bound
        return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "frac_of_measured_stream_ceiling": gbs / HBM_STREAM_CEILING_GBS, "measured_stream_ceiling": HBM_STREAM_CEILING_GBS,
                "measured_stream_ceiling_source": HBM_STREAM_CEILING_SOURCE,
                "traffic": traffic, "algorithmic_TFLOPs_equivalent": tflops, **common}
    return {"bound": "mfma", "achieved": tflops, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tflops / MFMA_F32_PEAK_TFLOPS, "traffic": traffic, "hbm_GBs_algorithmic": gbs, **common}


def kernel_rooflines(timings, precision, storage):
    """Per-kernel roofline of one run: both GEMMs, each against the resource that bounds it on this path."""
    out = {}
    for name in ("gemm_GQ", "gemm_GtT"):
        t = timings.get(name)
        if not t or not t["launches"]:
            continue
        ms = t["total_ms"] / t["launches"]
        tf = t["flops"] / t["launches"] / (ms * 1e-3) / 1e12
        gbs = t["bytes"] / t["launches"] / (ms * 1e-3) / 1e9
        d = {"avg_launch_ms": ms, "launches": t["launches"], "algorithmic_TFLOPs": tf, "algorithmic_GBs": gbs}
        if precision == "f32":
            d.update(bound="mfma", achieved=tf, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / MFMA_F32_PEAK_TFLOPS)
        elif storage == "2bit":
            tops = 2.0 * 32 * 4 * (t["bytes"] / t["launches"] * 4) / (ms * 1e-3) / 1e12
            d.update(bound="mfma", achieved=tops, peak=5000.0, unit="TOP/s (int8, executed digit-plane MFMAs)", frac=tops / 5000.0)
        else:
            d.update(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS, frac_of_measured_stream_ceiling=gbs / HBM_STREAM_CEILING_GBS)
        out[name] = d
    return out


def extra_resident_line(g, a, name, M, N, k, precision, storage, steps, warmup, device):
    """One more resident workload of BASELINE.json, timed by THIS process (so the driver's run carries it): generated on the
    device (GPCA_PANEL_SYNTH16: 3 populations, Hardy-Weinberg proportions), stats, `warmup` + `steps` gpca_rsvd calls, HIP-event
    rooflines of both GEMMs, and the size-independent properties of the result (there is no oracle at this size)."""
    PREC = {"f32": g._lib.PREC_F32_MFMA, "i8": g._lib.PREC_I8_EXACT}
    t0 = time.perf_counter()
    th16 = g.synth_thresholds16(M, 3, seed=a.rfit_seed)
    eng = g.GpcaEngine(device=device, precision=PREC[precision], storage=g._lib.STORE_2BIT if storage == "2bit" else g._lib.STORE_INT8)
    try:
        eng.load_from_source(g.PanelSource.synth16(th16, a.rfit_seed), M, N)
        del th16
        eng.synchronize()
        t_gen = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.snp_stats(g.QcConfig.none(), fetch=False)
        t_stats = time.perf_counter() - t0
        t_w = time.perf_counter()
        n_w = 0
        while n_w < warmup or time.perf_counter() - t_w < EXTRA_PATH_MIN_WARM_S:       # (an extra path: see EXTRA_PATH_MIN_WARM_S)
            eng.rsvd(k, a.oversample, a.power_iters, a.rfit_seed); n_w += 1
        eng.enable_timings(True); eng.reset_timings()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.rsvd(k, a.oversample, a.power_iters, a.rfit_seed)
        eng.synchronize()
        per_step = (time.perf_counter() - t0) / steps
        tim = eng.timings()
        sc, ev, sv = eng.scores(f64=True), eng.eigenvalues(), eng.singular_values()
        gram = sc.T @ sc
        off = gram - np.diag(np.diag(gram))
        props = {"scores_orthogonality_max_offdiag_rel": float(np.max(np.abs(off)) / sv[0] ** 2),
                 "scores_norms_vs_singular_values_max_rel": float(np.max(np.abs(np.diag(gram) - sv[:k] ** 2) / sv[:k] ** 2)),
                 "centring_max_abs_colsum_rel": float(np.max(np.abs(sc.sum(axis=0))) / np.abs(sc).sum(axis=0).max()),
                 "structured_eigenvalues_found": int(np.sum(ev > 20 * ev[-1])), "structured_eigenvalues_expected": 2,
                 "eigenvalues_descending": bool(np.all(np.diff(ev) <= 0))}
        tr = eng.transform()
        a_, b_ = tr[:, :2] / np.linalg.norm(tr[:, :2], axis=0), sc[:, :2] / np.linalg.norm(sc[:, :2], axis=0)
        a_ = a_ * np.sign(np.sum(a_ * b_, axis=0))      # sign-aligned unit-norm PCs: PCA::transform against the scores of the fit
        props["transform_vs_scores_max_abs_dPC_structured"] = float(np.max(np.abs(a_ - b_)))
        l = k + a.oversample
        return {"workload": name, "snps": M, "samples": N, "k": k, "l": l, "q": a.power_iters, "gemm_path": precision, "residency": storage,
                "steps": steps, "warmup": warmup, "ms_per_step": per_step * 1e3, "value": M * N / per_step, "unit": "SNPs*samples/s",
                "generate_s": t_gen, "snp_stats_s": t_stats, "roofline": kernel_rooflines(tim, precision, storage),
                "all_kernels_ms_per_step": {n_: t_["total_ms"] / steps for n_, t_ in tim.items()},
                "top_eigenvalues": [float(x) for x in ev[:3]], "properties": props}
    finally:
        eng.close()


def extra_config2_line(g, a, device):
    """BASELINE.json configs[2] at its own size: data/chr22_subset50.bed is 1 066 557 SNPs x 64 samples.  /root/reference does not exist
    on the GPU box, so the committed 120 000-SNP slice of that file (tests/golden/chr22_subset50_120k.npz: .bed bytes, data only) is
    tiled to the full row count; --eigensnp defaults (QC 0.98 / 0.01 / 1e-6, k = 20 here, l = 30, q = 2), int8 rows, the narrow kernels."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "chr22_subset50_120k.npz"))
    rows, n = z["bed_rows"], int(z["n_samples"])
    M = 1_066_557
    bed = np.tile(rows, (M // rows.shape[0] + 1, 1))[:M]
    with g.GpcaEngine(device=device, precision=g._lib.PREC_I8_EXACT, storage=g._lib.STORE_INT8) as e:
        t0 = time.perf_counter(); e.upload_bed2bit(bed, n); t_up = time.perf_counter() - t0
        t0 = time.perf_counter(); e.snp_stats(g.QcConfig(), fetch=False); t_qc = time.perf_counter() - t0
        t_w = time.perf_counter()
        e.rsvd(20, 10, 2, seed=2025)
        while time.perf_counter() - t_w < EXTRA_PATH_MIN_WARM_S:                         # (an extra path: see EXTRA_PATH_MIN_WARM_S)
            e.rsvd(20, 10, 2, seed=2025)
        e.enable_timings(True); e.reset_timings(); e.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            e.rsvd(20, 10, 2, seed=2025)
        e.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tim = e.timings()
        n_pca = e.num_pca_snps()
        return {"workload": f"chr22_subset50-shaped: {M} SNPs x {n} samples (.bed rows of the committed slice of the reference's own file, tiled), "
                            f"QC defaults -> {n_pca} PCA SNPs, k = 20, l = 30, q = 2, int8 rows",
                "snps": M, "samples": n, "pca_snps": n_pca, "steps": reps, "ms_per_step": dt * 1e3, "value": M * n / dt, "unit": "SNPs*samples/s",
                "upload_s": t_up, "snp_stats_s": t_qc,
                "gemm_launch_us": {k_: v_["total_ms"] / v_["launches"] * 1e3 for k_, v_ in tim.items() if k_.startswith("gemm")},
                "gemm_ms_per_step": sum(v_["total_ms"] for k_, v_ in tim.items() if k_.startswith("gemm")) / reps,
                "hbm_GBs_algorithmic_per_gemm_launch": {k_: M * n / (v_["total_ms"] / v_["launches"] * 1e-3) / 1e9 for k_, v_ in tim.items() if k_.startswith("gemm")},
                "top_eigenvalues": [float(x) for x in e.eigenvalues()[:3]]}


def streamed_run(g, a, M, N, k, storage, device, snp_offset, rdzv, uid_fn=None, steps=None, warmup=None, cache_gb=0.0, bench_hold=False):
    """One out-of-core job: stats sweep + `steps` timed gpca_rsvd calls over panels that are regenerated on every sweep."""
    steps = a.steps if steps is None else steps
    warmup = a.warmup if warmup is None else warmup
    th16 = g.synth_thresholds16(M, 3, seed=a.rfit_seed, snp_offset=snp_offset)
    eng = g.GpcaEngine(device=device, precision=g._lib.PREC_I8_EXACT, storage=g._lib.STORE_2BIT if storage == "2bit" else g._lib.STORE_INT8,
                       digit_planes=a.digit_planes if storage == "2bit" else 0)
    rank = rdzv.rank if rdzv is not None else 0
    mem = memory_preflight(eng, f"the panel ring of {M} SNPs x {N} samples ({storage}, streamed)",
                           shard_bytes_needed(M, N, storage, k, a.oversample, streamed=True, panel_rows=a.panel_rows, ring=a.ring), rank)
    eng.stream_open(g.PanelSource.synth16(th16, a.rfit_seed, snp_offset=snp_offset, bench_hold=bench_hold), M, N, panel_rows=a.panel_rows, ring_slots=a.ring,
                    fused=not a.unfused)
    del th16
    solo = None
    if uid_fn is not None:
        # this rank's shard as a stream of its own first (no exchange, no panel cache): the weak-scaling reference of this very run
        eng.snp_stats(g.QcConfig.none(), fetch=False)
        eng.rsvd(k, a.oversample, a.power_iters, a.rfit_seed)
        eng.synchronize(); rdzv.barrier()
        t0 = time.perf_counter()
        eng.rsvd(k, a.oversample, a.power_iters, a.rfit_seed)
        eng.synchronize()
        solo = rdzv.allgather(time.perf_counter() - t0)
        uid_fn(eng)            # the communicator first: its device buffers exist before the panel cache sizes itself on what is free
    n_cached = eng.stream_set_cache(-1 if cache_gb < 0 else int(cache_gb * 2**30)) if cache_gb else 0
    t0 = time.perf_counter()
    eng.snp_stats(g.QcConfig.none(), fetch=False)
    t_stats = time.perf_counter() - t0

    def barrier():
        eng.synchronize()              # = hipStreamSynchronize on the engine's streams: the only GPU work of this process
        if rdzv is not None:
            rdzv.barrier()
    for _ in range(warmup):
        eng.rsvd(k, a.oversample, a.power_iters, a.rfit_seed)
    eng.enable_timings(True); eng.reset_timings()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.rsvd(k, a.oversample, a.power_iters, a.rfit_seed)
    barrier()
    dt = time.perf_counter() - t0
    tim = eng.timings()
    if rdzv is not None:
        ar = tim.get("allreduce")
        every = rdzv.allgather({"dt": dt, "allreduce_ms_per_step": (ar["total_ms"] / steps) if ar else None})
        per_rank = [e_["dt"] for e_ in every]
        dt = max(per_rank)                         # the contract's max over ranks
        ranks_seen = eng.comm_count_ranks()
        tim["_ranks"] = {"per_rank_ms_per_step": [t / steps * 1e3 for t in per_rank],
                         "per_rank_ms_per_step_min": min(per_rank) / steps * 1e3, "per_rank_ms_per_step_max": max(per_rank) / steps * 1e3,
                         "per_rank_allreduce_ms_per_step": [e_["allreduce_ms_per_step"] for e_ in every],
                         "same_shard_without_exchange_ms_per_step_per_rank": [t * 1e3 for t in solo] if solo else None,
                         ("ranks_seen_by_rccl" if a.exchange == "rccl" else "ranks_seen_by_the_host_hook"): ranks_seen,
                         "exchange": ("ncclAllReduce (RCCL) on the engine's stream inside libgpca.so" if a.exchange == "rccl" else
                                      "host-staged hook over the launcher's hub (rehearsal: not the product's transport)"),
                         "note": "the solo reference streams without the panel cache; allreduce spans include the wait for the slowest rank"}
        tim["_ranks"].update(weak_scaling_fields(dt / steps * 1e3, tim["_ranks"]["same_shard_without_exchange_ms_per_step_per_rank"] or []))
    ev = eng.eigenvalues()
    eng.close()
    tim["_panels_cached"] = n_cached
    tim["_memory"] = mem
    return dt, tim, ev, t_stats


def streamed_summary(tim, steps, M, N, l, storage):
    """Per-step times of the streamed sweeps.  gemm_* records span a whole sweep over the panels on the compute stream (waits
    for panels included); panel_fill records are the generator's own kernel times on the fill stream."""
    fill = tim.get("panel_fill", {"total_ms": 0.0, "launches": 0, "bytes": 0.0})
    zero = {"total_ms": 0.0, "launches": 0}
    gq, gt, gf = tim.get("gemm_GQ", zero), tim.get("gemm_GtT", zero), tim.get("gemm_fused", zero)
    sweeps = (gq["launches"] + gt["launches"] + gf["launches"]) / steps      # = passes over the source per call
    per_b = 0.25 if storage == "2bit" else 1.0
    gemm_ms = (gq["total_ms"] + gt["total_ms"] + gf["total_ms"]) / steps
    return {"sweeps_per_step": sweeps, "panel_fills_per_step": fill["launches"] / steps,
            "generator_ms_per_step": fill["total_ms"] / steps,
            "generator_rate_genotypes_per_s": (fill["bytes"] / (fill["total_ms"] * 1e-3)) if fill["total_ms"] else None,
            "gemm_sweeps_ms_per_step": gemm_ms, "gemm_GQ_ms_per_sweep": gq["total_ms"] / max(gq["launches"], 1),
            "gemm_GtT_ms_per_sweep": gt["total_ms"] / max(gt["launches"], 1),
            "fused_power_iteration_ms_per_sweep": (gf["total_ms"] / gf["launches"]) if gf["launches"] else None,
            "hbm_GBs_per_sweep_algorithmic": M * N * per_b * (1 if l <= 32 else 2) / (gemm_ms / sweeps * 1e-3) / 1e9,
            "note": "fills run one panel ahead on a second stream; a sweep's span includes any wait for the generator"}


def streamed_main(a, g, rank, world, local_rank, rdzv):
    M_local, N, k = a.snps, a.samples, a.components
    l = k + a.oversample
    snp_offset = rank * M_local
    uid_fn = None
    if rdzv is not None:
        def uid_fn(eng):
            connect(g, a, eng, rdzv, rank, world, snp_offset)
    dt, tim, ev, t_stats = streamed_run(g, a, M_local, N, k, a.storage, local_rank, snp_offset, rdzv, uid_fn, cache_gb=a.cache_gb, bench_hold=a.bench_hold)
    n_cached = tim.pop("_panels_cached")
    mem = tim.pop("_memory", None)
    rank_info = tim.pop("_ranks", None)
    if rank == 0:
        M_total = M_local * world
        per_step = dt / a.steps
        ssum = streamed_summary(tim, a.steps, M_local, N, l, a.storage)
        # dominant "kernel" of a streamed run = one sweep of the slower GEMM over all panels of this rank
        dom = max((n_ for n_ in ("gemm_GQ", "gemm_GtT", "gemm_fused") if n_ in tim), key=lambda n_: tim[n_]["total_ms"] / tim[n_]["launches"])
        sweep_ms = tim[dom]["total_ms"] / tim[dom]["launches"]
        by = tim[dom]["bytes"] / tim[dom]["launches"]
        out = {"metric": "SNPs x samples / sec through rSVD at k=%d (streamed panels); max|dPC| vs ref" % k,
               "value": M_total * N / per_step, "unit": "SNPs*samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": per_step * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "i8 (exact-integer GEMMs, f64 recombination)", "data": "synthetic (device generator, regenerated on every sweep)",
               "config": {"workload": f"out-of-core: synthetic {M_total} SNPs x {N} samples (never resident), k={k}, l={l}, q={a.power_iters}, "
                                      f"panels generated on the device by a SplitMix64 counter generator (GPCA_PANEL_SYNTH16) into a ring of {a.ring} HBM buffers"
                                      + (f"; the leading {n_cached} panels stay in spare HBM (gpca_stream_set_cache), the others are regenerated on every pass" if n_cached else ""),
                          "snps_per_gpu": M_local, "samples": N, "k": k, "oversample": a.oversample, "power_iters": a.power_iters,
                          "parallelism": f"snp-row-shards x{world}", "gemm_path": "i8", "residency": f"streamed/{a.storage}",
                          "panel_rows": a.panel_rows, "ring": a.ring, "passes_over_the_source_per_call": ssum["sweeps_per_step"],
                          "panels_cached_in_hbm": n_cached},
               "roofline": {"bound": "hbm" if a.storage == "int8" else "mfma", "kernel": dom + " sweep over all panels",
                            "avg_launch_ms": sweep_ms, "achieved": by / (sweep_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": by / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                            "note": "sweep-level (all panels, waits for the generator included); the per-launch roofline of the same kernels is "
                                    "the resident bench line"},
               "streaming": ssum, "snp_stats_s": t_stats, "top_eigenvalues": [float(x) for x in ev[:3]], "cpu_baseline": None,
               "device_memory_preflight": mem}
        if rank_info is not None:
            out["multi_gpu"] = rank_info
            out["config"]["workload"] += ("; weak scaling vs the same shard streamed on one GPU in this run (multi_gpu.weak_scaling_efficiency), "
                                          "NOT vs a --gpus 1 line")
        print(json.dumps(out), flush=True)
    if rdzv is not None:
        rdzv.barrier()
        rdzv.close()


def workload_name(world, M_local, M_total, N, k, l, a):
    tail = f"k={k}, l={l}, q={a.power_iters}, seed={a.rfit_seed}, resident in HBM"
    if world == 1:
        return f"synthetic {M_total} SNPs x {N} samples int8 genotypes (3 populations, F_ST 0.05), {tail}"
    what = ("BASELINE.json configs[3]: " if (M_total, N) == (10_000_000, 100_000) else
            "BASELINE.json configs[3]'s per-GPU shard (1.25M x 100k) on every rank, weak scaling: " if (M_local, N) == (1_250_000, 100_000) else "")
    return (f"{what}synthetic {M_total} SNPs x {N} samples int8 genotypes (3 populations, Hardy-Weinberg proportions, device generator), SNP rows "
            f"sharded {M_local} per GPU over {world} GPUs, the N x l sketch and one l x l Gram all-reduced per call, {tail}; weak scaling vs the "
            f"same shard on one GPU in this run (multi_gpu.weak_scaling_efficiency), NOT vs the --gpus 1 line, which is configs[1]")


def connect(g, a, eng, rdzv, rank, world, snp_offset):
    """Join this rank's engine to the sharded matrix: RCCL inside libgpca.so (rank 0 draws the unique id, the launcher's hub hands its
    128 bytes round), or -- `--exchange host`, the rehearsal for boxes with fewer GPUs than ranks -- the host-staged hook over the hub."""
    if a.exchange == "host":
        eng.set_allreduce_hook(rdzv.allreduce_hook(), world, rank, snp_offset)
    else:
        uid = g.distributed.broadcast_unique_id(g.GpcaEngine, rank, rdzv=rdzv)
        eng.comm_init(world, rank, uid, snp_offset)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--snps", type=int, default=None, help="SNP rows per GPU (default: 1 000 000 on one GPU = configs[1]; 1 250 000 per rank on "
                                                           "several = configs[3]'s per-GPU shard)")
    ap.add_argument("--samples", type=int, default=None, help="default: 10 000 on one GPU, 100 000 per rank on several")
    ap.add_argument("--components", "-k", type=int, default=20)
    ap.add_argument("--oversample", type=int, default=10)
    ap.add_argument("--power-iters", type=int, default=2)
    ap.add_argument("--rfit-seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-path", action="store_true", help="skip the extra f32-MFMA measurement")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra BASELINE.json workloads of the default one-GPU run (configs[3]'s per-GPU shard 1.25M x 100k int8 and "
                         "north_star's literal 10M x 100k MFMA-fp32 job on 2-bit rows, configs[4]'s per-GPU shard streamed out of core): use it under rocprofv3, whose per-kernel averages "
                         "would otherwise mix three shapes")
    ap.add_argument("--storage", default="int8", choices=["int8", "2bit"],
                    help="HBM residency of the genotypes: int8 = 1 B/genotype (the BASELINE.json configs), 2bit = 0.25 B, decoded in the GEMM prologues")
    ap.add_argument("--digit-planes", type=int, default=0, choices=[0, 3, 4],
                    help="exact path: 4 signed base-128 digit planes or 3 base-256 planes (24-bit; --storage 2bit only); 0 = the library's choice "
                         "(4 on int8 rows, 3 on 2-bit rows)")
    ap.add_argument("--precision", default="i8", choices=["f32", "i8"],
                    help="i8 = exact-integer GEMMs (default, fastest parity-green path); f32 = v_mfma_f32_32x32x2_f32")
    ap.add_argument("--streamed", action="store_true",
                    help="BASELINE.json configs[4] mode: the matrix is never resident; panels come from the device generator "
                         "(GPCA_PANEL_SYNTH16) through a ring of HBM buffers, one panel ahead of the GEMMs")
    ap.add_argument("--streamed-extra", action="store_true", help="resident run: also time the same shape out-of-core (8 panels)")
    ap.add_argument("--panel-rows", type=int, default=0, help="--streamed: SNP rows per panel (0 = 131072 rows or what fits)")
    ap.add_argument("--ring", type=int, default=3, help="--streamed: panel buffers in the ring")
    ap.add_argument("--cache-gb", type=float, default=-1.0,
                    help="--streamed: GiB of spare HBM that keep the leading panels resident (gpca_stream_set_cache); -1 = what is free, 0 = none")
    ap.add_argument("--unfused", action="store_true", help="--streamed: 6 passes per call (bit-identical to the resident engine) instead of 4")
    ap.add_argument("--bench-hold", action="store_true",
                    help="--streamed, MEASUREMENT ONLY (GPCA_SOURCE_BENCH_HOLD): buffers keep the panel they were given first, nothing is regenerated; "
                         "the engine's own rate on streamed panels, meaningless eigenvalues")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "host"],
                    help="N > 1: rccl = ncclAllReduce on the engine's stream inside libgpca.so (the product path); host = the host-staged hook over the "
                         "launcher's hub, a rehearsal of everything but RCCL for boxes with fewer GPUs than ranks")
    ap.add_argument("--one-device", action="store_true", help="N > 1 rehearsal: every rank on device 0 (needs --exchange host: RCCL refuses two ranks on one GPU)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: this process becomes the parent of N ranks.  It makes no GPU call and never re-execs; it
        # hosts the rendezvous hub, starts this same command once per rank (RANK / LOCAL_RANK / WORLD_SIZE in the environment -- the
        # variables torch.distributed.run would set) and leaves with a non-zero code if any rank does.
        from genomic_pca_amd import launch
        codes = launch.run_ranks(a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                 local_ranks=[0] * a.gpus if a.one_device else None)
        if any(codes):
            print(f"bench.py: rank exit codes {codes}", file=sys.stderr, flush=True)
        sys.exit(launch.exit_code(codes))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if a.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(a.gpus, 1):
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if world > 1:
        # (ranks started by torch.distributed.run do not pass through launch.run_ranks, which sets this for its children.  This pool's host
        #  driver supports dmabuf IPC only; with the legacy mode RCCL's ncclCommInitRank fails in hipIpcGetMemHandle.  Before any HIP call;
        #  a value the caller exported wins.)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # No torch anywhere in this harness: libgpca.so carries its own RCCL exchange, and what the host side of a multi-rank run needs (the
    # unique id handed round, a barrier on both sides of the timed region, the max over ranks) goes through genomic_pca_amd/launch.py's hub
    # -- also when torch.distributed.run started the ranks (rank 0 then hosts the hub).  One HIP runtime per process, whatever the launcher.
    import genomic_pca_amd as g
    from genomic_pca_amd import launch
    rdzv = launch.from_env()
    if a.one_device and a.exchange != "host" and world > 1:
        sys.exit("bench.py: --one-device needs --exchange host (RCCL refuses two ranks on one GPU)")
    if a.snps is None:
        a.snps = 1_000_000 if world == 1 else 1_250_000
    if a.samples is None:
        a.samples = 10_000 if world == 1 else 100_000

    if a.streamed:
        return streamed_main(a, g, rank, world, local_rank, rdzv)
    M_local, N, k = a.snps, a.samples, a.components
    global DEFAULT_SHAPE
    DEFAULT_SHAPE = (M_local, N, k) == (1_000_000, 10_000, 20)
    l = k + a.oversample
    if a.precision == "i8" and l > 64:
        a.precision = "f32"      # (neither path goes beyond 64 columns; the f32 one gives the clearer error)
    M_total = M_local * world
    snp_offset = rank * M_local
    PREC = {"f32": g._lib.PREC_F32_MFMA, "i8": g._lib.PREC_I8_EXACT}

    engines = []

    def barrier():
        for e_ in engines:          # = hipStreamSynchronize on the engine's streams (the only GPU work of this process)
            e_.synchronize()
        if rdzv is not None:
            rdzv.barrier()

    results = {}
    # the extra measurements (f32-MFMA path, 2-bit residency) are single-GPU information: multi-rank runs time the headline only
    extras = not a.no_second_path and world == 1
    order = [a.precision] + ([] if (not extras or a.precision == "f32" or l > 32) else ["f32"])
    if a.precision == "i8" and a.storage == "int8" and extras:
        order.append("i8_2bit")
        order.append("i8_2bit_4p")
    # one GPU: the Philox generator the oracle restates (gpca_synth_genotypes); several: the fast device generator (one 16-bit uniform per
    # genotype, SplitMix64 in counter mode), which fills a 125 GB shard in a fraction of a second -- both draw row i by its GLOBAL index
    big = world > 1 or M_local * N > 4 * 10**10
    th = (g.synth_thresholds16 if big else g.synth_thresholds)(M_local, 3, seed=a.rfit_seed, snp_offset=snp_offset)
    t_stats = None
    solo = None
    head_rank_info = None
    extra_errors = {}

    def one_path(prec):
        nonlocal t_stats, head_rank_info, solo
        packed = prec in ("i8_2bit", "i8_2bit_4p")
        store = g._lib.STORE_2BIT if ((a.storage == "2bit" and prec == a.precision) or packed) else g._lib.STORE_INT8
        planes = 4 if prec == "i8_2bit_4p" else (a.digit_planes if (prec == "i8" and store == g._lib.STORE_2BIT) else 0)
        eng = g.GpcaEngine(device=local_rank, precision=PREC["i8" if packed else prec], storage=store, digit_planes=planes)
        try:
            engines[:] = [eng]
            if prec == a.precision:     # the headline shard must fit BEFORE anything is generated (extras report their own errors)
                results["_memory"] = memory_preflight(eng, f"the resident shard of {M_local} SNPs x {N} samples ({'2bit' if store == g._lib.STORE_2BIT else 'int8'})",
                                                      shard_bytes_needed(M_local, N, "2bit" if store == g._lib.STORE_2BIT else "int8", k, a.oversample), rank)
            if big:
                eng.load_from_source(g.PanelSource.synth16(th, a.rfit_seed, snp_offset=snp_offset), M_local, N)
            else:
                eng.synth_genotypes(M_local, N, a.rfit_seed, th, snp_offset=snp_offset)
            t0 = time.perf_counter()
            eng.snp_stats(g.QcConfig.none(), fetch=False)
            if t_stats is None:
                t_stats = time.perf_counter() - t0
            if rdzv is not None:
                # this rank's shard as a matrix of its own first (no exchange, a few calls): what the same GPU does without peers, measured
                # in the same process minutes apart -- the weak-scaling reference of this very run
                eng.rsvd(k, a.oversample, a.power_iters, a.rfit_seed)
                eng.synchronize(); rdzv.barrier()
                t0 = time.perf_counter()
                for _ in range(3):
                    eng.rsvd(k, a.oversample, a.power_iters, a.rfit_seed)
                eng.synchronize()
                solo = rdzv.allgather((time.perf_counter() - t0) / 3)
                connect(g, a, eng, rdzv, rank, world, snp_offset)
            ranks_seen = eng.comm_count_ranks() if rdzv is not None else 1   # a 1.0 per rank through libgpca's own communicator (or the hook)
            dt, timings = timed_run(eng, a, k, barrier, rdzv, min_warm_s=0.0 if prec == a.precision else EXTRA_PATH_MIN_WARM_S)
            rank_info = timings.pop("_ranks", None)
            if rank_info is not None:
                rank_info["ranks_seen_by_rccl" if a.exchange == "rccl" else "ranks_seen_by_the_host_hook"] = ranks_seen
                rank_info["exchange"] = ("ncclAllReduce (RCCL) on the engine's stream inside libgpca.so" if a.exchange == "rccl" else
                                         "host-staged hook over the launcher's hub (rehearsal: not the product's transport)")
                rank_info["same_shard_without_exchange_ms_per_step_per_rank"] = [t * 1e3 for t in solo] if solo else None
                rank_info["exchange_volume_per_step"] = (f"{a.power_iters + 1} x {N} x {32 if l <= 32 else 64} f64 (sketch) + one "
                                                         f"{32 if l <= 32 else 64}^2 + 16 f64 (Gram + status) + 16 f64 (status), in-place all-reduce")
                rank_info["launcher"] = "torch.distributed.run (ranks given)" if "GPCA_RDZV" not in os.environ else "bench.py's own (genomic_pca_amd/launch.py)"
                rank_info.update(weak_scaling_fields(dt / a.steps * 1e3, rank_info["same_shard_without_exchange_ms_per_step_per_rank"] or []))
            results[prec] = (dt, timings, eng.eigenvalues())
            if prec == a.precision:
                head_rank_info = rank_info
        finally:
            engines[:] = []
            eng.close()

    for prec in order:
        if prec == a.precision:
            one_path(prec)                # the headline: a failure here is the run's failure
        else:
            try:                          # the other paths of the same job are extras (single GPU only): reported, never fatal
                one_path(prec)
            except Exception as ex:       # noqa: BLE001
                extra_errors[prec] = f"{type(ex).__name__}: {ex}"
    del th

    if rank == 0:
        dt, timings, ev = results[a.precision]
        per_step = dt / a.steps
        value = M_total * N / per_step
        dtype = ("i8 (exact: int8 dosages x 4 signed 7-bit fixed-point digit planes of the skinny operand, i32 accumulate, "
                 "f64 recombination; accuracy >= f32)") if a.precision == "i8" else "f32"
        out = {
            "metric": "SNPs x samples / sec through rSVD at k=20; max|dPC| vs ref", "value": value, "unit": "SNPs*samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": per_step * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload_name(world, M_local, M_total, N, k, l, a),
                       "snps_per_gpu": M_local, "samples": N, "k": k, "oversample": a.oversample,
                       "power_iters": a.power_iters, "parallelism": f"snp-row-shards x{world}", "gemm_path": a.precision,
                       "residency": a.storage},
            "roofline": roofline_of(timings, a.precision, a.steps, a.storage, planes=(a.digit_planes or 3) if a.storage == "2bit" else 4),
            "snp_stats_s": t_stats,
            "top_eigenvalues": [float(x) for x in ev[:3]],
        }
        # SURVEY.md 8(d) counts a call as 4 reads of the matrix (sketch, two fused power iterations, projection) = 4 B per int8 genotype; the
        # resident engine makes 6 sweeps (K1 and K2 of a power iteration are separate launches: DESIGN.md section 3), so the step-level figure
        # sits below the per-launch one by that 4/6 and by the share of the call outside the GEMMs
        per_b = 0.25 if a.storage == "2bit" else 1.0
        step_bytes = 4.0 * per_b * M_local * N
        out["roofline"]["step_level"] = {"algorithmic_bytes": step_bytes, "definition": "SURVEY.md 8(d): 4 passes x %.2f B per genotype per call, per GPU" % per_b,
                                         "achieved_GBs": step_bytes / per_step / 1e9, "frac": step_bytes / per_step / 1e9 / HBM_PEAK_GBS,
                                         "sweeps_the_engine_makes": 2 + 2 * a.power_iters}
        out["device_memory_preflight"] = results.get("_memory")
        if head_rank_info is not None:
            out["multi_gpu"] = head_rank_info
        if "f32" in results and a.precision != "f32":
            dt2, tim2, ev2 = results["f32"]
            out["f32_mfma_path"] = {"value": M_total * N / (dt2 / a.steps), "unit": "SNPs*samples/s", "ms_per_step": dt2 / a.steps * 1e3,
                                    "dtype": "f32", "roofline": roofline_of(tim2, "f32", a.steps),
                                    "max_rel_d_eigenvalue_vs_default_path": float(np.max(np.abs(ev2 - ev) / ev))}
        if "i8_2bit" in results:
            dt3, tim3, ev3 = results["i8_2bit"]
            out["packed_2bit_residency"] = {
                "note": "same job, same exact-integer arithmetic, genotypes resident as 2-bit dosage codes (0.25 B each, decoded in the GEMM prologue); "
                        "the library's default for 2-bit rows: three signed base-256 digit planes (24-bit fixed point per column, exact integer "
                        "accumulation; max|dPC| <= 3e-7 against the f64 checker on every parity shape, profiles/r3_planes3_parity.json)",
                "value": M_total * N / (dt3 / a.steps), "unit": "SNPs*samples/s", "ms_per_step": dt3 / a.steps * 1e3,
                "roofline": roofline_of(tim3, "i8", a.steps, "2bit", planes=3),
                "max_rel_d_eigenvalue_vs_default_path": float(np.max(np.abs(ev3 - ev) / ev))}
        if "i8_2bit_4p" in results:
            dt4, tim4, ev4 = results["i8_2bit_4p"]
            out["packed_2bit_four_planes"] = {
                "note": "2-bit residency with gpca_config.digit_planes = 4: four signed base-128 digit planes (28-bit fixed point per column), "
                        "bit-compatible with int8 residency",
                "value": M_total * N / (dt4 / a.steps), "unit": "SNPs*samples/s", "ms_per_step": dt4 / a.steps * 1e3,
                "roofline": roofline_of(tim4, "i8", a.steps, "2bit", planes=4),
                "max_rel_d_eigenvalue_vs_default_path": float(np.max(np.abs(ev4 - ev) / ev))}
        if a.streamed_extra and a.precision == "i8":
            # (not in the default run: its panel launches use the same kernels as the headline and would blur the per-kernel
            #  averages of a rocprofv3 --stats run of this command)  the headline matrix shape again, never resident: 8 panels of 131 072 rows regenerated by the device generator on every sweep
            a2 = argparse.Namespace(**vars(a)); a2.panel_rows = 131072; a2.ring = 3; a2.digit_planes = 0
            dts, tims, evs, _ = streamed_run(g, a2, M_local, N, k, a.storage, local_rank, 0, None, steps=min(a.steps, 3), warmup=1)
            tims.pop("_panels_cached"); tims.pop("_memory", None)
            out["streamed_panels"] = {"note": "same shape out-of-core (BASELINE.json configs[4] mode at configs[1] size): panels come from the "
                                              "device generator (GPCA_PANEL_SYNTH16; a different synthetic draw than the resident matrix)",
                                      "value": M_local * N / (dts / min(a.steps, 3)), "unit": "SNPs*samples/s",
                                      "ms_per_step": dts / min(a.steps, 3) * 1e3, **streamed_summary(tims, min(a.steps, 3), M_local, N, l, a.storage)}
        if world == 1 and DEFAULT_SHAPE and not a.no_extras and not a.no_second_path and a.precision == "i8" and a.storage == "int8":
            # the other single-GPU workloads BASELINE.json names, timed inside this (the driver's) run.  Each is an extra: one that
            # cannot run here (a device that is not all ours, say: the 10M x 100k matrix takes 251 of the 288 GB) is reported as an
            # error under its key and never costs the headline line
            def extra(key, fn, *args, **kw):
                try:
                    out[key] = fn(*args, **kw)
                except Exception as ex:   # noqa: BLE001
                    out[key] = {"error": f"{type(ex).__name__}: {ex}"}
            extra("config3_per_gpu_shard", extra_resident_line,
                  g, a, "BASELINE.json configs[3] per-GPU shard: 1.25M SNPs x 100k samples int8 (125 GB resident), exact-integer GEMMs",
                  1_250_000, 100_000, k, "i8", "int8", steps=5, warmup=1, device=local_rank)
            extra("north_star_literal", extra_resident_line,
                  g, a, "north_star's literal target: 10M SNPs x 100k samples, k = 20, MFMA-fp32 GEMMs on ONE MI355X (2-bit resident rows, 250 GB)",
                  10_000_000, 100_000, k, "f32", "2bit", steps=2, warmup=1, device=local_rank)
            extra("config2_chr22_shape", extra_config2_line, g, a, local_rank)
            # configs[4]'s per-GPU shard, out of core: 6.25M SNPs x 500k samples, k = 40, 2-bit panels from the device generator through
            # the ring, the leading panels kept in spare HBM (one call: the first pass's workspace allocation is noise next to ~7 s)
            def config5_line():
              a5 = argparse.Namespace(**vars(a)); a5.panel_rows = 0; a5.ring = 3; a5.unfused = False; a5.digit_planes = 0
              M5, N5, k5 = 6_250_000, 500_000, 40
              dt5, tim5, ev5, t_stats5 = streamed_run(g, a5, M5, N5, k5, "2bit", local_rank, 0, None, steps=1, warmup=0, cache_gb=-1.0)
              n_cached5 = tim5.pop("_panels_cached"); tim5.pop("_memory", None)
              # the same call with the generator out of the way (GPCA_SOURCE_BENCH_HOLD: the ring's buffers keep the panels they were given
              # first, nothing is regenerated): what the ENGINE does with streamed panels of this shape.  Its eigenvalues mean nothing.
              dt5h, tim5h, _, _ = streamed_run(g, a5, M5, N5, k5, "2bit", local_rank, 0, None, steps=1, warmup=1, cache_gb=-1.0, bench_hold=True)
              tim5h.pop("_panels_cached"); tim5h.pop("_memory", None)
              hidden = streamed_summary(tim5h, 1, M5, N5, k5 + a.oversample, "2bit")
              return {
                "engine_rate_with_fills_hidden": {
                    "definition": "the same call, same ring and HBM panel cache, with GPCA_SOURCE_BENCH_HOLD: no panel is generated in the timed call "
                                  "(every buffer keeps the panel it was given first; the eigenvalues mean nothing): the engine's own rate on streamed "
                                  "panels of this shape.  The real line above shares the GPU with the synthetic generator (136 fills x ~24 ms of its "
                                  "kernels per step), which slows every GEMM launch beside it: that line is a lower bound on the engine, this one is "
                                  "what a source that costs the GPU nothing (host link, disk) would see",
                    "ms_per_step": dt5h * 1e3, "value": M5 * N5 / dt5h, "unit": "SNPs*samples/s",
                    "gemm_sweeps_ms_per_step": hidden["gemm_sweeps_ms_per_step"],
                    "hbm_GBs_per_sweep_algorithmic": hidden["hbm_GBs_per_sweep_algorithmic"]},
                "workload": "BASELINE.json configs[4] per-GPU shard, out of core: 6.25M SNPs x 500k samples (781 GB of 2-bit codes per pass, never "
                            "resident), k = 40, l = 50, panels of 131 072 rows from the device generator (GPCA_PANEL_SYNTH16) through a ring of 3",
                "snps": M5, "samples": N5, "k": k5, "steps": 1, "warmup": 0, "ms_per_step": dt5 * 1e3, "value": M5 * N5 / dt5,
                "unit": "SNPs*samples/s", "snp_stats_s": t_stats5, "panels_cached_in_hbm": n_cached5,
                "streaming": streamed_summary(tim5, 1, M5, N5, k5 + a.oversample, "2bit"),
                "top_eigenvalues": [float(x) for x in ev5[:3]],
                "properties": {"eigenvalues_descending": bool(np.all(np.diff(ev5) <= 0)), "structured_eigenvalues_found": int(np.sum(ev5 > 20 * ev5[-1])),
                               "structured_eigenvalues_expected": 2}}
            extra("config5_per_gpu_shard_streamed", config5_line)
        # bench lines of the other BASELINE.json configs, measured with this build by the scripts named in DESIGN.md (too large or too
        # long for the default run; each file holds one line in this same format)
        out["see_also"] = {k_: v_ for k_, v_ in {
            "this command on one box this round: bench line, rocprofv3 --kernel-trace --stats, PMC passes": "profiles/r5a_summary.md",
            "one call of that run launch by launch": "profiles/r5a_call_timeline.md",
            "A/B and ablation evidence behind the round's kernel changes": "profiles/r5_kbench_summary.md",
            "the same for round 4's kernels": "profiles/r4_kbench_summary.md",
            "10M x 100k on ONE GPU, 2-bit rows, exact path": "profiles/r2_bench_10Mx100k_2bit_one_gpu.json",
            "configs[2] end to end through the command line": "profiles/r2_cli_config3_end_to_end.json",
            "host panel sources at link rate (32 GB matrix)": "profiles/r3_stream_host_link_rates_32GB.jsonl",
        }.items() if os.path.exists(os.path.join(ROOT, v_))}
        if extra_errors:
            out["extra_path_errors"] = extra_errors
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["parity"] = parity_check(g, PREC[a.precision], a.rfit_seed)
            except Exception as ex:       # noqa: BLE001
                out["parity"] = {"error": f"{type(ex).__name__}: {ex}"}
            try:
                out["cpu_baseline"] = cpu_baseline(N, k, a.oversample, a.power_iters, a.rfit_seed)
            except Exception as ex:       # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(ex).__name__}: {ex}"}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if rdzv is not None:
        rdzv.barrier()
        rdzv.close()


if __name__ == "__main__":
    main()
