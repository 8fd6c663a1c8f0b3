"""`python -m genomic_pca_amd` -- the reference's CLI surface (main.rs:501-593) over the MI355X engine.

Two workflows, dispatched on --eigensnp like main.rs:109-122:
  * VCF  (run_vcf_workflow, main.rs:133-247):  --vcf-dir D -k K [--maf f] [--rfit-seed s] --out P
        -> P.vcf.pca.tsv, P.eigenvalues.tsv (header only, as main.rs:676 leaves the vector empty;
           --write-eigenvalues is an extension that fills it)
  * BED  (run_eigensnp_rust_workflow, main.rs:250-442):  --eigensnp --bed-file B --ld-block-file L --out P [--eigensnp-*]
        -> P.eigensnp.pca.tsv, P.eigenvalues.tsv, P.eigensnp.loadings.tsv
EigenSNP's per-LD-block local stage is defined only in the un-vendored efficient_pca crate; this CLI runs the global
randomized PCA over all SNPs that pass QC and fall in an LD block (identical to the reference's own README usage of a
single genome-wide block) by default; --gpca-eigensnp-local-stage runs the multi-stage algorithm the local-stage / refine
flags parameterise, as published for that crate (parity unpinned).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

from . import io as gio
from .engine import (EigenSNPCoreAlgorithm, EigenSNPCoreAlgorithmConfig, GpcaEngine, LdBlockSpecification,
                     MicroarrayGenotypeAccessor, PCA, QcConfig)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="genomic_pca", description="Genomic PCA Tool from VCF or BED/LD-block files.")
    p.add_argument("-o", "--out", dest="output_prefix", required=True, help="Output file prefix.")
    p.add_argument("-t", "--threads", type=int, default=None, help="accepted for compatibility (the GPU does the work)")
    p.add_argument("--log-level", default="Info")
    p.add_argument("-d", "--vcf-dir", default=None)
    p.add_argument("-k", "--components", type=int, default=None,
                   help="Number of principal components to compute (for VCF workflow); at most 118 here (the sketch holds components + 10 <= 128 "
                        "columns, gpca.h; the reference clamps only to min(samples, variants), main.rs:621-628)")
    p.add_argument("--maf", type=float, default=None)
    p.add_argument("--rfit-seed", type=int, default=None)
    p.add_argument("--eigensnp", action="store_true")
    p.add_argument("--bed-file", default=None)
    p.add_argument("--ld-block-file", default=None)
    p.add_argument("--eigensnp-sample-keep-file", default=None)
    # clap's effective defaults when --eigensnp is given (main.rs:545-588)
    p.add_argument("--eigensnp-min-call-rate", type=float, default=0.98)
    p.add_argument("--eigensnp-min-maf", type=float, default=0.01)
    p.add_argument("--eigensnp-max-hwe-p", type=float, default=1e-6)
    p.add_argument("--eigensnp-k-global", type=int, default=10)
    p.add_argument("--eigensnp-components-per-block", type=int, default=7)
    p.add_argument("--eigensnp-subset-factor", type=float, default=0.075)
    p.add_argument("--eigensnp-min-subset-size", type=int, default=10000)
    p.add_argument("--eigensnp-max-subset-size", type=int, default=40000)
    p.add_argument("--eigensnp-global-oversampling", type=int, default=10)
    p.add_argument("--eigensnp-global-power-iter", type=int, default=2)
    p.add_argument("--eigensnp-local-oversampling", type=int, default=10)
    p.add_argument("--eigensnp-local-power-iter", type=int, default=2)
    p.add_argument("--eigensnp-seed", type=int, default=2025)
    p.add_argument("--eigensnp-snp-strip-size", type=int, default=2000)
    p.add_argument("--eigensnp-refine-passes", type=int, default=1)
    p.add_argument("--eigensnp-collect-diagnostics", action="store_true")
    # extensions
    p.add_argument("--device", type=int, default=-1, help="HIP device ordinal")
    p.add_argument("--write-eigenvalues", action="store_true", help="VCF workflow: fill P.eigenvalues.tsv (the reference leaves it header-only)")
    # engine selection (extensions; the defaults are the path bench.py's headline times)
    p.add_argument("--gpca-precision", default="i8", choices=["i8", "f32"], help="i8 = exact-integer GEMMs (default); f32 = f32 matrix cores")
    p.add_argument("--gpca-storage", default="auto", choices=["auto", "int8", "2bit"],
                   help="HBM residency of the genotypes: int8 (1 B per genotype) or 2bit (PLINK-style codes, 0.25 B, faster from 1 024 "
                        "samples up); auto = 2bit for a .bed with at least 1 024 samples, int8 otherwise")
    p.add_argument("--gpca-stream", default="auto", choices=["auto", "on", "off"],
                   help="EigenSNP workflow: walk the .bed out of core through a ring of HBM panels instead of holding it resident "
                        "(auto = when the resident load runs out of device memory; needs --gpca-precision i8)")
    p.add_argument("--gpca-panel-rows", type=int, default=0, help="--gpca-stream: SNP rows per panel (0 = engine default)")
    p.add_argument("--gpca-rfit-power-iters", type=int, default=2,
                   help="VCF workflow: power iterations of the randomized PCA (PCA::rfit's own count is fixed inside the un-vendored "
                        "efficient_pca crate and unknown; 2 = the EigenSNP default, main.rs:318; 4 puts the PCs within 2e-6 of exact PCA)")
    p.add_argument("--gpca-eigensnp-local-stage", action="store_true",
                   help="EigenSNP workflow: run the multi-stage algorithm the --eigensnp-* local / refine flags parameterise (per-block "
                        "local bases on a sample subset, condensed features, global PCA, refinement) instead of one global randomized "
                        "PCA over all blocks (the default: fewer passes, more accurate); needs a resident matrix")
    return p


def _engine_modes(a, bed_samples: int = 0):
    """(precision, storage); storage "auto": a .bed of >= 1 024 samples stays in its own 2-bit form (a quarter of the HBM, and the
    packed kernels are faster there: 7.9 vs 11.1 ms at 1M x 10k); narrower matrices and VCF input are int8 (the packed rows pad
    to 1 024 samples)."""
    from . import _lib
    if a.gpca_storage == "auto":
        a.gpca_storage = "2bit" if bed_samples >= 1024 else "int8"
    return (_lib.PREC_I8_EXACT if a.gpca_precision == "i8" else _lib.PREC_F32_MFMA,
            _lib.STORE_2BIT if a.gpca_storage == "2bit" else _lib.STORE_INT8)


def _log(msg: str):
    print(f"[genomic_pca_amd] {msg}", file=sys.stderr, flush=True)


def _ensure_parent(prefix: str):
    parent = os.path.dirname(prefix)
    if parent and not os.path.exists(parent):
        os.makedirs(parent, exist_ok=True)                                         # main.rs:372-377


def run_vcf_workflow(a) -> int:
    if not a.vcf_dir or a.components is None:
        raise SystemExit("error: --vcf-dir and --components are required unless --eigensnp is given")
    t0 = time.time()
    files = sorted(os.path.join(a.vcf_dir, f) for f in os.listdir(a.vcf_dir) if f.endswith(".vcf") or f.endswith(".vcf.gz"))
    if not files:
        raise SystemExit(f"No VCF files found in {a.vcf_dir}")                       # main.rs:153-155
    maf = 0.01 if a.maf is None else a.maf
    samples, ids, chunks = None, [], []
    for f in files:
        s, i, g = gio.read_vcf(f, maf)
        if samples is None:
            samples = s
        elif s != samples:
            raise SystemExit(f"Sample mismatch between VCF files: {f}")             # vcf.rs:78-95
        ids += i
        chunks.append(g)
    G = np.concatenate(chunks, axis=0) if chunks else np.zeros((0, 0), np.int8)
    _log(f"{len(files)} VCF files, {G.shape[0]} variants x {len(samples or [])} samples in {time.time() - t0:.2f}s")
    if G.shape[0] == 0:
        raise SystemExit("No variants available to build matrix.")                  # vcf.rs:321-323
    prec, store = _engine_modes(a)
    model = PCA(device=a.device, precision=prec, storage=store)
    model.rfit(G.T, a.components, 10, a.rfit_seed, None, power_iters=a.gpca_rfit_power_iters)                            # main.rs:636-656 (x = samples x variants)
    pcs = model.transform()
    _ensure_parent(a.output_prefix)
    gio.write_principal_components(a.output_prefix, "vcf.pca.tsv", samples, pcs)    # main.rs:231
    gio.write_eigenvalues(a.output_prefix, model.explained_variance() if a.write_eigenvalues else [])   # main.rs:232, 676
    _log(f"VCF workflow done in {time.time() - t0:.2f}s")
    return 0


def _load_bed(eng, a, fs, cols):
    """The .bed payload into the engine: resident (decoded on the GPU from 256 MiB row chunks of the memory map), or -- when
    it does not fit the device, or on request -- out of core: every pass walks the memory map panel by panel through a ring
    of HBM buffers, and the HBM that is left keeps the leading panels (gpca_stream_set_cache).  The reference's solver pulls
    strips through the accessor on every pass in the same way (prepare.rs:1839-2022, main.rs:322)."""
    from . import _lib
    from .engine import PanelSource
    rows = fs.bed_rows
    lut = np.array([2, -127, 1, 0], np.int8)                                         # count_a1 (prepare.rs:622-629)
    shift = None if cols is None else (2 * (cols % 4)).astype(np.uint8)

    def subset(r0, n):                                                               # kept sample columns of rows [r0, r0 + n)
        return lut[(np.asarray(rows[r0:r0 + n])[:, cols // 4] >> shift) & 3]
    n_samples = fs.n_samples if cols is None else len(cols)
    if a.gpca_stream == "auto":
        # resident needs the matrix (1 B or 0.25 B per genotype, rows padded) plus the solver's workspace (gpca.h,
        # gpca_get_device_memory): a load that fits with nothing to spare would only fail later, in gpca_rsvd
        free, _ = eng.device_memory()
        free = int(os.environ.get("GPCA_CLI_FREE_BYTES", free))       # (test hook: pretend the device is smaller)
        per_row = -(-n_samples // 1024) * 1024 // (4 if a.gpca_storage == "2bit" else 1) + 512
        need = rows.shape[0] * (per_row + 1024) + n_samples * 8192 + (1 << 30)
        if need > free:
            _log(f"the genotype matrix needs about {need / 2**30:.1f} GiB resident, {free / 2**30:.1f} GiB are free: walking it out of core")
            a.gpca_stream = "on"
    if a.gpca_stream != "on":
        try:
            if cols is None:
                eng.upload_bed2bit(rows, fs.n_samples)
            else:
                eng.load_from_source(PanelSource.host_i8(subset), rows.shape[0], n_samples)
            return
        except _lib.GpcaError as e:
            if a.gpca_stream == "off" or e.status != _lib.GPCA_ERR_OOM:
                raise
            _log("the genotype matrix does not fit the device: walking it out of core")
    # the memory-mapped payload itself is the source (GPCA_PANEL_MAPPED_BED): the library's copy threads stage its panels, no callback
    src = PanelSource.mapped_bed(rows) if cols is None else PanelSource.host_i8(subset)
    eng.stream_open(src, rows.shape[0], n_samples, panel_rows=a.gpca_panel_rows, cache_bytes=-1)


def run_eigensnp_workflow(a) -> int:
    if not a.bed_file or not a.ld_block_file:
        raise SystemExit("error: --bed-file and --ld-block-file are required when --eigensnp is used")   # main.rs:296-301
    t0 = time.time()
    fs = gio.read_plink(a.bed_file)
    prec, store = _engine_modes(a, fs.n_samples)
    eng = GpcaEngine(device=a.device, precision=prec, storage=store)
    sample_ids = fs.sample_ids
    cols = None
    if a.eigensnp_sample_keep_file:                                                  # prepare.rs:1058-1096
        keep_ids = set(gio.read_sample_keep_file(a.eigensnp_sample_keep_file))
        cols = np.array([i for i, s in enumerate(fs.sample_ids) if s in keep_ids], np.int64)
        if len(cols) == 0:
            _log("No samples available after sample QC."); return 0
        sample_ids = [fs.sample_ids[i] for i in cols]
    _load_bed(eng, a, fs, cols)
    st = eng.snp_stats(QcConfig(a.eigensnp_min_call_rate, a.eigensnp_min_maf, a.eigensnp_max_hwe_p))
    blocks = gio.parse_ld_block_file(a.ld_block_file)
    keep, by_tag = gio.map_snps_to_ld_blocks(blocks, fs.chromosomes, fs.positions, st["keep"])
    _log(f"{int(st['keep'].sum())} / {len(st['keep'])} SNPs passed QC; {int(keep.sum())} fall in {len(by_tag)} LD blocks")
    if len(sample_ids) == 0 or int(keep.sum()) == 0:
        _log("No samples or SNPs available for EigenSNP PCA after preparation.")     # main.rs:349-352
        return 0
    eng.set_standardization(st["mu"], st["sigma"], keep)
    acc = MicroarrayGenotypeAccessor(eng)
    rows = acc.original_indices_of_pca_snps()
    row_to_id = {int(r): i for i, r in enumerate(rows)}
    specs = [LdBlockSpecification(tag, [row_to_id[r] for r in rs]) for tag, rs in by_tag]
    cfg = EigenSNPCoreAlgorithmConfig(
        target_num_global_pcs=a.eigensnp_k_global, components_per_ld_block=a.eigensnp_components_per_block,
        subset_factor_for_local_basis_learning=a.eigensnp_subset_factor,
        min_subset_size_for_local_basis_learning=a.eigensnp_min_subset_size,
        max_subset_size_for_local_basis_learning=a.eigensnp_max_subset_size,
        global_pca_sketch_oversampling=a.eigensnp_global_oversampling,
        global_pca_num_power_iterations=a.eigensnp_global_power_iter,
        local_rsvd_sketch_oversampling=a.eigensnp_local_oversampling,
        local_rsvd_num_power_iterations=a.eigensnp_local_power_iter, random_seed=a.eigensnp_seed,
        snp_processing_strip_size=a.eigensnp_snp_strip_size, refine_pass_count=a.eigensnp_refine_passes,
        collect_diagnostics=a.eigensnp_collect_diagnostics)
    k = min(cfg.target_num_global_pcs, len(sample_ids), len(rows))
    cfg.target_num_global_pcs = k
    cfg.global_pca_sketch_oversampling = max(0, min(cfg.global_pca_sketch_oversampling, min(len(sample_ids), len(rows)) - k))
    out, _ = EigenSNPCoreAlgorithm(cfg).compute_pca(acc, specs, local_stage=a.gpca_eigensnp_local_stage)
    _ensure_parent(a.output_prefix)
    gio.write_principal_components(a.output_prefix, "eigensnp.pca.tsv", sample_ids, out.final_sample_principal_component_scores)
    gio.write_eigenvalues(a.output_prefix, out.final_principal_component_eigenvalues)
    gio.write_loadings(a.output_prefix, [fs.variant_ids[r] for r in rows], [fs.chromosomes[r] for r in rows],
                       [int(fs.positions[r]) for r in rows], out.final_snp_principal_component_loadings)
    eng.close()
    _log(f"EigenSNP workflow done in {time.time() - t0:.2f}s")
    return 0


def main(argv=None) -> int:
    a = build_parser().parse_args(argv)
    return run_eigensnp_workflow(a) if a.eigensnp else run_vcf_workflow(a)


if __name__ == "__main__":
    sys.exit(main())
