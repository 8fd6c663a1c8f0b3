"""Row shards in SEPARATE PROCESSES, started by the product's own torch-free launcher (genomic_pca_amd/launch.py), exchanging
(a) through RCCL inside libgpca.so (gpca_comm_init / ncclAllReduce on the engine's stream) on two GPUs -- the path the driver's
2/4/8-GPU bench takes; needs two visible GPUs, skipped on the one-GPU test box -- and (b) on ONE GPU through the host-staged
hook over the launcher's hub: everything of the multi-rank run except RCCL itself (which refuses two ranks on one device), and
`bench.py --gpus 2` end to end in that form."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import gpu_count

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    import genomic_pca_amd as g
    from genomic_pca_amd import _lib, launch
    from genomic_pca_amd.distributed import broadcast_unique_id, shard_rows
    M, N, k, seed, poison_rank, streamed, exchange, out_dir = {M}, {N}, {k}, {seed}, {poison_rank}, {streamed}, {exchange!r}, {out_dir!r}
    r = launch.from_env()
    rank, world = r.rank, r.world
    dev = int(os.environ["LOCAL_RANK"])
    a, b_ = shard_rows(M, world, rank)
    th = g.synth_thresholds(b_ - a, 8, seed=seed, fst=0.3, snp_offset=a)
    with g.GpcaEngine(device=dev, precision=_lib.PREC_I8_EXACT) as e:
        if streamed:
            e.stream_open(g.PanelSource.synth(th, seed, snp_offset=a), b_ - a, N, panel_rows=1024, ring_slots=2, fused=False)
        else:
            e.synth_genotypes(b_ - a, N, seed, th, snp_offset=a)
            if rank == poison_rank:
                G = e.download_genotypes_i8(); G[11, 3] = -127; e.upload_genotypes_i8(G)
        e.snp_stats(g.QcConfig(0.5, 0.0, 1.0))
        if exchange == "rccl":
            e.comm_init(world, rank, broadcast_unique_id(g.GpcaEngine, rank, rdzv=r), a)
        else:
            e.set_allreduce_hook(r.allreduce_hook(), world, rank, a)
        assert e.comm_count_ranks() == world      # the exchange reaches every rank
        try:
            e.rsvd(k, 10, 2, seed=seed)
            np.savez(os.path.join(out_dir, f"rank{{rank}}.npz"), status=0, ev=e.eigenvalues(), sc=e.scores(f64=True), ld=e.loadings())
        except g.GpcaError as err:
            np.savez(os.path.join(out_dir, f"rank{{rank}}.npz"), status=err.status)
    r.barrier(); r.close()
    """)


def _run(tmp_path, gpca, oracle, world, exchange, poison_rank, streamed, one_device, k=6):
    from genomic_pca_amd import launch
    M, N, seed = 6000, 512, 23
    w = tmp_path / "worker.py"
    w.write_text(WORKER.format(root=ROOT, M=M, N=N, k=k, seed=seed, poison_rank=poison_rank, streamed=streamed, exchange=exchange, out_dir=str(tmp_path)))
    codes = launch.run_ranks(world, [sys.executable, str(w)], timeout_s=500, local_ranks=[0] * world if one_device else None)
    assert codes == [0] * world, codes
    z = [np.load(os.path.join(tmp_path, f"rank{i}.npz")) for i in range(world)]
    if poison_rank >= 0:
        assert all(int(z_["status"]) == -5 for z_ in z)          # every rank leaves with the failing rank's code
        return
    assert all(int(z_["status"]) == 0 for z_ in z)
    for z_ in z[1:]:
        assert np.array_equal(z[0]["ev"], z_["ev"]) and np.array_equal(z[0]["sc"], z_["sc"])   # replicated results, same bits
    from genomic_pca_amd import _lib
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT) as e:
        e.synth_genotypes(M, N, seed, gpca.synth_thresholds(M, 8, seed=seed, fst=0.3))
        e.snp_stats(gpca.QcConfig(0.5, 0.0, 1.0)); e.rsvd(k, 10, 2, seed=seed)
        assert np.max(np.abs(z[0]["ev"] - e.eigenvalues()) / e.eigenvalues()) < 5e-8
        ns = min(k, 7)                       # (8 populations: 7 structured PCs; the PCs of the noise bulk of a wide sketch are not compared vector by vector)
        assert oracle.max_abs_dpc(z[0]["sc"][:, :ns], e.scores(f64=True)[:, :ns]) < 1e-7
        ld = np.concatenate([z_["ld"] for z_ in z], axis=0).astype(np.float64)
        assert oracle.max_abs_dpc(ld[:, :ns], e.loadings().astype(np.float64)[:, :ns]) < 1e-6


@pytest.mark.timeout(600)
@pytest.mark.skipif(gpu_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("poison_rank,streamed", [(-1, False), (1, False), (-1, True)])
def test_two_gpus_through_rccl(tmp_path, gpca, oracle, poison_rank, streamed):
    _run(tmp_path, gpca, oracle, 2, "rccl", poison_rank, streamed, one_device=False)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,poison_rank,streamed", [(2, -1, False), (3, 1, False), (2, -1, True)])
def test_ranks_on_one_gpu_through_the_launcher_and_the_host_hook(tmp_path, gpca, oracle, world, poison_rank, streamed):
    _run(tmp_path, gpca, oracle, world, "host", poison_rank, streamed, one_device=True)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("streamed", [False, True])
def test_wide_sketch_across_two_ranks(tmp_path, gpca, oracle, streamed):
    """k = 70 -> l = 80 (128 padded sketch columns) on two row shards: the (L x L + 16)-double Gram exchange, the four blocks of column
    maxima and the N x 128 sketch all-reduce of the wide path, resident and streamed."""
    _run(tmp_path, gpca, oracle, 2, "host", -1, streamed, one_device=True, k=70)


@pytest.mark.timeout(900)
def test_bench_gpus_2_rehearsal_on_one_device():
    """`python bench.py --gpus 2` started plainly: the parent spawns two ranks, the ranks rendezvous without torch, time the same steps
    between barriers, rank 0 prints ONE line carrying the multi-GPU record.  Both ranks share device 0 and exchange through the host
    hook here (RCCL needs a GPU per rank: that form is what the driver's multi-GPU run executes)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--exchange", "host", "--snps", "60000",
                        "--samples", "2048", "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["snps_per_gpu"] == 60000 and d["config"]["parallelism"] == "snp-row-shards x2"
    mg = d["multi_gpu"]
    assert mg["ranks_seen_by_the_host_hook"] == 2 and len(mg["per_rank_ms_per_step"]) == 2
    assert abs(d["ms_per_step"] - mg["per_rank_ms_per_step_max"]) < 1e-9 and d["value"] == pytest.approx(2 * 60000 * 2048 / (d["ms_per_step"] * 1e-3))
    assert len(mg["same_shard_without_exchange_ms_per_step_per_rank"]) == 2
    assert d["top_eigenvalues"][0] > d["top_eigenvalues"][2] > 0
    # the line computes its own weak-scaling efficiency, against the same shard without the exchange in this very run
    solo = mg["same_shard_without_exchange_ms_per_step_per_rank"]
    assert mg["weak_scaling_efficiency"] == pytest.approx(sum(solo) / 2 / d["ms_per_step"])
    assert mg["weak_scaling_efficiency_vs_slowest_rank"] == pytest.approx(max(solo) / d["ms_per_step"])
    assert 0.0 < mg["weak_scaling_efficiency"] <= mg["weak_scaling_efficiency_vs_slowest_rank"]
    assert "NOT vs the --gpus 1 line" in d["config"]["workload"] and d["device_memory_preflight"]["needed_GiB"] < d["device_memory_preflight"]["free_GiB"]


@pytest.mark.timeout(900)
def test_bench_gpus_2_streamed_rehearsal_on_one_device():
    """configs[4]'s form of the same: `bench.py --gpus 2 --streamed` (two ranks streaming their shards out of core, the sketch summed
    through the hook) prints the same multi-GPU record, reference and efficiency included."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--one-device", "--exchange", "host", "--streamed", "--storage", "2bit",
                        "--snps", "40000", "--samples", "2048", "--panel-rows", "8192", "--cache-gb", "0", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=800, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["residency"] == "streamed/2bit" and d["config"]["passes_over_the_source_per_call"] == 4
    mg = d["multi_gpu"]
    assert mg["ranks_seen_by_the_host_hook"] == 2 and len(mg["per_rank_ms_per_step"]) == 2 and len(mg["same_shard_without_exchange_ms_per_step_per_rank"]) == 2
    assert abs(d["ms_per_step"] - mg["per_rank_ms_per_step_max"]) < 1e-9 and d["value"] == pytest.approx(2 * 40000 * 2048 / (d["ms_per_step"] * 1e-3))
    assert mg["weak_scaling_efficiency"] == pytest.approx(sum(mg["same_shard_without_exchange_ms_per_step_per_rank"]) / 2 / d["ms_per_step"])
    assert "NOT vs a --gpus 1 line" in d["config"]["workload"]
