"""Row shards on SEPARATE GPUs exchanging through RCCL inside libgpca.so (gpca_comm_init / ncclAllReduce on the engine's
stream) -- the path the driver's 2/4/8-GPU bench takes.  Needs two visible GPUs: skipped on the one-GPU test box (where the
same exchange is covered at world = 1 over RCCL and at world = 2 over the host hook, tests/test_gpu_stream.py)."""
import os

import numpy as np
import pytest

from conftest import gpu_count

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, M, N, k, seed, out_dir, poison_rank, streamed):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch                                   # torch first: one HIP runtime in the process ("PyTorch in the same process")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import genomic_pca_amd as g
    from genomic_pca_amd import _lib
    from genomic_pca_amd.distributed import broadcast_unique_id, shard_rows
    a, b_ = shard_rows(M, world, rank)
    th = g.synth_thresholds(b_ - a, 8, seed=seed, fst=0.3, snp_offset=a)
    with g.GpcaEngine(device=rank, precision=_lib.PREC_I8_EXACT) as e:
        if streamed:
            e.stream_open(g.PanelSource.synth(th, seed, snp_offset=a), b_ - a, N, panel_rows=1024, ring_slots=2, fused=False)
        else:
            e.synth_genotypes(b_ - a, N, seed, th, snp_offset=a)
            if rank == poison_rank:
                G = e.download_genotypes_i8(); G[11, 3] = -127; e.upload_genotypes_i8(G)
        e.snp_stats(g.QcConfig(0.5, 0.0, 1.0))
        e.comm_init(world, rank, broadcast_unique_id(g.GpcaEngine, rank), a)
        assert e.comm_count_ranks() == world      # libgpca's own RCCL communicator (beside torch's) reaches every rank
        try:
            e.rsvd(k, 10, 2, seed=seed)
            np.savez(os.path.join(out_dir, f"rank{rank}.npz"), status=0, ev=e.eigenvalues(), sc=e.scores(f64=True), ld=e.loadings())
        except g.GpcaError as err:
            np.savez(os.path.join(out_dir, f"rank{rank}.npz"), status=err.status)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.skipif(gpu_count() < 2, reason="needs two GPUs")
@pytest.mark.parametrize("poison_rank,streamed", [(-1, False), (1, False), (-1, True)])
def test_two_gpus_through_rccl(tmp_path, gpca, oracle, poison_rank, streamed):
    import torch.multiprocessing as mp
    M, N, k, seed, world = 6000, 512, 6, 23, 2
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, M, N, k, seed, str(tmp_path), poison_rank, streamed), nprocs=world, join=True)
    z = [np.load(os.path.join(tmp_path, f"rank{i}.npz")) for i in range(world)]
    if poison_rank >= 0:
        assert int(z[0]["status"]) == -5 and int(z[1]["status"]) == -5          # both ranks leave with the failing rank's code
        return
    assert int(z[0]["status"]) == 0 and int(z[1]["status"]) == 0
    assert np.array_equal(z[0]["ev"], z[1]["ev"]) and np.array_equal(z[0]["sc"], z[1]["sc"])   # replicated results, same bits
    from genomic_pca_amd import _lib
    with gpca.GpcaEngine(precision=_lib.PREC_I8_EXACT) as e:
        e.synth_genotypes(M, N, seed, gpca.synth_thresholds(M, 8, seed=seed, fst=0.3))
        e.snp_stats(gpca.QcConfig(0.5, 0.0, 1.0)); e.rsvd(k, 10, 2, seed=seed)
        assert np.max(np.abs(z[0]["ev"] - e.eigenvalues()) / e.eigenvalues()) < 5e-8
        assert oracle.max_abs_dpc(z[0]["sc"], e.scores(f64=True)) < 1e-7
        ld = np.concatenate([z[0]["ld"], z[1]["ld"]], axis=0).astype(np.float64)
        assert oracle.max_abs_dpc(ld, e.loadings().astype(np.float64)) < 1e-6
