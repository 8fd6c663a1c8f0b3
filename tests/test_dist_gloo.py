"""CPU-only, world_size = 2 over gloo: the N>1 exchange step of SURVEY.md 8(e).

No GPU here, so the per-shard products come from the oracle; what is under test is the product's sharding
contract: `shard_rows` partition, Omega/genotypes drawn by GLOBAL SNP index (snp_offset), the all-reduce hook
(`genomic_pca_amd.distributed.torch_allreduce_hook`, the function libgpca.so calls through
gpca_set_allreduce_hook) summing the N x l sketch and the l x l Gram, and replicated orthonormalisation.
The sharded result must equal the unsharded oracle run.  (On a GPU the same exchange is exercised by
test_allreduce_hook_two_shards_one_gpu and test_rccl_world1.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, M, N, P, k, seed, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import genomic_pca_amd as g
    from genomic_pca_amd.distributed import shard_rows, torch_allreduce_hook
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    allreduce = torch_allreduce_hook()
    a, b_ = shard_rows(M, world, rank, align=128)
    l = k + 10
    th = g.synth_thresholds(b_ - a, P, seed=seed, fst=0.25, snp_offset=a)
    G = O.synth_genotypes(b_ - a, N, seed, th, snp_offset=a)
    st = O.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = O.scale_shift(st["mu"], st["sigma"], st["keep"])

    def AtT(T):
        Y = O.prod_AtT(G, N, r, b, T)          # this shard's partial sketch (includes its own 1 c^T term)
        buf = np.ascontiguousarray(Y.reshape(-1)); allreduce(buf)
        return buf.reshape(N, l)
    Q = O.cholqr2(AtT(O.omega(b_ - a, l, seed, snp_offset=a)))
    for _ in range(2):
        Q = O.cholqr2(AtT(O.prod_AQ(G, N, r, b, Q)))
    B = O.prod_AQ(G, N, r, b, Q)
    C = np.ascontiguousarray((B.T @ B).reshape(-1)); allreduce(C); C = C.reshape(l, l)
    w, V = np.linalg.eigh(C); w = w[::-1]; V = V[:, ::-1]
    s = np.sqrt(w[:k])
    scores = (Q @ V[:, :k]) * s
    sgn = np.sign(scores[np.abs(scores).argmax(axis=0), np.arange(k)])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), scores=scores * sgn, ev=w[:k] / (N - 1), load=(B @ V[:, :k]) / s * sgn,
             span=np.array([a, b_]))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharded_equals_unsharded(tmp_path, oracle, gpca):
    import torch.multiprocessing as mp
    M, N, P, k, seed, world = 3000, 256, 8, 6, 17, 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, M, N, P, k, seed, str(tmp_path)), nprocs=world, join=True)
    th = gpca.synth_thresholds(M, P, seed=seed, fst=0.25)
    G = oracle.synth_genotypes(M, N, seed, th)
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r, b, k, 10, 2, seed=seed)
    z = [np.load(os.path.join(tmp_path, f"rank{i}.npz")) for i in range(world)]
    assert z[0]["span"][0] == 0 and z[0]["span"][1] == z[1]["span"][0] and z[1]["span"][1] == M
    assert np.array_equal(z[0]["scores"], z[1]["scores"]) and np.array_equal(z[0]["ev"], z[1]["ev"])   # replicated state identical
    assert np.max(np.abs(z[0]["ev"] - R["eigenvalues"]) / R["eigenvalues"]) < 1e-9
    assert oracle.max_abs_dpc(z[0]["scores"], R["scores"]) < 1e-8
    load = np.concatenate([z[0]["load"], z[1]["load"]], axis=0)
    assert oracle.max_abs_dpc(load, R["loadings"]) < 1e-8
