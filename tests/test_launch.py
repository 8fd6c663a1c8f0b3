"""The torch-free rank launcher and rendezvous of the row-sharded runs (genomic_pca_amd/launch.py) on the CPU:
collectives between real processes, a dying rank, the torchrun-style start (rank 0 hosts the hub), the sharded = unsharded
contract carried by the hub's all-reduce (per-shard products from the oracle, no GPU), and `bench.py --gpus 2` started plainly
on a box without a GPU -- it must fail inside its ranks with GPCA_ERR_NO_DEVICE, not with a usage message."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import gpu_present

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = f"import sys, os, json\nsys.path.insert(0, {ROOT!r})\nimport numpy as np\nfrom genomic_pca_amd import launch\n"


def _script(tmp_path, body, name="worker.py"):
    p = tmp_path / name
    p.write_text(PRELUDE + textwrap.dedent(body))
    return str(p)


@pytest.mark.timeout(120)
def test_collectives_between_three_processes(tmp_path):
    from genomic_pca_amd import launch
    w = _script(tmp_path, f"""
        r = launch.from_env()
        assert r is not None and r.world == 3
        got = r.allgather({{"rank": r.rank, "blob": bytes([r.rank]) * 128}})
        assert [g["rank"] for g in got] == [0, 1, 2] and got[2]["blob"] == b"\\x02" * 128
        uid = r.broadcast(b"id-from-rank-0" if r.rank == 0 else None)
        assert uid == b"id-from-rank-0"
        assert r.broadcast("from-2" if r.rank == 2 else None, src=2) == "from-2"
        r.barrier()
        assert r.max(10.0 + r.rank) == 12.0
        buf = np.arange(6, dtype=np.float64).reshape(2, 3) * (r.rank + 1)
        r.allreduce_sum_inplace(buf)
        assert np.array_equal(buf, np.arange(6, dtype=np.float64).reshape(2, 3) * 6)
        big = np.full(400_000, 0.1 * (r.rank + 1))          # 3.2 MB: larger than a socket buffer
        r.allreduce_sum_inplace(big)
        ref = np.full(400_000, 0.1); ref = ref + np.full(400_000, 0.2); ref = ref + np.full(400_000, 0.1 * 3)
        assert np.array_equal(big, ref)                     # fixed rank order: the same bits on every rank
        open(os.path.join({str(tmp_path)!r}, f"ok{{r.rank}}"), "w").write(os.environ["LOCAL_RANK"])
        r.close()
        """)
    codes = launch.run_ranks(3, [sys.executable, w], timeout_s=90)
    assert codes == [0, 0, 0] and launch.exit_code(codes) == 0
    assert [open(tmp_path / f"ok{i}").read() for i in range(3)] == ["0", "1", "2"]


@pytest.mark.timeout(120)
def test_a_dying_rank_ends_the_run(tmp_path):
    from genomic_pca_amd import launch
    w = _script(tmp_path, """
        r = launch.from_env()
        r.barrier()
        if r.rank == 1:
            sys.exit(7)                       # leaves before the next collective
        try:
            r.barrier()                       # the survivors are told instead of waiting for ever
        except RuntimeError as e:
            print("rank", r.rank, "saw:", e, flush=True)
            sys.exit(3)
        sys.exit(0)
        """)
    out = open(tmp_path / "out.txt", "w")
    codes = launch.run_ranks(3, [sys.executable, w], timeout_s=60, stdout=out, stderr=subprocess.STDOUT)
    out.close()
    assert codes[1] == 7 and all(c != 0 for c in codes), codes
    assert launch.exit_code(codes) != 0
    # a rank that hangs for good is ended by the parent (by PID) once a peer has failed
    w2 = _script(tmp_path, """
        import time
        r = launch.from_env()
        if r.rank == 0:
            sys.exit(5)
        time.sleep(600)
        """, "hang.py")
    codes = launch.run_ranks(2, [sys.executable, w2], timeout_s=60)
    assert codes[0] == 5 and codes[1] in (-15, -9), codes


@pytest.mark.timeout(120)
def test_ranks_started_by_someone_else_find_each_other(tmp_path):
    """torch.distributed.run's environment (RANK / WORLD_SIZE / MASTER_PORT, no GPCA_RDZV): rank 0 hosts the hub itself."""
    w = _script(tmp_path, f"""
        r = launch.from_env()
        v = r.allgather(r.rank * 10)
        assert v == [0, 10] and r.max(float(r.rank)) == 1.0
        open(os.path.join({str(tmp_path)!r}, f"t{{r.rank}}"), "w").write("ok")
        r.barrier(); r.close()
        """)
    env = {k: v for k, v in os.environ.items() if k != "GPCA_RDZV"}
    env.update(WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29000 + os.getpid() % 2000), TORCHELASTIC_RUN_ID="t%d" % os.getpid())
    # rank 1 first: it must wait for the hub to come up
    p1 = subprocess.Popen([sys.executable, w], env=dict(env, RANK="1", LOCAL_RANK="1"))
    p0 = subprocess.Popen([sys.executable, w], env=dict(env, RANK="0", LOCAL_RANK="0"))
    assert p0.wait(timeout=90) == 0 and p1.wait(timeout=90) == 0
    assert (tmp_path / "t0").exists() and (tmp_path / "t1").exists()


@pytest.mark.timeout(300)
def test_sharded_equals_unsharded_through_the_hub(tmp_path, oracle, gpca):
    """World = 2 over the launcher: the exchange step of gpca_rsvd (all-reduce of the N x l sketch and of one l x l Gram) carried by
    Rendezvous.allreduce_sum_inplace -- the function handed to gpca_set_allreduce_hook in the GPU rehearsal -- with the per-shard
    products from the oracle.  The sharded result equals the unsharded oracle run; both ranks hold the same bits."""
    from genomic_pca_amd import launch
    M, N, P, k, seed, world = 3000, 256, 8, 6, 19, 2
    w = _script(tmp_path, f"""
        import genomic_pca_amd as g
        from genomic_pca_amd.distributed import shard_rows
        from oracle import oracle as O
        r = launch.from_env()
        M, N, P, k, seed = {M}, {N}, {P}, {k}, {seed}
        l = k + 10
        a, b_ = shard_rows(M, r.world, r.rank, align=128)
        th = g.synth_thresholds(b_ - a, P, seed=seed, fst=0.25, snp_offset=a)
        G = O.synth_genotypes(b_ - a, N, seed, th, snp_offset=a)
        st = O.snp_stats(G, N, 0.0, 0.0, 1.0)
        rr, bb = O.scale_shift(st["mu"], st["sigma"], st["keep"])
        hook = r.allreduce_hook()
        def AtT(T):
            buf = np.ascontiguousarray(O.prod_AtT(G, N, rr, bb, T).reshape(-1)); hook(buf)
            return buf.reshape(N, l)
        Q = O.cholqr2(AtT(O.omega(b_ - a, l, seed, snp_offset=a)))
        for _ in range(2):
            Q = O.cholqr2(AtT(O.prod_AQ(G, N, rr, bb, Q)))
        B = O.prod_AQ(G, N, rr, bb, Q)
        C = np.ascontiguousarray((B.T @ B).reshape(-1)); hook(C); C = C.reshape(l, l)
        w_, V = np.linalg.eigh(C); w_ = w_[::-1]; V = V[:, ::-1]
        s = np.sqrt(w_[:k])
        scores = (Q @ V[:, :k]) * s
        sgn = np.sign(scores[np.abs(scores).argmax(axis=0), np.arange(k)])
        np.savez(os.path.join({str(tmp_path)!r}, f"rank{{r.rank}}.npz"), scores=scores * sgn, ev=w_[:k] / (N - 1), load=(B @ V[:, :k]) / s * sgn, span=np.array([a, b_]))
        r.barrier(); r.close()
        """)
    codes = launch.run_ranks(world, [sys.executable, w], timeout_s=240)
    assert codes == [0, 0]
    th = gpca.synth_thresholds(M, P, seed=seed, fst=0.25)
    G = oracle.synth_genotypes(M, N, seed, th)
    st = oracle.snp_stats(G, N, 0.0, 0.0, 1.0)
    r, b = oracle.scale_shift(st["mu"], st["sigma"], st["keep"])
    R = oracle.rsvd(G, N, r, b, k, 10, 2, seed=seed)
    z = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    assert z[0]["span"][0] == 0 and z[0]["span"][1] == z[1]["span"][0] and z[1]["span"][1] == M
    assert np.array_equal(z[0]["scores"], z[1]["scores"]) and np.array_equal(z[0]["ev"], z[1]["ev"])
    assert np.max(np.abs(z[0]["ev"] - R["eigenvalues"]) / R["eigenvalues"]) < 1e-9
    assert oracle.max_abs_dpc(z[0]["scores"], R["scores"]) < 1e-8
    assert oracle.max_abs_dpc(np.concatenate([z[0]["load"], z[1]["load"]], axis=0), R["loadings"]) < 1e-8


@pytest.mark.timeout(300)
@pytest.mark.skipif(gpu_present(), reason="the GPU-less behaviour: on a GPU box the ranks would run")
@pytest.mark.parametrize("how", ["plain", "torchrun"])
def test_bench_gpus_2_starts_its_ranks_and_fails_in_them_without_a_gpu(how):
    """`python bench.py --gpus 2` invoked the way `--gpus 1` is must reach gpca_create in two ranks of its own (here:
    GPCA_ERR_NO_DEVICE from each, non-zero exit), and the same under torch.distributed.run -- where torch only starts the processes."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"]
    if how == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(29100 + os.getpid() % 800)] + cmd[1:]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert p.returncode != 0
    assert "gpca status -8" in p.stderr and "must be launched" not in p.stderr
    if how == "plain":
        assert p.stderr.count("gpca status -8") >= 2 and "rank exit codes [1, 1]" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]     # no bench line from a run that did not run
