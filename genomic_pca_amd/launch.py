"""One process per GPU without torch: a rank launcher and a tiny rendezvous for the host side of SNP-row sharded runs
(SURVEY.md 8e).  libgpca.so does the data-path exchange itself (RCCL all-reduce of the N x l sketch on the engine's stream);
what the host needs around it is small: hand rank 0's 128-byte RCCL unique id to the other ranks, a barrier on both sides of a
timed region, and the maximum of the ranks' wall times.  That is an all-gather of small Python objects, served here by a hub on
an abstract Unix socket (all ranks of a run live on one node).

Two ways in, one code path for the ranks (`from_env`):
  * `run_ranks(world, argv)`: the caller is the parent.  It never touches a GPU and never re-execs: it hosts the hub in a
    thread, starts `argv` once per rank as a child process (RANK / LOCAL_RANK / WORLD_SIZE / GPCA_RDZV in the environment),
    waits, and when one rank dies it ends the others (by the exact PIDs it started) and reports a non-zero code.
  * under `python -m torch.distributed.run` the ranks already exist (RANK / WORLD_SIZE / MASTER_PORT set, no GPCA_RDZV):
    rank 0 hosts the hub in a daemon thread under a name derived from MASTER_PORT, the others connect to it.  torch itself is
    never imported -- torchrun is only the process starter.

The reference has no multi-process mode (one process, rayon: main.rs:100-106); this is the launcher of the build's own row shards."""
from __future__ import annotations

import hashlib
import os
import socket
import stat
import struct
import subprocess
import sys
import tempfile
import threading
import time
from multiprocessing import AuthenticationError
from multiprocessing.connection import Client, Connection, answer_challenge, deliver_challenge
from typing import Any, Callable, List, Optional, Sequence

import numpy as np

_ENV = "GPCA_RDZV"
_ENV_KEY = "GPCA_RDZV_KEY"       # hex of the run's random authkey, handed to the ranks by run_ranks (never derivable from the address)
_ERR = "__gpca_rdzv_peer_lost__"


def _derived_key(address: str) -> bytes:
    """Handshake key when no parent handed one out (ranks started by torch.distributed.run).  NOT a secret: what keeps other
    users out in that mode is the socket's home -- a directory only this user can enter -- and the peer-credential check."""
    return hashlib.sha256(("gpca:" + address).encode()).digest()


def _private_dir() -> str:
    """A directory only this user can enter (0700, owned by us), for the rendezvous socket of ranks that have no parent of ours."""
    base = os.environ.get("XDG_RUNTIME_DIR") or tempfile.gettempdir()
    d = os.path.join(base, "gpca-rdzv-%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError(f"rendezvous: {d} is not a private directory of uid {os.getuid()} (mode {oct(st.st_mode & 0o777)})")
    return d


class Hub:
    """Serves all-gather rounds to `world` clients: every round it takes one message from each rank and answers all of them
    with the list.  A rank that goes away ends the service: the ranks still waiting receive an error marker.

    Messages are pickles, so who may connect matters (an abstract socket has no file permissions and is listed in /proc/net/unix):
      * every accepted connection must come from a process of THIS user (SO_PEERCRED), else it is closed and counted;
      * then the multiprocessing challenge with the run's key: 32 random bytes that run_ranks hands to its children in
        the environment; ranks started by someone else (torch.distributed.run) use a filesystem socket inside a 0700 directory
        instead, with a key derived from its path;
      * a connection that fails either test is dropped, reported on stderr and never ends the service."""

    def __init__(self, world: int, address: Optional[str] = None, authkey: Optional[bytes] = None):
        self.world = world
        self.address = address or ("\0gpca-rdzv-%d-%d" % (os.getpid(), time.monotonic_ns()))
        self.authkey = authkey if authkey is not None else (os.urandom(32) if address is None else _derived_key(self.address))
        self._sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        if not self.address.startswith("\0"):
            try:
                if stat.S_ISSOCK(os.lstat(self.address).st_mode):      # a stale socket of an earlier run of ours (the directory is private)
                    os.unlink(self.address)
            except FileNotFoundError:
                pass
        self._sock.bind(self.address)
        self._sock.listen(max(8, world))
        self._thread: Optional[threading.Thread] = None
        self.rounds = 0
        self.rejected = 0                    # connections turned away (foreign uid, failed challenge, bad announcement)
        self.lost: Optional[int] = None      # rank whose connection ended inside a round

    def _accept(self) -> Any:
        """One authenticated connection of this user, or None for one that was turned away."""
        s, _ = self._sock.accept()
        try:
            _pid, uid, _gid = struct.unpack("3i", s.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i")))
        except OSError:
            uid = -1
        if uid != os.getuid():
            s.close()
            self.rejected += 1
            print(f"gpca rendezvous: turned away a connection from uid {uid} (hub of uid {os.getuid()})", file=sys.stderr, flush=True)
            return None
        c = Connection(s.detach())
        try:
            deliver_challenge(c, self.authkey)
            answer_challenge(c, self.authkey)
        except (AuthenticationError, EOFError, OSError) as e:
            c.close()
            self.rejected += 1
            print(f"gpca rendezvous: turned away a connection that failed the key challenge ({type(e).__name__})", file=sys.stderr, flush=True)
            return None
        return c

    def start(self) -> "Hub":
        self._thread = threading.Thread(target=self.serve, name="gpca-rdzv-hub", daemon=True)
        self._thread.start()
        return self

    def serve(self) -> None:
        conns: List[Any] = [None] * self.world
        try:
            while any(c is None for c in conns):
                c = self._accept()
                if c is None:
                    continue
                try:
                    r = c.recv()
                except (EOFError, OSError):
                    r = None
                if not isinstance(r, int) or isinstance(r, bool) or not (0 <= r < self.world) or conns[r] is not None:
                    c.close()
                    self.rejected += 1
                    print(f"gpca rendezvous: turned away a connection with the rank announcement {r!r}", file=sys.stderr, flush=True)
                    continue
                conns[r] = c
            while True:
                msgs = []
                for r, c in enumerate(conns):
                    try:
                        msgs.append(c.recv())
                    except (EOFError, OSError):
                        if msgs or r > 0:          # somebody is inside this round: tell the others why it will never finish
                            self.lost = r
                        else:                      # rank 0 closed between rounds: the normal end (everybody closes after the last round)
                            self.lost = None
                        raise EOFError
                for c in conns:
                    c.send(msgs)
                self.rounds += 1
        except EOFError:
            pass
        except OSError:                            # the listening socket was closed under us: shutdown
            pass
        finally:
            for c in conns:
                if c is not None:
                    try:
                        if self.lost is not None:
                            c.send(_ERR)
                    except Exception:              # noqa: BLE001
                        pass
                    try:
                        c.close()
                    except Exception:              # noqa: BLE001
                        pass
            self.close()

    def close(self) -> None:
        try:
            self._sock.close()
        except OSError:
            pass
        if not self.address.startswith("\0"):
            try:
                os.unlink(self.address)
            except OSError:
                pass


class Rendezvous:
    """A rank's end of the hub.  Every method is a collective: all `world` ranks must call it, in the same order."""

    def __init__(self, address: str, rank: int, world: int, connect_timeout_s: float = 180.0, authkey: Optional[bytes] = None):
        self.rank, self.world, self.address = rank, world, address
        key = authkey if authkey is not None else _derived_key(address)
        t_end = time.monotonic() + connect_timeout_s
        last: Optional[Exception] = None
        while True:
            try:
                self._c = Client(address, family="AF_UNIX", authkey=key)
                break
            except (FileNotFoundError, ConnectionRefusedError, OSError) as e:   # the hub (rank 0 under torchrun) is not up yet
                last = e
                if time.monotonic() > t_end:
                    raise RuntimeError(f"rendezvous: rank {rank} could not reach the hub within {connect_timeout_s:.0f} s: {last}") from e
                time.sleep(0.05)
        self._c.send(rank)

    def allgather(self, obj: Any) -> List[Any]:
        try:
            self._c.send(obj)
            out = self._c.recv()
        except (EOFError, OSError) as e:
            raise RuntimeError(f"rendezvous: rank {self.rank} lost the hub (another rank died?)") from e
        if isinstance(out, str) and out == _ERR:
            raise RuntimeError(f"rendezvous: a peer of rank {self.rank} left inside a collective")
        return out

    def barrier(self) -> None:
        self.allgather(None)

    def broadcast(self, obj: Any, src: int = 0) -> Any:
        return self.allgather(obj if self.rank == src else None)[src]

    def max(self, x: float) -> float:
        return max(self.allgather(float(x)))

    def allreduce_sum_inplace(self, buf: np.ndarray) -> None:
        """Host all-reduce (sum, f64) in a fixed rank order: every rank computes the same bits.  This is the transport of
        `GpcaEngine.set_allreduce_hook` for rehearsals without RCCL (two ranks on one GPU, CPU tests)."""
        parts = self.allgather(np.ascontiguousarray(buf))
        acc = np.array(parts[0], dtype=np.float64, copy=True)
        for p in parts[1:]:
            acc += p
        buf[...] = acc.reshape(buf.shape)

    def allreduce_hook(self) -> Callable[[np.ndarray], None]:
        return self.allreduce_sum_inplace

    def close(self) -> None:
        try:
            self._c.close()
        except Exception:                          # noqa: BLE001
            pass


_env_hub: Optional[Hub] = None


def from_env(environ=os.environ) -> Optional[Rendezvous]:
    """The rank's rendezvous, or None for a single-process run.  GPCA_RDZV (set by run_ranks) names the parent's hub; without it,
    WORLD_SIZE > 1 means another starter (torch.distributed.run) made the ranks: rank 0 hosts the hub itself."""
    global _env_hub
    world = int(environ.get("WORLD_SIZE", "1"))
    rank = int(environ.get("RANK", "0"))
    addr = environ.get(_ENV)
    if addr:
        key = environ.get(_ENV_KEY)
        return Rendezvous(addr.replace("@", "\0", 1) if addr.startswith("@") else addr, rank, world, authkey=bytes.fromhex(key) if key else None)
    if world <= 1:
        return None
    # no parent of ours: a filesystem socket in a directory only this user can enter (an abstract name built from MASTER_PORT and
    # the run id could be guessed -- and connected to -- by any local user)
    name = "".join(ch if ch.isalnum() or ch in "-_." else "_" for ch in
                   "%s-%s" % (environ.get("MASTER_PORT", "0"), environ.get("TORCHELASTIC_RUN_ID", "none")))[:64]
    addr = os.path.join(_private_dir(), name + ".sock")
    if rank == 0 and _env_hub is None:
        _env_hub = Hub(world, addr).start()
    return Rendezvous(addr, rank, world)


def run_ranks(world: int, argv: Sequence[str], env_extra: Optional[dict] = None, timeout_s: Optional[float] = None,
              local_ranks: Optional[Sequence[int]] = None, stdout=None, stderr=None) -> List[int]:
    """Start `argv` once per rank and wait.  Returns the ranks' exit codes (a rank that had to be ended because a peer died or
    the timeout passed reports -15 / -9).  The caller's process makes no GPU call here."""
    hub = Hub(world).start()
    procs: List[subprocess.Popen] = []
    try:
        for r in range(world):
            env = dict(os.environ)
            env.update({"RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_RANK": str(local_ranks[r] if local_ranks else r),
                        "LOCAL_WORLD_SIZE": str(world), _ENV: "@" + hub.address[1:],   # (no NUL bytes in an environment)
                        _ENV_KEY: hub.authkey.hex()})
            # HSA_ENABLE_IPC_MODE_LEGACY=0: RCCL's intra-node transport shares device buffers between the ranks' processes through HIP IPC
            # handles, and this pool's host driver only supports the dmabuf flavour of them -- with the legacy mode (the runtime's default)
            # ncclCommInitRank fails in hipIpcGetMemHandle ("invalid argument").  The image exports the variable already; it is repeated
            # here only so that a caller with a scrubbed environment gets the same transport, and never overrides a value the user set.
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if env_extra:
                env.update({k: str(v) for k, v in env_extra.items()})
            procs.append(subprocess.Popen(list(argv), env=env, stdout=stdout, stderr=stderr))
        t_end = None if timeout_s is None else time.monotonic() + timeout_s
        failed = False
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            if any(c not in (None, 0) for c in codes) or (t_end is not None and time.monotonic() > t_end):
                failed = True
                break
            time.sleep(0.05)
        if failed:                                  # end exactly the children started above, nobody else
            time.sleep(0.5)                         # (a peer that saw the hub's error marker leaves by itself with its own message)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.monotonic() + 10.0
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_kill - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        return [p.returncode for p in procs]
    finally:
        hub.close()


def exit_code(codes: Sequence[int]) -> int:
    """One process exit status for a set of rank codes: 0 only if every rank returned 0."""
    bad = [c for c in codes if c != 0]
    if not bad:
        return 0
    pos = [c for c in bad if c > 0]
    return pos[0] if pos else 1
