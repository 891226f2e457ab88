"""Sharding of query batches across the GPUs of one node -- one process per GPU, no PyTorch.

The path has no exchange step (the reference itself is single-process,
``docs/roadmap.md:245``): every query point is independent and the model (<= 1.3 MB) is
replicated, so rank g evaluates a contiguous row block of ``points`` on its own GPU.  What
is left is getting the result blocks into one place.  Two ways, both here:

* :class:`RcclComm` -- the gather on RCCL over xGMI (``pcx_comm_*`` of the C ABI: grouped
  ``ncclSend``/``ncclRecv``, one direct link per peer), result on rank 0's GPU;
* :class:`SharedResult` -- every rank copies its block device-to-host straight into its
  slice of one POSIX shared-memory result array (G parallel PCIe copies, no collective).

:class:`HostGroup` is the host-side rendezvous both need (ranks, barrier, small blobs such
as the RCCL unique id, max-over-ranks).  It is a file in ``/dev/shm`` with one
single-writer slot per rank -- it works without a GPU, which is how the world-size-2 CPU
tests drive this module.  Environment: ``RANK`` / ``LOCAL_RANK`` / ``WORLD_SIZE`` as set by
``torch.distributed.run`` (used as a plain process launcher) or by ``bench.py``'s own
launcher; ``PCX_RDZV_DIR`` names the rendezvous directory explicitly.
"""
from __future__ import annotations

import ctypes
import mmap
import os
import struct
import time
from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(n_rows: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block ``[lo, hi)`` of rank ``rank``: blocks of ``ceil(N / G)`` rows."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    per = -(-n_rows // world_size)
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)


def shard_table(n_rows: int, world_size: int, width: int = 1):
    """(counts, offsets) in elements of every rank's block, ``width`` values per row."""
    b = [shard_bounds(n_rows, r, world_size) for r in range(world_size)]
    counts = np.array([(hi - lo) * width for lo, hi in b], dtype=np.int64)
    offsets = np.array([lo * width for lo, _ in b], dtype=np.int64)
    return counts, offsets


# --------------------------------------------------------------------------------------
# host rendezvous
# --------------------------------------------------------------------------------------
_MAGIC = 0x5043584752503031          # "PCXGRP01"
_HDR = 64                            # magic, world, created_ns, launch nonce
_SLOT = 512                          # gen (i64), blob length (i64), blob (<= 496 bytes)
_BLOB_MAX = _SLOT - 16


class _StaleRendezvous(TimeoutError):
    """The mapped rendezvous file was replaced before this rank's first barrier passed."""


def _launch_nonce() -> int:
    """The same number on every rank of one launch, different between launches that share a directory."""
    import zlib
    key = "%s|%d|%s|%s" % (os.environ.get("PCX_RDZV_NONCE", ""), os.getppid(), os.environ.get("MASTER_PORT", ""),
                           os.environ.get("TORCHELASTIC_RESTART_COUNT", ""))
    return zlib.crc32(key.encode()) + 1


class HostGroup:
    """Ranks of one node meeting in a shared-memory file: one slot per rank, written only
    by its owner (generation counter + a small blob), read by everyone.  ``barrier``,
    ``allgather_bytes``, ``broadcast_bytes``, ``max``.  Spin-waits with short sleeps; every
    wait has a timeout (a missing rank raises instead of hanging)."""

    def __init__(self, rank: int, world: int, directory: str, timeout: float = 300.0):
        if world < 1 or not (0 <= rank < world):
            raise ValueError(f"bad rank/world {rank}/{world}")
        self.rank, self.world, self.timeout = rank, world, timeout
        self.directory = directory
        self.path = os.path.join(directory, "group.bin")
        self._gen = 0
        self._mm = None
        self._size = _HDR + world * _SLOT
        # every rank of ONE launch computes the same nonce (same launcher process, same master port / restart count,
        # or an explicit PCX_RDZV_NONCE): a file left in the same directory by another launch does not match it
        self._nonce = _launch_nonce()
        if rank == 0:
            os.makedirs(directory, exist_ok=True)
            try:
                os.unlink(self.path)                     # leftovers of a crashed run with the same key
            except FileNotFoundError:
                pass
            tmp = self.path + f".{os.getpid()}.tmp"
            with open(tmp, "wb") as f:
                f.write(struct.pack("<qqqq", _MAGIC, world, time.time_ns(), self._nonce).ljust(_HDR, b"\0"))
                f.write(b"\0" * (world * _SLOT))
            os.rename(tmp, self.path)                    # appears complete or not at all
        deadline = time.monotonic() + timeout
        # Attach, then pass the first barrier.  A rank that started before rank 0 may have mapped the (still fresh)
        # file of a crashed launch whose slot for this rank was never written: rank 0 of THIS launch then replaces the
        # file under it.  That is noticed inside the first barrier (the path names another inode): the mapping is
        # dropped and the rank goes back to polling for rank 0's new file, until the deadline.
        while True:
            self._attach(deadline)
            self._attached = False       # until the handshake passes, keep checking that the path still names this file
            try:
                self._hello()
            except _StaleRendezvous:
                self._drop_mapping()
                self._gen = 0
                continue
            self._attached = True
            break

    def _attach(self, deadline: float):
        rank, world, size = self.rank, self.world, self._size
        self._ino = None
        while True:
            try:
                fd = os.open(self.path, os.O_RDWR)
                try:
                    st = os.fstat(fd)
                    if st.st_size == size:
                        head = os.pread(fd, 32, 0)
                        magic, w, created, nonce = struct.unpack("<qqqq", head)
                        fresh = abs(time.time_ns() - created) < 900e9
                        # A file left by a crashed run with the same key is recognised by its nonce, or by this rank's
                        # own slot: only its owner ever writes a slot, so in the file of THIS launch its generation is
                        # still 0 (rank 0 unlinks and recreates the file; wait for the new one).
                        (own_gen,) = struct.unpack("<q", os.pread(fd, 8, _HDR + rank * _SLOT))
                        if (magic == _MAGIC and w == world and fresh and nonce == self._nonce and own_gen == 0
                                and self._same_file(st)):
                            self._mm = mmap.mmap(fd, size)
                            self._ino = (st.st_dev, st.st_ino)
                            break
                finally:
                    os.close(fd)
            except FileNotFoundError:
                pass
            if time.monotonic() > deadline:
                raise TimeoutError(f"rank {rank}: no rendezvous file {self.path} from rank 0")
            time.sleep(0.002)
        self._gens = np.frombuffer(self._mm, dtype=np.int64, count=world * (_SLOT // 8),
                                   offset=_HDR)[:: _SLOT // 8]

    def _hello(self):
        """Handshake behind every attach: each rank puts a random token into its slot, rank 0 answers with a digest of
        all of them, every rank checks the digest.  Generation counters alone cannot tell a live file from the one a
        crashed launch left behind -- its dead ranks sit at high generations and every barrier against them passes at
        once; a dead rank 0 cannot have signed THIS rank's token."""
        import hashlib
        token = os.urandom(8)
        off = self._slot(self.rank)
        self._mm[off + 8: off + 16] = struct.pack("<q", 8)
        self._mm[off + 16: off + 24] = token
        self.barrier()                                   # every token is in place

        def tokens():
            return b"".join(bytes(self._mm[self._slot(r) + 16: self._slot(r) + 24]) for r in range(self.world))

        if self.rank == 0:
            self._mm[off + 24: off + 56] = hashlib.sha256(tokens()).digest()
            self._mm[off + 8: off + 16] = struct.pack("<q", 40)
        self.barrier()                                   # rank 0 has signed
        o0 = self._slot(0)
        signed = bytes(self._mm[o0 + 24: o0 + 56])
        if signed != hashlib.sha256(tokens()).digest() or bytes(self._mm[off + 16: off + 24]) != token:
            raise _StaleRendezvous(f"rank {self.rank}: {self.path} was not signed by rank 0 of this launch")
        self.barrier()                                   # everyone has checked: the slots may be reused

    def _drop_mapping(self):
        self._gens = None
        if self._mm is not None:
            try:
                self._mm.close()
            except BufferError:
                pass
        self._mm = None

    def _same_file(self, st) -> bool:
        """The path still names the file behind `st` (rank 0 replaces a stale file by unlink + rename)."""
        try:
            now = os.stat(self.path)
        except FileNotFoundError:
            return False
        return (now.st_dev, now.st_ino) == (st.st_dev, st.st_ino)

    def subgroup(self, name: str, timeout: Optional[float] = None) -> "HostGroup":
        """A second, independent group of the same ranks (own file, own generation counters) -- for a helper
        thread whose collectives must not interleave with the main thread's.  Collective: every rank calls it."""
        return HostGroup(self.rank, self.world, os.path.join(self.directory, name),
                         self.timeout if timeout is None else timeout)

    # -- construction from the launcher's environment ---------------------------------
    @classmethod
    def from_env(cls, timeout: float = 300.0) -> "HostGroup":
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        d = os.environ.get("PCX_RDZV_DIR")
        if not d:
            # ranks of one launch share their parent (the launcher) and its master port
            key = "%d_%s_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"),
                                os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
            base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
            d = os.path.join(base, f"pcx_rdzv_{os.getuid()}_{key}")
        return cls(rank, world, d, timeout)

    # -- primitives -------------------------------------------------------------------
    def _slot(self, r: int) -> int:
        return _HDR + r * _SLOT

    def _wait_all(self, gen: int):
        deadline = time.monotonic() + self.timeout
        spins = 0
        while True:
            if int(self._gens.min()) >= gen:
                return
            spins += 1
            if spins > 200:
                time.sleep(0.0002)
                if not self._attached and spins % 50 == 0:
                    now = None
                    try:
                        now = os.stat(self.path)
                    except FileNotFoundError:
                        pass
                    if now is None or (now.st_dev, now.st_ino) != self._ino:
                        # rank 0 of this launch replaced the file this rank had mapped: attach again (__init__)
                        raise _StaleRendezvous(f"rank {self.rank}: attached to a stale rendezvous file {self.path}")
                if time.monotonic() > deadline:
                    late = [r for r in range(self.world) if int(self._gens[r]) < gen]
                    raise TimeoutError(f"rank {self.rank}: ranks {late} did not reach barrier {gen}")

    def barrier(self):
        self._gen += 1
        self._gens[self.rank] = self._gen
        self._wait_all(self._gen)

    def allgather_bytes(self, blob: bytes):
        if len(blob) > _BLOB_MAX:
            raise ValueError(f"blob of {len(blob)} bytes exceeds {_BLOB_MAX}")
        off = self._slot(self.rank)
        self._mm[off + 8: off + 16] = struct.pack("<q", len(blob))
        self._mm[off + 16: off + 16 + len(blob)] = blob
        self.barrier()                                   # everyone has written
        out = []
        for r in range(self.world):
            o = self._slot(r)
            (n,) = struct.unpack("<q", self._mm[o + 8: o + 16])
            out.append(bytes(self._mm[o + 16: o + 16 + n]))
        self.barrier()                                   # everyone has read: slots may be reused
        return out

    def broadcast_bytes(self, blob: Optional[bytes], src: int = 0) -> bytes:
        return self.allgather_bytes(blob if self.rank == src else b"")[src]

    def gather_floats(self, value: float):
        vals = self.allgather_bytes(struct.pack("<d", float(value)))
        bad = [r for r, v in enumerate(vals) if len(v) != 8]
        if bad:       # another rank was in a different collective: say so instead of failing in struct.unpack
            raise RuntimeError(f"rank {self.rank}: ranks {bad} answered a float gather with {[len(vals[r]) for r in bad]} "
                               "bytes -- the ranks are not running the same sequence of collectives")
        return [struct.unpack("<d", v)[0] for v in vals]

    def max(self, value: float) -> float:
        return max(self.gather_floats(value))

    def close(self):
        if getattr(self, "_mm", None) is None:
            return
        try:
            self.barrier()
        except TimeoutError:
            pass
        self._gens = None
        try:
            self._mm.close()
        except BufferError:
            pass
        self._mm = None
        if self.rank == 0:
            for p in (self.path,):
                try:
                    os.unlink(p)
                except OSError:
                    pass
            try:
                os.rmdir(self.directory)
            except OSError:
                pass


# --------------------------------------------------------------------------------------
# result collection
# --------------------------------------------------------------------------------------
class SharedResult:
    """One host result array in POSIX shared memory that every rank writes its own block
    into ("G parallel D2H copies straight into the host result").  With a
    GPU the mapping is pinned (``pcx_host_register``) so the copies are asynchronous and
    run at PCIe rate; ``pinned`` says whether that succeeded."""

    def __init__(self, group: HostGroup, n_values: int, device: Optional[int] = None, name: str = "result"):
        self.group, self.n = group, int(n_values)
        self.path = os.path.join(group.directory, f"{name}.f64")
        nbytes = max(8, self.n * 8)
        if group.rank == 0:
            with open(self.path, "wb") as f:
                f.truncate(nbytes)
        group.barrier()
        fd = os.open(self.path, os.O_RDWR)
        try:
            self._mm = mmap.mmap(fd, nbytes)
        finally:
            os.close(fd)
        self.array = np.frombuffer(self._mm, dtype=np.float64, count=self.n)
        self.pinned = False
        self._lib = None
        if device is not None:
            from . import _lib
            self._lib = _lib.load()
            self._addr = ctypes.addressof(ctypes.c_char.from_buffer(self._mm))
            self.pinned = self._lib.pcx_host_register(device, ctypes.c_void_p(self._addr), nbytes) == 0
        group.barrier()

    def address(self, offset_values: int = 0) -> int:
        return ctypes.addressof(ctypes.c_char.from_buffer(self._mm)) + 8 * int(offset_values)

    def close(self):
        if self._mm is None:
            return
        if self.pinned:
            self._lib.pcx_host_unregister(ctypes.c_void_p(self._addr))
            self.pinned = False
        self.group.barrier()
        self.array = None
        try:
            self._mm.close()
        except BufferError:
            pass
        self._mm = None
        if self.group.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass


class RcclComm:
    """``pcx_comm`` of the C ABI: an RCCL communicator over the ranks of a :class:`HostGroup`
    (rank 0 draws the unique id, the group hands it round)."""

    def __init__(self, group: HostGroup, device: int):
        from . import _lib
        self._libmod = _lib
        self.lib = _lib.load()
        self.group, self.device = group, device
        # The broadcast is unconditional: if rank 0 cannot draw the id (librccl missing, ...) it still takes part,
        # sending a status byte + message, and EVERY rank raises after the collective -- a rank that raised before
        # it would leave the others pairing this broadcast with whatever collective rank 0 runs next.
        msg = b""
        if group.rank == 0:
            uid = ctypes.create_string_buffer(128)
            rc = self.lib.pcx_comm_unique_id(uid)
            if rc == 0:
                msg = b"\x01" + uid.raw
            else:
                err = self.lib.pcx_last_error() or b"pcx_comm_unique_id failed"
                msg = b"\x00" + bytes(err)[:300]
        blob = group.broadcast_bytes(msg if group.rank == 0 else None, 0)
        if len(blob) != 129 or blob[:1] != b"\x01":
            raise RuntimeError("RCCL unique id unavailable on rank 0: "
                               + (blob[1:].decode("utf-8", "replace") if len(blob) > 1 else "empty bootstrap message"))
        blob = blob[1:]
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.pcx_comm_create(device, group.rank, group.world, blob, ctypes.byref(self.handle)),
                   self.lib)
        ver = ctypes.c_int32(0)
        _lib.check(self.lib.pcx_comm_info(self.handle, None, None, None, ctypes.byref(ver)), self.lib)
        self.rccl_version = int(ver.value)

    def stream(self) -> ctypes.c_void_p:
        st = ctypes.c_void_p()
        self._libmod.check(self.lib.pcx_comm_stream(self.handle, ctypes.byref(st)), self.lib)
        return st

    def gatherv_dev(self, d_send, d_recv, counts: np.ndarray, offsets: np.ndarray, root: int = 0, stream=None):
        """Enqueue the gather (device pointers as ints / c_void_p); does not synchronize."""
        c = np.ascontiguousarray(counts, dtype=np.int64)
        o = np.ascontiguousarray(offsets, dtype=np.int64)
        self._libmod.check(self.lib.pcx_comm_gatherv_dev(self.handle, d_send, d_recv, self._libmod.p_i64(c),
                                                         self._libmod.p_i64(o), root, stream), self.lib)

    def max(self, value: float) -> float:
        v = ctypes.c_double(float(value))
        self._libmod.check(self.lib.pcx_comm_allreduce_max(self.handle, ctypes.byref(v)), self.lib)
        return float(v.value)

    def barrier(self):
        self._libmod.check(self.lib.pcx_comm_barrier(self.handle), self.lib)

    def close(self):
        if self.handle:
            self.lib.pcx_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()


def eval_sharded(evaluate: Callable[[np.ndarray], np.ndarray], points: np.ndarray, group: HostGroup,
                 width: int = 1, dst: int = 0):
    """Evaluate ``points`` (same array on every rank) block-wise: each rank runs ``evaluate``
    (a model's ``vectorized_eval_batch`` / ``eval_batch`` bound to this rank's GPU) on its own
    row block and stores the values in its slice of a shared host array -- no collective.
    Returns the full ``(N,)`` (or ``(N, width)``) array on rank ``dst`` and ``None`` elsewhere."""
    n = points.shape[0]
    lo, hi = shard_bounds(n, group.rank, group.world)
    res = SharedResult(group, n * width)
    try:
        if hi > lo:
            local = np.asarray(evaluate(points[lo:hi]), dtype=np.float64).reshape(-1)
            if local.size != (hi - lo) * width:
                raise ValueError(f"evaluate returned {local.size} values for {hi - lo} rows x {width}")
            res.array[lo * width: hi * width] = local
        group.barrier()
        full = None
        if group.rank == dst:
            full = np.array(res.array, copy=True)
            if width > 1:
                full = full.reshape(n, width)
    finally:
        res.close()
    return full
