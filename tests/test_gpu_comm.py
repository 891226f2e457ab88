"""The multi-GPU plumbing on the one GPU a test box has: the RCCL communicator of the C ABI
with a single rank (init, gather, max, barrier all go through librccl), the pinned shared
host result, and bench.py's own launcher with two ranks sharing GPU 0 (RCCL refuses two
ranks on one device, so that rehearsal collects the blocks through shared host memory)."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, GOLDEN

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_gather_through_the_c_abi(tmp_path):
    from pychebyshev_amd import _lib
    from pychebyshev_amd.distributed import HostGroup, RcclComm, shard_table
    lib = _lib.load()
    group = HostGroup(0, 1, str(tmp_path / "rdzv"))
    comm = RcclComm(group, 0)
    assert comm.rccl_version > 20000
    rank, world, dev = ctypes.c_int32(-1), ctypes.c_int32(-1), ctypes.c_int32(-1)
    _lib.check(lib.pcx_comm_info(comm.handle, ctypes.byref(rank), ctypes.byref(world), ctypes.byref(dev), None), lib)
    assert (rank.value, world.value, dev.value) == (0, 1, 0)
    n = 100_003
    x = np.random.default_rng(0).standard_normal(n)
    d_send, d_recv = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.pcx_dev_malloc(0, n * 8, ctypes.byref(d_send)), lib)
    _lib.check(lib.pcx_dev_malloc(0, (n + 5) * 8, ctypes.byref(d_recv)), lib)
    _lib.check(lib.pcx_memcpy_h2d(0, d_send, x.ctypes.data_as(ctypes.c_void_p), n * 8), lib)
    counts, offsets = shard_table(n, 1)
    offsets = offsets + 5                                   # the block may land anywhere in the result
    comm.gatherv_dev(d_send, d_recv, counts, offsets, 0, None)
    comm.barrier()
    got = np.empty(n + 5)
    _lib.check(lib.pcx_memcpy_d2h(0, got.ctypes.data_as(ctypes.c_void_p), d_recv, (n + 5) * 8), lib)
    assert np.array_equal(got[5:], x)
    assert comm.max(3.25) == 3.25
    # argument errors come back as codes, not crashes
    bad = np.array([-1], dtype=np.int64)
    with pytest.raises(ValueError):
        comm.gatherv_dev(d_send, d_recv, bad, offsets, 0, None)
    with pytest.raises(ValueError):
        comm.gatherv_dev(d_send, d_recv, counts, offsets, 3, None)
    lib.pcx_dev_free(0, d_send)
    lib.pcx_dev_free(0, d_recv)
    comm.close()
    group.close()


def test_pinned_shared_result_and_async_download(tmp_path):
    from pychebyshev_amd import _lib
    from pychebyshev_amd.distributed import HostGroup, SharedResult
    lib = _lib.load()
    group = HostGroup(0, 1, str(tmp_path / "rdzv"))
    n = 1 << 20
    res = SharedResult(group, n, device=0)
    assert res.pinned, "hipHostRegister of the shared-memory mapping failed"
    x = np.random.default_rng(1).standard_normal(n)
    d = ctypes.c_void_p()
    _lib.check(lib.pcx_dev_malloc(0, n * 8, ctypes.byref(d)), lib)
    _lib.check(lib.pcx_memcpy_h2d(0, d, x.ctypes.data_as(ctypes.c_void_p), n * 8), lib)
    st = ctypes.c_void_p()
    _lib.check(lib.pcx_stream_create(0, ctypes.byref(st)), lib)
    ev = ctypes.c_void_p()
    _lib.check(lib.pcx_event_create(0, ctypes.byref(ev)), lib)
    _lib.check(lib.pcx_event_record(ev, None), lib)
    _lib.check(lib.pcx_stream_wait_event(st, ev), lib)
    half = n // 2
    _lib.check(lib.pcx_memcpy_d2h_async(ctypes.c_void_p(res.address(half)), ctypes.c_void_p(d.value + half * 8),
                                        half * 8, st), lib)
    _lib.check(lib.pcx_memcpy_d2h_async(ctypes.c_void_p(res.address(0)), d, half * 8, st), lib)
    _lib.check(lib.pcx_stream_synchronize(st), lib)
    assert np.array_equal(res.array, x)
    lib.pcx_event_destroy(ev)
    lib.pcx_stream_destroy(st)
    lib.pcx_dev_free(0, d)
    res.close()
    group.close()


def _bench(args, env_extra, timeout=600):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                         text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_bench_single_rank_with_rccl_forced():
    """--gpus 1 with the communicator forced on: every step ends with the RCCL gather."""
    line = _bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--points", "200000", "--no-companion",
                   "--no-cpu-baseline"], {"PCX_BENCH_FORCE_COMM": "1"})
    assert line["n_gpus"] == 1 and line["config"]["gather"] == "rccl"
    assert line["comm"]["backend"] == "rccl" and line["comm"]["torch"] is False
    for mode in ("none", "rccl", "rccl+d2h", "d2h", "h2d+d2h"):
        assert line["gather"][mode]["value"] > 0
    assert line["gather"]["d2h"]["host_buffer_pinned"] is True
    assert line["gather"]["h2d+d2h"]["points_page_locked"] is True
    assert 0 < line["roofline"]["frac"] < 1
    # round 4: what every rank ran on, and every rank's block of the collected result verified
    comm = line["comm"]
    assert comm["world"] == 1 and comm["distinct_gpus"] == 1 and comm["rccl_world_seen_by_every_rank"] is True
    rec = comm["ranks"][0]
    assert rec["rank"] == 0 and rec["comm_world"] == 1 and rec["rccl_version"] > 0 and ":" in rec["pci_bus_id"]
    assert line["config"]["blocks_verified"] == 1
    for mode in ("rccl", "rccl+d2h", "d2h", "h2d+d2h"):
        assert "block of every rank (1)" in line["gather"][mode]["blocks_verified"]


def test_bench_goes_on_when_the_rccl_bootstrap_misses_its_deadline():
    """A communicator that is not there in time (deadline 0 here) must not hang the bench: the step falls
    back to the host-memory gather and the line says why."""
    line = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--points", "200000", "--no-companion",
                   "--no-cpu-baseline"], {"PCX_BENCH_FORCE_COMM": "1", "PCX_BENCH_RCCL_TIMEOUT": "0"})
    assert line["config"]["gather"] == "d2h" and line["comm"]["backend"] is None
    assert "did not finish" in line["comm"]["rccl_error"]
    assert line["gather"]["d2h"]["value"] > 0 and "rccl" not in line["gather"]


def test_bench_launches_two_ranks_itself():
    """Plain `python bench.py --gpus 2`: the script starts the ranks (both on GPU 0 here)."""
    line = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--points", "200000", "--no-companion"],
                  {"PCX_BENCH_SHARE_DEVICE": "1"})
    assert line["n_gpus"] == 2 and line["config"]["gather"] == "d2h"
    assert line["comm"]["backend"] is None and "duplicate" in line["comm"]["rccl_error"]
    assert len(line["roofline"]["avg_launch_ms_per_rank"]) == 2
    assert line["gather"]["none"]["value"] > 0 and line["gather"]["d2h"]["value"] > 0
    assert "cpu_baseline" not in line                       # rank 0 at N = 1 only
    # round 4: one record per rank (both on GPU 0 here: one distinct bus id), both blocks verified, the host-to-host leg
    ranks = line["comm"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and len({r["pid"] for r in ranks}) == 2
    assert line["comm"]["distinct_gpus"] == 1 and line["comm"]["world"] == 2
    assert line["config"]["blocks_verified"] == 2
    assert line["gather"]["h2d+d2h"]["value"] > 0 and "block of every rank (2)" in line["gather"]["h2d+d2h"]["blocks_verified"]


def test_bench_under_the_driver_launcher():
    """How the driver starts the multi-GPU bench: `python -m torch.distributed.run --nproc-per-node N
    bench.py --gpus N` (torch only launches the processes; the ranks never import it).  Two ranks, both
    on GPU 0 here."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = str(sock.getsockname()[1])
    env = dict(os.environ, PCX_BENCH_SHARE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PCX_RDZV_DIR"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1", "--points", "200000", "--no-companion"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["comm"]["torch"] is False
    assert line["gather"]["d2h"]["value"] > 0 and len(line["roofline"]["avg_launch_ms_per_rank"]) == 2


def test_eval_sharded_two_ranks_on_one_gpu(tmp_path):
    """eval_sharded with the HIP evaluator: two processes, one GPU, result = one-process result."""
    worker = tmp_path / "w.py"
    worker.write_text(f"""
import os, sys
import numpy as np
sys.path.insert(0, {ROOT!r})
from pychebyshev_amd import ChebyshevApproximation
from pychebyshev_amd.distributed import HostGroup, eval_sharded
g = np.load(os.path.join({GOLDEN!r}, "g1_sincos2d.npz"))
cheb = ChebyshevApproximation.from_values(g["tensor"], 2, [[-1.0, 1.0], [-1.0, 1.0]], [12, 12])
group = HostGroup.from_env(timeout=120)
pts = np.random.default_rng(8).uniform(-1, 1, (50001, 2))
full = eval_sharded(lambda b: cheb.vectorized_eval_batch(b, [1, 0]), pts, group)
if group.rank == 0:
    assert np.array_equal(full, cheb.vectorized_eval_batch(pts, [1, 0]))
group.close()
print("ok", group.rank)
""")
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", PCX_RDZV_DIR=str(tmp_path / "rdzv"), PCX_DEVICE="0")
        procs.append(subprocess.Popen([sys.executable, str(worker)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for r, p in enumerate(procs):
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0 and f"ok {r}" in so, so[-1000:] + se[-3000:]
