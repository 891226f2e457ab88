#!/usr/bin/env python3
"""bench.py -- headline benchmark: 5-D Black-Scholes barycentric point-evals/sec on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  No PyTorch anywhere: the C ABI (libpcx_hip.so) does the compute, the RCCL gather
  (pcx_comm_*) and the copies; ranks meet through pychebyshev_amd.distributed.HostGroup.
  N = 1   : one process.
  N > 1   : one process per GPU.  Either the driver starts the ranks
            (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`, used
            as a plain process launcher: RANK / LOCAL_RANK / WORLD_SIZE come from the
            environment), or -- when WORLD_SIZE is unset -- this script starts N child
            ranks itself BEFORE touching the GPU and forwards rank 0's JSON line.

A "step" is one pass of the hot path over one batch already resident in HBM:
  workload bary5d (default, BASELINE.json configs[1]): 5-D Black-Scholes n=11^5 full
  tensor, 10^6 fp64 query points per GPU (seed 99 + rank, column-wise uniform), value
  spec.  Other workloads (--workload greeks5d | tt5d | tt10d) are the other BASELINE
  configs; config 5 per GPU is `--workload tt10d --points 12500000`.

Weak scaling: every rank owns its own batch; value = all point-evals of all ranks /
max-over-ranks wall time of the K steps (host barrier + device sync on both sides).  With
N > 1 each timed step ends with the RCCL gather of the N result blocks on rank 0's GPU (the
path's only exchange; it overlaps the next step's kernel).  The same K steps are timed
again (a) without any gather, (b) with rank 0 downloading the gathered result, and (c) with
every rank downloading its block straight into one pinned shared-memory host array (no
collective) -- reported under "gather" beside the headline.

Also on the JSON line:
  roofline     dominant kernel: algorithmic flop per launch / mean launch duration from
               HIP events on the launch stream, vs the FP64 MFMA peak; `traffic` is the
               HBM bytes per launch from the committed PMC profile (cached; see
               traffic_source), not collected in this run.
  end_to_end   the host-pointer entry point on the same batch (H2D + kernel + D2H) through the Python method (a fresh
               result array per call), with the PCIe GB/s; `preallocated`: the C-ABI call into result arrays the
               caller owns; `page_locked`: on caller arrays page-locked beforehand (pcx_host_register).
  greeks, tt, tt10d, c1
               companions: config 4 (6 derivative specs in ONE multi-spec call: delta and gamma share a
               GEMM -- the pairs the library's accuracy probe admits at 3e-13 -- roofline on the 5 executed
               GEMMs; `span0` (no sharing) and `tol_1e-12` (looser probe tolerance) beside it), config 3 (TT-Cross
               build + 10^7-point eval_batch; 200 timed steps after 50 warm-ups: a 0.8 ms step is
               inside the FP64 clock transient for the first ~30 ms), config 5's model at a per-GPU
               batch (10-D, rank 16, 12.5 M points = 10^8 / 8; `full_batch_one_gpu`: all 10^8 x 10 in ONE launch on one
               GPU) and config 1 (12 x 12, 10^4 points: microseconds
               per call, GPU and CPU) -- same timing discipline, never mixed into `value`.
  midsize      N = 1 only: device-resident value evaluations of mid-size full tensors (21^3 ... 65^3, 64^4: the shapes
               VERDICT r3 #6 names and their neighbours) on the kernel auto picks, HIP events around ten launches each:
               kernel, point-evals/s and the fraction of the FP64 matrix peak per shape.
  cpu_baseline the CPU oracle (C restatement of the reference, OpenMP) on all usable
               host cores, on the per-GPU CPU share (16) and the reference's NumPy shape on
               one core -- bounded samples of the same workload, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Host arrays this process page-locked with pcx_host_register stay registered -- and alive -- until it exits: a heap range
# that was registered and RELEASED has ended later calls over the same addresses in a GPU memory access fault
# (tools/soak.py --pin), and the legs that follow a page-locked leg here copy to and from fresh NumPy arrays.
_LOCKED_UNTIL_EXIT = []

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X FP64 matrix peak (= FP64 vector peak), AMD spec;
                                  # v_mfma_f64_16x16x4_f64 at 64 cycles/SIMD x 1024 SIMDs x 2.4 GHz
HBM_PEAK_GBS = 8000.0
METRIC = "point-evals/sec, 5D Black-Scholes n=11^5 barycentric + TT, 1/2/4/8 GPU"

# ----------------------------------------------------------------------------------
# synthetic workload definitions (own code; the reference's benchmark recipe is
# compare_methods_time_accuracy.py:36-43,64-72,233-254 and tests/conftest.py:86-100 there)
# ----------------------------------------------------------------------------------
BS5_DOMAIN = [[80.0, 120.0], [90.0, 110.0], [0.25, 1.0], [0.15, 0.35], [0.01, 0.08]]
BS5_NODES = [11, 11, 11, 11, 11]
BS_Q = 0.02
GREEK_SPECS = [[0, 0, 0, 0, 0],   # price
               [1, 0, 0, 0, 0],   # delta
               [2, 0, 0, 0, 0],   # gamma
               [0, 0, 0, 1, 0],   # vega
               [0, 0, 1, 0, 0],   # dV/dT
               [0, 0, 0, 0, 1]]   # rho
GREEK_NAMES = ["price", "delta", "gamma", "vega", "dV/dT", "rho"]


def bs_5d(x, _=None):
    """European call V(S, K, T, sigma, r), dividend yield 0.02 (math.erfc closed form)."""
    S, K, T, sigma, r = x
    sq = sigma * math.sqrt(T)
    d1 = (math.log(S / K) + (r - BS_Q + 0.5 * sigma * sigma) * T) / sq
    ncdf = lambda v: 0.5 * math.erfc(-v / math.sqrt(2.0))
    return S * math.exp(-BS_Q * T) * ncdf(d1) - K * math.exp(-r * T) * ncdf(d1 - sq)


def uniform_points(domain, n, seed):
    """One rng.uniform(lo, hi, n) per dimension, stacked column-wise (the reference's recipe)."""
    rng = np.random.default_rng(seed)
    return np.column_stack([rng.uniform(lo, hi, n) for lo, hi in domain])


def usable_cores() -> int:
    """Host cores this process may use: CPU affinity capped by a cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except (OSError, ValueError):
            pass
    return max(1, n)


# ----------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------
class Workload:
    name = ""
    key = ""
    d = 0
    points_per_gpu = 0
    evals_per_point = 1          # point-evals per query point per step (= kernel launches per step)
    flop_per_eval = 0.0          # algorithmic flop per point-eval
    bytes_per_eval = 0.0         # algorithmic HBM bytes per point-eval
    kernel = ""
    build_info = None


class Bary5D(Workload):
    d = 5
    flop_per_eval = 354310.0     # 177,155 FMA: 161051 + 14641 + 1331 + 121 + 11
    bytes_per_eval = 48.0        # 5 x 8 in + 8 out
    kernel = "k_bary_mfma<31,2,false,3>"

    def __init__(self, lib_mod, n_points, specs, key):
        from pychebyshev_amd import ChebyshevApproximation
        self._lib = lib_mod
        self.key = key
        self.points_per_gpu = n_points
        self.specs = [list(s) for s in specs]
        self.evals_per_point = len(self.specs)
        self.name = ("5D Black-Scholes n=11^5 full-tensor barycentric, %s fp64 queries per GPU%s"
                     % (f"{n_points:,}", "" if len(specs) == 1 else f" x {len(specs)} derivative specs (price + 5 Greeks)"))
        info = ChebyshevApproximation.nodes(5, BS5_DOMAIN, BS5_NODES)
        tensor = np.array([bs_5d(list(p)) for p in info["full_grid"]]).reshape(info["shape"])
        self.model = ChebyshevApproximation.from_values(tensor, 5, BS5_DOMAIN, BS5_NODES)
        self.model.to_device()
        self.m = self.model._model()
        self.spec_arrays = [lib_mod.i32(s) for s in self.specs]
        self.spec_block = lib_mod.i32(np.asarray(self.specs).reshape(-1))
        # GEMMs a step executes: pairs of specs one order apart along one dimension share one (the library's count)
        self.gemms_per_step = self.count_gemms()

    def points(self, rank):
        return uniform_points(BS5_DOMAIN, self.points_per_gpu, 99 + rank)

    def stream(self):
        st = ctypes.c_void_p()
        self._lib.check(self.m.lib.pcx_bary_stream(self.m.handle, ctypes.byref(st)), self.m.lib)
        return st

    def launch(self, d_pts, n, d_out, stream, which=None):
        """One spec (`which`, or a single-spec workload): one launch into out[which*n : (which+1)*n].  All specs of
        the Greeks workload: ONE multi-spec call into out (n x m, row-major) -- the library shares a GEMM between
        pairs of specs one order apart along one dimension (delta / gamma, price / vega; pcx_bary_set_group_span)."""
        if which is None and len(self.spec_arrays) > 1:
            self._lib.check(self.m.lib.pcx_bary_eval_multi_batch_dev(self.m.handle, d_pts, n, self._lib.p_i32(self.spec_block),
                                                                     len(self.spec_arrays), d_out, stream), self.m.lib)
            return
        for i, s in enumerate(self.spec_arrays):
            if which is not None and i != which:
                continue
            out_i = ctypes.c_void_p(d_out.value + i * n * 8)
            self._lib.check(self.m.lib.pcx_bary_eval_batch_dev(self.m.handle, d_pts, n, self._lib.p_i32(s),
                                                               out_i, stream), self.m.lib)

    def set_group_span(self, span):
        self._lib.check(self.m.lib.pcx_bary_set_group_span(self.m.handle, span), self.m.lib)
        self.gemms_per_step = self.count_gemms()

    def set_group_tolerance(self, tol):
        self._lib.check(self.m.lib.pcx_bary_set_group_tolerance(self.m.handle, tol), self.m.lib)
        self.gemms_per_step = self.count_gemms()

    def count_gemms(self):
        if len(self.specs) == 1:
            return 1
        out = self._lib.i32([0])
        self._lib.check(self.m.lib.pcx_bary_count_gemms(self.m.handle, self._lib.p_i32(self.spec_block), len(self.specs),
                                                        self.points_per_gpu, self._lib.p_i32(out)), self.m.lib)
        return int(out[0])

    def host_eval(self, pts):
        """The host-pointer entry point (what ChebyshevApproximation.vectorized_eval_batch /
        vectorized_eval_multi_batch call): one upload of the points whatever the number of specs."""
        if len(self.specs) > 1:
            return [self.model.vectorized_eval_multi_batch(pts, self.specs)]
        return [self.model.vectorized_eval_batch(pts, s) for s in self.specs]

    def host_outputs(self, n):
        return [np.empty((n, len(self.specs)) if len(self.specs) > 1 else n)]

    def host_eval_into(self, pts, outs):
        """The C-ABI host-pointer call straight into caller-owned arrays (what the Python method wraps)."""
        self._lib.check(self.m.lib.pcx_bary_eval_multi_batch(self.m.handle, self._lib.p_f64(pts), len(pts),
                                                             self._lib.p_i32(self.spec_block), len(self.specs),
                                                             self._lib.p_f64(outs[0])), self.m.lib)

    def cpu_rates(self, seconds):
        import oracle
        om = oracle.BaryModel(self.model.nodes, self.model.weights, self.model.diff_matrices,
                              self.model.tensor_values)
        pts = self.points(0)
        nspec = len(self.specs)

        def c_port(threads, budget):
            oracle.set_num_threads(threads)
            oracle.bary_eval_batch(om, pts[:4000], self.specs[0])           # thread start-up
            probe = 20000
            t0 = time.perf_counter()
            oracle.bary_eval_batch(om, pts[:probe], self.specs[0])
            rate = probe / (time.perf_counter() - t0)
            sample = int(min(len(pts), max(probe, rate * budget / nspec)))
            # the whole batch is shorter than the time budget on a many-core host: pass over it again
            passes = max(1, int(round(rate * budget / nspec / sample))) if sample == len(pts) else 1
            t0 = time.perf_counter()
            for _ in range(passes):
                for s in self.specs:
                    oracle.bary_eval_batch(om, pts[:sample], s)
            dt = time.perf_counter() - t0
            return {"value": sample * nspec * passes / dt, "unit": "point-evals/s", "cores": oracle.num_threads(),
                    "kind": "port",
                    "sample": f"first {sample} of rank 0's seed-99 points x {nspec} spec(s) x {passes} pass(es), {dt:.1f} s"}

        npn = 2000
        t0 = time.perf_counter()
        for s in self.specs[:1]:
            oracle.bary_eval_batch_numpy(om, pts[:npn], s)
        numpy_loop = {"value": npn / (time.perf_counter() - t0), "unit": "point-evals/s", "cores": 1,
                      "sample": f"first {npn} points, per-point NumPy matvec loop in the reference's shape "
                                "(barycentric.py:1035-1046), default BLAS threads"}
        return c_port, numpy_loop


class TTWork(Workload):
    def __init__(self, lib_mod, n_points, kind):
        from pychebyshev_amd import ChebyshevTT
        self._lib = lib_mod
        self.key = kind
        self.points_per_gpu = n_points
        if kind == "tt5d":
            # config 3 = TT-Cross build (max_rank 8, seed 42) + batched queries: the model is built
            # here, through the Python callback and the device-side cross steps, and timed
            g = np.load(os.path.join(ROOT, "tests", "golden", "g4_tt_bs5d.npz"))
            built = ChebyshevTT(bs_5d, 5, BS5_DOMAIN, BS5_NODES, max_rank=8)
            t0 = time.perf_counter()
            built.build(verbose=False, seed=42)
            self.build_info = {"method": "cross", "seconds": time.perf_counter() - t0,
                               "tt_ranks": list(built.tt_ranks), "unique_evals": int(built.total_build_evals),
                               "reference_ranks": [int(v) for v in g["r8_ranks"]],
                               "reference_unique_evals": int(g["r8_evals"])}
            cores = built._coeff_cores
            self.domain = BS5_DOMAIN
            self.name = "5D Black-Scholes ChebyshevTT ranks %s eval_batch, %s fp64 queries per GPU" % (
                str(list(built.tt_ranks)).replace(" ", ""), f"{n_points:,}")
            self.flop_per_eval = 2.0 * sum((11 + 1) * a * b for a, b in zip(built.tt_ranks[:-1], built.tt_ranks[1:]))
            self.bytes_per_eval = 48.0
            self.kernel = "k_tt_eval_lpp<8,11>"     # lane-per-point v_fma_f64 form (FP64 vector peak = matrix peak)
        else:
            rng = np.random.default_rng(16)
            ranks = [1] + [16] * 9 + [1]
            cores = [rng.standard_normal((ranks[k], 11, ranks[k + 1])) / np.sqrt(ranks[k] * 11) for k in range(10)]
            self.domain = [[-1.0, 1.0]] * 10
            self.name = "10D synthetic rank-16 ChebyshevTT eval_batch, %s fp64 queries per GPU%s" % (
                f"{n_points:,}", " (12.5 M = 10^8 / 8: config 5's per-GPU share)" if n_points == 12_500_000 else "")
            self.flop_per_eval, self.bytes_per_eval = 49920.0, 88.0
            self.kernel = "k_tt_eval_mfma<4,1,4>"
        self.d = len(cores)
        self.cores = cores
        self.model = ChebyshevTT.from_coeff_cores(cores, self.domain)
        self.model.to_device()
        self.t = self.model._dev()

    def points(self, rank):
        return uniform_points(self.domain, self.points_per_gpu, 99 + rank)

    def stream(self):
        st = ctypes.c_void_p()
        self._lib.check(self.t.lib.pcx_tt_stream(self.t.handle, ctypes.byref(st)), self.t.lib)
        return st

    def launch(self, d_pts, n, d_out, stream, which=None):
        self._lib.check(self.t.lib.pcx_tt_eval_batch_dev(self.t.handle, d_pts, n, d_out, stream), self.t.lib)

    def host_eval(self, pts):
        return [self.model.eval_batch(pts)]

    def host_outputs(self, n):
        return [np.empty(n)]

    def host_eval_into(self, pts, outs):
        self._lib.check(self.t.lib.pcx_tt_eval_batch(self.t.handle, self._lib.p_f64(pts), len(pts),
                                                     self._lib.p_f64(outs[0])), self.t.lib)

    def cpu_rates(self, seconds):
        import oracle
        pts = self.points(0)

        def c_port(threads, budget):
            oracle.set_num_threads(threads)
            probe = min(len(pts), 200_000)
            oracle.tt_eval_batch(self.cores, self.domain, pts[:20000])      # thread start-up
            t0 = time.perf_counter()
            oracle.tt_eval_batch(self.cores, self.domain, pts[:probe])
            rate = probe / (time.perf_counter() - t0)
            sample = int(min(len(pts), max(probe, rate * budget)))
            # the whole batch is shorter than the time budget on a many-core host: pass over it again
            passes = max(1, int(round(rate * budget / sample))) if sample == len(pts) else 1
            t0 = time.perf_counter()
            for _ in range(passes):
                oracle.tt_eval_batch(self.cores, self.domain, pts[:sample])
            dt = time.perf_counter() - t0
            return {"value": sample * passes / dt, "unit": "point-evals/s", "cores": oracle.num_threads(), "kind": "port",
                    "sample": f"first {sample} points of rank 0's batch x {passes} pass(es), {dt:.1f} s"}

        npn = min(len(pts), 200_000)
        t0 = time.perf_counter()
        oracle.tt_eval_batch_numpy(self.cores, self.domain, pts[:npn])
        numpy_loop = {"value": npn / (time.perf_counter() - t0), "unit": "point-evals/s", "cores": 1,
                      "sample": f"first {npn} points, chebval + einsum in the reference's shape "
                                "(tensor_train.py:2252-2263), default BLAS threads"}
        return c_port, numpy_loop


def make_workload(lib_mod, name, n_points):
    if name == "bary5d":
        return Bary5D(lib_mod, n_points or 1_000_000, GREEK_SPECS[:1], "bary5d")
    if name == "greeks5d":
        return Bary5D(lib_mod, n_points or 1_000_000, GREEK_SPECS, "greeks5d")
    if name in ("tt5d", "tt10d"):
        # config 3: 10^7 points; config 5: 10^8 points over 8 GPUs = 12.5 M per GPU (1 GB of coordinates)
        return TTWork(lib_mod, n_points or (10_000_000 if name == "tt5d" else 12_500_000), name)
    raise SystemExit(f"unknown workload {name}")


# ----------------------------------------------------------------------------------
# launcher: N child ranks, started before anything touches the GPU
# ----------------------------------------------------------------------------------
def launch_children(n: int, script: str | None = None, argv=None) -> int:
    """Start n ranks of `script` (this file by default) with the environment a rank expects, forward rank 0's JSON
    line, stop the others if one fails.  `script` / `argv` exist for the CPU rehearsal of this launcher
    (tests/test_distributed_cpu.py runs a stand-in rank script through it with 8 ranks)."""
    from pychebyshev_amd import _build
    if script is None and _build.needs_build():                 # hipcc only; no GPU call
        _build.build()
    script = script or os.path.abspath(__file__)
    argv = sys.argv[1:] if argv is None else list(argv)
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    rdzv = tempfile.mkdtemp(prefix="pcx_rdzv_", dir=base)
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       PCX_RDZV_DIR=rdzv, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
            env.setdefault("OMP_NUM_THREADS", "2")
            procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
        out0 = b""
        rc = 0
        alive = set(range(n))
        buf = []
        import selectors
        sel = selectors.DefaultSelector()
        sel.register(procs[0].stdout, selectors.EVENT_READ)
        eof = False
        while alive:
            if not eof:
                for key, _ in sel.select(timeout=0.2):
                    chunk = os.read(key.fileobj.fileno(), 65536)
                    if chunk:
                        buf.append(chunk)
                    else:
                        eof = True
                        sel.unregister(key.fileobj)
            else:
                time.sleep(0.2)
            for r in list(alive):
                code = procs[r].poll()
                if code is not None:
                    alive.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        sys.stderr.write(f"bench.py: rank {r} exited with code {code}; stopping the other ranks\n")
                        for q in alive:       # the exact PIDs this launcher started
                            procs[q].terminate()
        out0 = b"".join(buf) + (procs[0].stdout.read() or b"")
        lines = [ln for ln in out0.decode("utf-8", "replace").splitlines() if ln.strip().startswith("{")]
        if rc == 0 and lines:
            sys.stdout.write(lines[-1] + "\n")
            sys.stdout.flush()
        elif rc == 0:
            rc = 1
            sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return rc
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(rdzv, ignore_errors=True)


# ----------------------------------------------------------------------------------
def run_rank(args) -> int:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    args.gpus = world
    share_device = os.environ.get("PCX_BENCH_SHARE_DEVICE") == "1"      # rehearsal: every rank on GPU 0, no RCCL
    force_comm = os.environ.get("PCX_BENCH_FORCE_COMM") == "1"          # rehearsal: RCCL + shared result with 1 rank
    # The contract is ONE JSON line on stdout.  RCCL prints banners to the C-level stdout
    # at init, so keep the real stdout aside and point fd 1 at stderr meanwhile.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    dev = 0 if share_device else local_rank
    os.environ["PCX_DEVICE"] = str(dev)
    from pychebyshev_amd import _build, _lib
    from pychebyshev_amd.distributed import HostGroup, RcclComm, SharedResult, shard_table
    if rank == 0 and _build.needs_build():    # fresh checkout: the .so is git-ignored
        _build.build()
    group = HostGroup.from_env(timeout=600) if (world > 1 or force_comm) else None
    lib = _lib.load()
    ndev = _lib.device_count()
    if not share_device and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} HIP device(s) visible")

    comm, rccl_error = None, None
    stuck_init = False
    boot_group = None
    if group is not None and not share_device:
        # ncclCommInitRank is collective: if it never returns (a rank that died, a fabric problem) the bench
        # would hang without a line.  Bootstrap on a helper thread with a deadline; past it the run goes on
        # with the host-memory gather and says so on the line.
        import threading
        box = {}
        # its own rendezvous: a bootstrap that never returns must not share generation counters with the
        # collectives the main thread goes on to run
        boot_group = group.subgroup("rccl_boot")

        def bootstrap():
            try:
                box["comm"] = RcclComm(boot_group, dev)
            except Exception as exc:                              # reported on the line, never silent
                box["error"] = f"{type(exc).__name__}: {exc}"

        th = threading.Thread(target=bootstrap, daemon=True)
        th.start()
        deadline = float(os.environ.get("PCX_BENCH_RCCL_TIMEOUT", "180"))
        th.join(deadline)
        if th.is_alive():
            stuck_init = True
            rccl_error = f"RCCL initialisation did not finish within {deadline:g} s"
        else:
            comm, rccl_error = box.get("comm"), box.get("error")
        flags = group.gather_floats(0.0 if comm is not None else 1.0)
        if any(flags) and comm is not None:                       # a communicator on some ranks only is useless
            comm.close()
            comm = None
            rccl_error = "RCCL initialisation failed on another rank"
        if comm is None and rank == 0:
            sys.stderr.write(f"bench.py: RCCL unavailable ({rccl_error}); gather falls back to shared host memory\n")
    elif share_device and world > 1:
        rccl_error = "PCX_BENCH_SHARE_DEVICE=1: all ranks on GPU 0, RCCL refuses duplicate devices"

    def chk(rc):
        _lib.check(rc, lib)

    def rank_record():
        """What this rank ran on: answers "did RCCL see N ranks on N distinct GPUs" from the line alone."""
        import socket
        rec = {"rank": rank, "local_rank": local_rank, "device": dev, "pid": os.getpid(), "host": socket.gethostname()[:24]}
        buf = ctypes.create_string_buffer(64)
        if lib.pcx_device_pci_bus_id(dev, buf, 64) == 0:
            rec["pci_bus_id"] = buf.value.decode()
        if comm is not None:
            r_, w_, d_, v_ = ctypes.c_int32(-1), ctypes.c_int32(-1), ctypes.c_int32(-1), ctypes.c_int32(0)
            if lib.pcx_comm_info(comm.handle, ctypes.byref(r_), ctypes.byref(w_), ctypes.byref(d_), ctypes.byref(v_)) == 0:
                rec.update({"comm_rank": int(r_.value), "comm_world": int(w_.value), "comm_device": int(d_.value),
                            "rccl_version": int(v_.value)})
        return rec

    def dev_sync():
        chk(lib.pcx_device_synchronize(dev))

    def barrier():
        dev_sync()
        if group is not None:
            group.barrier()

    live_events = []

    def new_event():
        e = ctypes.c_void_p()
        chk(lib.pcx_event_create(dev, ctypes.byref(e)))
        live_events.append(e)
        return e

    def free_events():
        for e in live_events:
            lib.pcx_event_destroy(e)
        live_events.clear()

    def elapsed_ms(a, b):
        ms = ctypes.c_float()
        chk(lib.pcx_event_elapsed_ms(a, b, ctypes.byref(ms)))
        return float(ms.value)

    copy_stream = ctypes.c_void_p()
    chk(lib.pcx_stream_create(dev, ctypes.byref(copy_stream)))
    side_stream = comm.stream() if comm is not None else copy_stream

    def block_crc(arr):
        import zlib
        return float(zlib.crc32(np.ascontiguousarray(arr).view(np.uint8)))

    def measure(wl, steps, warmup, mode):
        """W untimed + K timed steps of one workload, batch resident in HBM beforehand.
        mode: "none" (kernel only), "rccl" (gather of all blocks on rank 0's GPU each step),
        "rccl+d2h" (... and rank 0 downloads the gathered result), "d2h" (every rank downloads
        its block into the shared pinned host array), "h2d+d2h" (every step ALSO uploads the rank's points from
        page-locked host memory first: the whole host-to-host path on every GPU at once).
        Returns a dict; times are max over ranks."""
        n = wl.points_per_gpu
        n_out = n * wl.evals_per_point
        pts = np.ascontiguousarray(wl.points(rank))
        upload = mode == "h2d+d2h"
        # two result buffers: the gather / download of step i (side stream) overlaps the
        # kernel of step i+1, which writes the other buffer
        nbuf = 1 if mode == "none" else 2
        d_ptss = []
        for _ in range(nbuf if upload else 1):
            p = ctypes.c_void_p()
            chk(lib.pcx_dev_malloc(dev, pts.nbytes, ctypes.byref(p)))
            chk(lib.pcx_memcpy_h2d(dev, p, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes))
            d_ptss.append(p)
        pts_locked = upload and lib.pcx_host_register(dev, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes) == 0
        d_outs = []
        for _ in range(nbuf):
            p = ctypes.c_void_p()
            chk(lib.pcx_dev_malloc(dev, n_out * 8, ctypes.byref(p)))
            d_outs.append(p)
        stream = wl.stream()
        counts, offsets = shard_table(n * world, world, width=wl.evals_per_point)
        d_full, shared = [], None
        if mode in ("rccl", "rccl+d2h") and rank == 0:
            for _ in range(nbuf):
                p = ctypes.c_void_p()
                chk(lib.pcx_dev_malloc(dev, n_out * world * 8, ctypes.byref(p)))
                d_full.append(p)
        if mode in ("d2h", "rccl+d2h", "h2d+d2h"):
            shared = SharedResult(group, n_out * world, device=dev, name="bench_" + wl.key)
        busy = [None] * nbuf          # event: the side stream has finished with this result buffer
        kdone = [None] * nbuf         # event: the kernel that read this points buffer has finished (uploads only)
        count = [0]
        g_events = []

        def step(kev=None):
            slot = count[0] % nbuf
            count[0] += 1
            d_pts = d_ptss[slot % len(d_ptss)]
            if upload:
                if kdone[slot] is not None:
                    chk(lib.pcx_stream_wait_event(copy_stream, kdone[slot]))   # the points buffer is free again
                chk(lib.pcx_memcpy_h2d_async(d_pts, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes, copy_stream))
                up = new_event()
                chk(lib.pcx_event_record(up, copy_stream))
                chk(lib.pcx_stream_wait_event(stream, up))
            if busy[slot] is not None:
                chk(lib.pcx_stream_wait_event(stream, busy[slot]))     # kernel must not overwrite a block in flight
            if kev is not None:
                chk(lib.pcx_event_record(kev[0], stream))
            wl.launch(d_pts, n, d_outs[slot], stream)
            if kev is not None:
                chk(lib.pcx_event_record(kev[1], stream))
            if mode == "none":
                return
            done = kev[1] if kev is not None else new_event()
            if kev is None:
                chk(lib.pcx_event_record(done, stream))
            kdone[slot] = done
            chk(lib.pcx_stream_wait_event(side_stream, done))
            g0 = g1 = None
            if kev is not None:
                g0, g1 = new_event(), new_event()
                chk(lib.pcx_event_record(g0, side_stream))
            if mode in ("rccl", "rccl+d2h"):
                comm.gatherv_dev(d_outs[slot], d_full[slot] if rank == 0 else None, counts, offsets, 0, side_stream)
                if mode == "rccl+d2h" and rank == 0:
                    chk(lib.pcx_memcpy_d2h_async(ctypes.c_void_p(shared.address(0)), d_full[slot],
                                                 n_out * world * 8, side_stream))
            else:
                chk(lib.pcx_memcpy_d2h_async(ctypes.c_void_p(shared.address(int(offsets[rank]))), d_outs[slot],
                                             n_out * 8, side_stream))
            ev = new_event()
            chk(lib.pcx_event_record(ev, side_stream))
            if kev is not None:
                g1 = ev
                g_events.append((g0, g1))
            busy[slot] = ev

        kevs = [(new_event(), new_event()) for _ in range(steps)]
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(kevs[i])
        barrier()                              # device sync: every kernel, gather and download has completed
        elapsed = time.perf_counter() - t0
        elapsed = group.max(elapsed) if group is not None else elapsed
        kernel_ms = [elapsed_ms(a, b) for a, b in kevs]
        gather_ms = [elapsed_ms(a, b) for a, b in g_events]
        # sanity: the last step's results are finite and, collected, complete -- EVERY rank's block of the gathered /
        # shared result must equal what that rank itself downloads from its own GPU (CRC-32 of the bytes)
        got = np.empty(n_out)
        chk(lib.pcx_memcpy_d2h(dev, got.ctypes.data_as(ctypes.c_void_p), d_outs[(count[0] - 1) % nbuf], n_out * 8))
        if not np.isfinite(got).all():
            raise SystemExit(f"non-finite results in the benchmark batch ({wl.name})")
        verified = None
        if group is not None and mode != "none":
            own = group.gather_floats(block_crc(got))
            if rank == 0:
                if mode in ("rccl", "rccl+d2h"):
                    full = np.empty(n_out * world)
                    chk(lib.pcx_memcpy_d2h(dev, full.ctypes.data_as(ctypes.c_void_p), d_full[(count[0] - 1) % nbuf],
                                           full.nbytes))
                    bad = [r for r in range(world)
                           if block_crc(full[int(offsets[r]): int(offsets[r] + counts[r])]) != own[r]]
                    if bad or not np.isfinite(full).all():
                        raise SystemExit(f"RCCL-gathered result: the blocks of ranks {bad} differ from what those ranks computed")
                if shared is not None:
                    host = np.array(shared.array, copy=True)
                    bad = [r for r in range(world)
                           if block_crc(host[int(offsets[r]): int(offsets[r] + counts[r])]) != own[r]]
                    if bad or not np.isfinite(host).all():
                        raise SystemExit(f"shared host result: the blocks of ranks {bad} differ from what those ranks computed")
                verified = world
        pinned = shared.pinned if shared is not None else None
        if shared is not None:
            shared.close()
        if pts_locked:
            _LOCKED_UNTIL_EXIT.append(pts)      # stays registered (see _LOCKED_UNTIL_EXIT)
        for p in d_ptss + d_outs + d_full:
            lib.pcx_dev_free(dev, p)
        free_events()
        launches = getattr(wl, "gemms_per_step", wl.evals_per_point)
        avg_launch = float(np.mean(kernel_ms)) / launches
        rec = {"elapsed": elapsed, "kernel_ms": kernel_ms, "avg_launch_ms": avg_launch,
               "avg_launch_ms_per_rank": group.gather_floats(avg_launch) if group is not None else [avg_launch],
               "side_ms": float(np.mean(gather_ms)) if gather_ms else None, "pinned": pinned,
               "blocks_verified": verified, "points_page_locked": pts_locked if upload else None}
        return rec

    def rate(wl, rec, steps):
        return float(wl.points_per_gpu) * wl.evals_per_point * world * steps / rec["elapsed"]

    def sustained_of(wl, frac):
        """the vector-pipe kernel against what a stream of ITS OWN operand pattern sustains with nothing
        around it (profiles/r04_tt_w4_lab.txt: 88 scalar operands per 97 FP64 instructions through s_load_dwordx16: 0.96 of
        the nominal instruction rate at 2.37 GHz -- not measured in this run) times its instruction mix (2,280 algorithmic
        FMAs in 2,469 vector instructions per 64 points)."""
        if "k_tt_eval_lpp<8,11>" not in wl.kernel:
            return {}
        ceiling = 0.96 * 2280.0 / 2469.0
        return {"sustained": {"stream_frac_of_peak": 0.96, "algorithmic_fma_per_vector_instruction": 2280.0 / 2469.0,
                              "ceiling_frac": ceiling, "source": "cached: profiles/r04_tt_w4_lab.txt (tools/tt_w4_lab.hip, lpp-like stream)"},
                "frac_of_sustained": frac / ceiling}

    def roofline_of(wl, rec):
        n = wl.points_per_gpu
        avg_launch_s = rec["avg_launch_ms"] / 1e3
        flop_per_launch = wl.flop_per_eval * n
        achieved = flop_per_launch / avg_launch_s / 1e12
        traffic, source = None, None
        executed, executed_basis = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            try:
                r = json.load(open(pmc_path)).get(wl.key)
                if r and r.get("points") == n:
                    traffic = r.get("hbm_bytes_per_launch")
                    source = f"cached PMC (not collected in this run): {r.get('source')}"
                    executed, executed_basis = r.get("executed_flop_per_launch"), r.get("executed_basis")
            except Exception:
                traffic = None
        lpp = "lpp" in wl.kernel
        pipe = ("FP64 VALU (v_fma_f64, lane per point): the FP64 vector peak equals the FP64 matrix peak on MI355X and the two "
                "share one pipe, so the same 78.6 TFLOP/s roofline binds") if lpp else "FP64 MFMA (v_mfma_f64_16x16x4_f64)"
        # `frac` prices the ALGORITHMIC flop; `executed_frac` the flop the FP64 pipe was actually asked for
        # (matrix kernels: the MFMA flop counter -- the folded-K GEMM needs 6 % fewer FMAs than the reference's nested
        # reduction, so it is BELOW frac; vector kernels: every vector instruction as a 64-lane FMA, i.e. issue-slot
        # occupancy at the nominal clock, ABOVE frac), from the committed PMC profile of the same batch size
        return {"bound": "valu_f64" if lpp else "mfma", "pipe": pipe, "kernel": wl.kernel, "achieved": achieved,
                "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                "executed_frac": (executed / avg_launch_s / 1e12 / FP64_MFMA_PEAK_TFLOPS) if executed else None,
                "executed_flop_per_launch": executed, "executed_basis": executed_basis,
                "avg_launch_ms": avg_launch_s * 1e3,
                "algorithmic_flop_per_launch": flop_per_launch,
                "algorithmic_hbm_bytes_per_launch": wl.bytes_per_eval * n,
                "hbm_frac": wl.bytes_per_eval * n / avg_launch_s / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": source,
                **sustained_of(wl, achieved / FP64_MFMA_PEAK_TFLOPS)}

    def end_to_end(wl, reps=3):
        """Host-pointer entry point on rank 0's batch: pageable NumPy in, NumPy out; then the same call on arrays the
        caller page-locked beforehand (pcx_host_register: asynchronous copies at PCIe rate, registration not timed)."""
        pts = np.ascontiguousarray(wl.points(rank))
        wl.host_eval(pts[: min(len(pts), 65536)])                  # staging buffers, derivative tensors
        res = wl.host_eval(pts)
        t0 = time.perf_counter()
        for _ in range(reps):
            wl.host_eval(pts)
        dt = (time.perf_counter() - t0) / reps
        moved = pts.nbytes + sum(np.asarray(r).nbytes for r in res)
        out = {"value": len(pts) * wl.evals_per_point / dt, "unit": "point-evals/s", "ms_per_call": dt * 1e3,
               "pcie_gb_per_s": moved / dt / 1e9,
               "what": "host-pointer C-ABI call on the same batch: pageable H2D + kernel + D2H inclusive"}
        if hasattr(wl, "host_eval_into"):
            try:
                # the same C-ABI call into result arrays the caller already owns (touched once): what is left when the
                # host's page faults on a freshly allocated result array are taken out
                outs = wl.host_outputs(len(pts))
                for a in outs:
                    a.fill(0.0)
                wl.host_eval_into(pts, outs)
                t0 = time.perf_counter()
                for _ in range(reps):
                    wl.host_eval_into(pts, outs)
                dtq = (time.perf_counter() - t0) / reps
                out["preallocated"] = {"value": len(pts) * wl.evals_per_point / dtq, "ms_per_call": dtq * 1e3,
                                       "pcie_gb_per_s": moved / dtq / 1e9,
                                       "what": "pageable arrays again, results into arrays the caller allocated (and touched) before"}
                regs = []
                t0 = time.perf_counter()
                for a in [pts] + outs:
                    if lib.pcx_host_register(dev, a.ctypes.data_as(ctypes.c_void_p), a.nbytes) == 0:
                        regs.append(a)
                t_reg = time.perf_counter() - t0
                if len(regs) == 1 + len(outs):
                    wl.host_eval_into(pts, outs)
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        wl.host_eval_into(pts, outs)
                    dtp = (time.perf_counter() - t0) / reps
                    out["page_locked"] = {"value": len(pts) * wl.evals_per_point / dtp, "ms_per_call": dtp * 1e3,
                                          "pcie_gb_per_s": moved / dtp / 1e9, "register_ms_not_timed": t_reg * 1e3,
                                          "what": "the same call on caller arrays page-locked once with pcx_host_register"}
                _LOCKED_UNTIL_EXIT.extend(regs)     # stay registered (see _LOCKED_UNTIL_EXIT)
            except Exception as exc:                                 # noqa: BLE001
                out["page_locked"] = {"error": f"{type(exc).__name__}: {exc}"}
        return out

    def cpu_baseline(wl, seconds):
        c_port, numpy_loop = wl.cpu_rates(seconds)
        cores = int(os.environ.get("PCX_CPU_THREADS", "0")) or usable_cores()
        rec = c_port(cores, seconds)
        rec["host_cpus"] = os.cpu_count()
        rec["cores_note"] = ("all host cores this process may use: min(CPU affinity %d, cgroup cpu.max quota) = %d"
                             % (len(os.sched_getaffinity(0)), usable_cores()))
        if cores > 16:
            share = c_port(16, min(seconds, 4.0))
            rec["per_gpu_cpu_share"] = {k: share[k] for k in ("value", "unit", "cores", "sample")}
        rec["numpy_loop"] = numpy_loop
        return rec

    def gather_report(wl, steps, warmup, headline):
        """The same K steps under every way of collecting the result blocks."""
        out = {}
        modes = [("none", "no gather (kernel launches only)")]
        if comm is not None:
            modes += [("rccl", "RCCL gather on rank 0's GPU each step, overlapping the next launch"),
                      ("rccl+d2h", "RCCL gather, then rank 0 downloads the full result to pinned host memory")]
        modes += [("d2h", "no collective: every rank downloads its block into one pinned shared-memory array"),
                  ("h2d+d2h", "host to host: every rank uploads its points from page-locked memory, evaluates, downloads its "
                              "block into the shared array -- all ranks on PCIe at once (SURVEY App. D.5 (ii))")]
        for mode, what in modes:
            rec = headline if mode == headline_mode else measure(wl, steps, warmup, mode)
            out[mode] = {"what": what, "value": rate(wl, rec, steps), "ms_per_step": rec["elapsed"] / steps * 1e3,
                         "side_stream_ms_per_step": rec["side_ms"]}
            if rec["pinned"] is not None:
                out[mode]["host_buffer_pinned"] = rec["pinned"]
            if rec.get("points_page_locked") is not None:
                out[mode]["points_page_locked"] = rec["points_page_locked"]
            if rec.get("blocks_verified") is not None:
                out[mode]["blocks_verified"] = ("block of every rank (%d) in the collected result = that rank's own download, "
                                                "CRC-32" % rec["blocks_verified"])
        return out

    def config1_companion():
        """BASELINE config 1 (2-D sin(x)cos(y) on [-1,1]^2, 12 x 12 nodes, 10^4 points, default_rng(1) -- the reference's
        own CPU-runnable case): the model is built through the Python callback API, then one call =
        one batch of 10^4 points.  GPU: microseconds per host-pointer call and per device-resident launch; CPU beside
        it: the C port and the per-point NumPy loop in the reference's shape, microseconds per point."""
        from pychebyshev_amd import ChebyshevApproximation
        c = ChebyshevApproximation(lambda x, _=None: math.sin(x[0]) * math.cos(x[1]), 2, [[-1.0, 1.0], [-1.0, 1.0]], [12, 12])
        t0 = time.perf_counter()
        c.build(verbose=False)
        build_s = time.perf_counter() - t0
        n = 10_000
        pts = np.random.default_rng(1).uniform(-1.0, 1.0, (n, 2))
        y = c.vectorized_eval_batch(pts, [0, 0])
        err = float(np.max(np.abs(y - np.sin(pts[:, 0]) * np.cos(pts[:, 1]))))
        reps = 300
        t0 = time.perf_counter()
        for _ in range(reps):
            c.vectorized_eval_batch(pts, [0, 0])
        host_us = (time.perf_counter() - t0) / reps * 1e6
        m = c._model()
        d_pts, d_out = ctypes.c_void_p(), ctypes.c_void_p()
        chk(lib.pcx_dev_malloc(dev, pts.nbytes, ctypes.byref(d_pts)))
        chk(lib.pcx_dev_malloc(dev, n * 8, ctypes.byref(d_out)))
        chk(lib.pcx_memcpy_h2d(dev, d_pts, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes))
        st = ctypes.c_void_p()
        chk(lib.pcx_bary_stream(m.handle, ctypes.byref(st)))
        spec = _lib.i32([0, 0])
        for _ in range(10):
            chk(lib.pcx_bary_eval_batch_dev(m.handle, d_pts, n, _lib.p_i32(spec), d_out, st))
        a, b = new_event(), new_event()
        chk(lib.pcx_event_record(a, st))
        for _ in range(reps):
            chk(lib.pcx_bary_eval_batch_dev(m.handle, d_pts, n, _lib.p_i32(spec), d_out, st))
        chk(lib.pcx_event_record(b, st))
        chk(lib.pcx_stream_synchronize(st))
        res_us = elapsed_ms(a, b) / reps * 1e3
        lib.pcx_dev_free(dev, d_pts)
        lib.pcx_dev_free(dev, d_out)
        free_events()
        info = _lib.i32(np.zeros(6))
        lib.pcx_bary_kernel_info(m.handle, _lib.p_i32(info))
        out = {"workload": "2D sin(x)cos(y) on [-1,1]^2, n=12x12 barycentric, 10,000 fp64 queries (default_rng(1)), built "
                           "through the Python callback API", "points_per_call": n, "build_seconds": build_s,
               "max_abs_error_vs_function": err,
               "gpu": {"kernel": {4: "k_bary_small<1,12>", 5: "k_bary_sq<12,0>"}.get(int(info[0]), f"variant {int(info[0])}"),
                       "host_pointer_us_per_call": host_us, "host_pointer_point_evals_per_s": n / host_us * 1e6,
                       "resident_us_per_launch": res_us, "resident_point_evals_per_s": n / res_us * 1e6,
                       "note": "one call is launch-latency-bound at 10^4 points: 156 FMA and 24 B per point "
                               "(at 4x10^6 points the same kernel runs 5x10^10 points/s, profiles/r02_bary_rate_probe.txt)"}}
        if not args.no_cpu_baseline:
            import oracle
            om = oracle.BaryModel(c.nodes, c.weights, c.diff_matrices, c.tensor_values)
            cpu = {}
            for label, threads in (("c_port_all_cores", usable_cores()), ("c_port_one_core", 1)):
                oracle.set_num_threads(threads)
                oracle.bary_eval_batch(om, pts, [0, 0])
                t0 = time.perf_counter()
                k = 0
                while time.perf_counter() - t0 < 1.0:
                    oracle.bary_eval_batch(om, pts, [0, 0])
                    k += 1
                dt = time.perf_counter() - t0
                cpu[label] = {"us_per_point": dt / (k * n) * 1e6, "point_evals_per_s": k * n / dt, "cores": oracle.num_threads(),
                              "kind": "port", "sample": f"{k} passes over the 10,000 points, {dt:.1f} s"}
            t0 = time.perf_counter()
            oracle.bary_eval_batch_numpy(om, pts, [0, 0])
            dt = time.perf_counter() - t0
            cpu["numpy_loop"] = {"us_per_point": dt / n * 1e6, "point_evals_per_s": n / dt, "cores": 1,
                                 "sample": "the 10,000 points once, per-point NumPy matvec loop in the reference's shape "
                                           "(barycentric.py:1035-1046)"}
            out["cpu"] = cpu
        return out

    multi = group is not None
    ranks_on = None
    if multi:
        blobs = group.allgather_bytes(json.dumps(rank_record(), separators=(",", ":")).encode()[:480])
        if rank == 0:
            ranks_on = []
            for b_ in blobs:
                try:
                    ranks_on.append(json.loads(b_.decode()))
                except ValueError:
                    ranks_on.append({"error": "unreadable rank record"})
    headline_mode = "none" if not multi else ("rccl" if comm is not None else "d2h")
    if args.gather != "auto":
        if args.gather in ("rccl", "rccl+d2h") and comm is None:
            raise SystemExit(f"--gather {args.gather} needs RCCL: {rccl_error}")
        if args.gather != "none" and not multi:
            raise SystemExit("--gather needs more than one rank (or PCX_BENCH_FORCE_COMM=1)")
        headline_mode = args.gather

    wl = make_workload(_lib, args.workload, args.points)
    if args.variant and hasattr(wl, "m"):
        chk(wl.m.lib.pcx_bary_set_kernel(wl.m.handle, args.variant))
    n = wl.points_per_gpu
    head = measure(wl, args.steps, args.warmup, headline_mode)
    gathers = gather_report(wl, args.steps, args.warmup, head) if multi else None

    def midsize_companion():
        """Mid-size full tensors (VERDICT r3 #6 names 21^3, 30^3, 40^3, 65^3, 64^4): device-resident vectorized_eval_batch of
        random-value tensors on the kernel auto picks, timed with HIP events on the handle's stream; frac = the reference's
        nested-reduction flop count (barycentric.py:1035-1046) over the FP64 matrix peak.  Parity of these kernel forms:
        tests/test_gpu_bary.py::test_grid_plans_against_oracle."""
        from pychebyshev_amd import ChebyshevApproximation
        rows = []
        for shape in ((21,) * 3, (30,) * 3, (32,) * 3, (40,) * 3, (48,) * 3, (64,) * 3, (65,) * 3, (64,) * 4):
            d = len(shape)
            rng = np.random.default_rng(d * 1000 + shape[0])
            c = ChebyshevApproximation.from_values(rng.standard_normal(shape), d, [[-1.0, 1.0]] * d, list(shape))
            c.to_device()
            m = c._model()
            kinfo, ginfo = (ctypes.c_int32 * 6)(), (ctypes.c_int32 * 4)()
            chk(lib.pcx_bary_kernel_info(m.handle, kinfo))
            chk(lib.pcx_bary_grid_info(m.handle, ginfo))
            npts = 1_000_000 if int(np.prod(shape)) < 4_000_000 else 125_000
            pts = rng.uniform(-1, 1, (npts, d))
            d_pts, d_out = ctypes.c_void_p(), ctypes.c_void_p()
            chk(lib.pcx_dev_malloc(dev, pts.nbytes, ctypes.byref(d_pts)))
            chk(lib.pcx_dev_malloc(dev, npts * 8, ctypes.byref(d_out)))
            try:
                chk(lib.pcx_memcpy_h2d(dev, d_pts, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes))
                st = ctypes.c_void_p()
                chk(lib.pcx_bary_stream(m.handle, ctypes.byref(st)))
                spec = _lib.i32([0] * d)
                reps = 10
                for _ in range(3):
                    chk(lib.pcx_bary_eval_batch_dev(m.handle, d_pts, npts, _lib.p_i32(spec), d_out, st))
                e0, e1 = new_event(), new_event()
                chk(lib.pcx_event_record(e0, st))
                for _ in range(reps):
                    chk(lib.pcx_bary_eval_batch_dev(m.handle, d_pts, npts, _lib.p_i32(spec), d_out, st))
                chk(lib.pcx_event_record(e1, st))
                dev_sync()
                ms = elapsed_ms(e0, e1) / reps
            finally:
                lib.pcx_dev_free(dev, d_pts)
                lib.pcx_dev_free(dev, d_out)
                free_events()
            fma, size = 0, int(np.prod(shape))
            for n_ in reversed(shape):
                fma += size
                size //= n_
            kern = {4: "k_bary_small", 5: "k_bary_sq", 1: "k_bary_rows"}.get(
                int(kinfo[0]), {1: "k_bary_mfma_grid", 2: "k_bary_mfma_kfold"}.get(int(ginfo[0]), "k_bary_mfma"))
            rows.append({"shape": "x".join(str(v) for v in shape), "kernel": kern, "points": npts, "ms_per_launch": ms,
                         "value": npts / (ms * 1e-3), "unit": "point-evals/s",
                         "frac": 2.0 * fma * npts / (ms * 1e-3) / (FP64_MFMA_PEAK_TFLOPS * 1e12)})
            del c, m
        return {"what": "device-resident value evaluations of mid-size full tensors, auto kernel choice (DESIGN 3.1d-f)",
                "peak": FP64_MFMA_PEAK_TFLOPS, "peak_unit": "TFLOP/s", "shapes": rows}

    def companion(name):
        cwl = make_workload(_lib, name, 0)
        # a TT step is 0.9 ms (3 ms for the 10-D model): 20 of them end inside the clock transient that follows the
        # start of a power-limited FP64 kernel stream (the first launches run ~6 % fast, the next ~10 % slow, steady
        # after ~30 ms: profiles/r03_tt5d_dispatch_times.txt).  The TT companions therefore warm up and time longer.
        c_steps, c_warm = args.steps, args.warmup
        if name == "tt5d":
            c_steps, c_warm = max(args.steps, 200), max(args.warmup, 50)
        elif name == "tt10d":
            c_steps, c_warm = max(args.steps, 60), max(args.warmup, 15)
        rec = measure(cwl, c_steps, c_warm, headline_mode)
        if rank != 0 and name != "greeks5d":
            return None
        out = {"workload": cwl.name, "points_per_gpu_per_step": cwl.points_per_gpu,
               "evals_per_point": cwl.evals_per_point, "value": rate(cwl, rec, c_steps),
               "unit": "point-evals/s", "ms_per_step": rec["elapsed"] / c_steps * 1e3, "steps": c_steps, "warmup": c_warm,
               "roofline": roofline_of(cwl, rec)}
        if name == "greeks5d":
            g = cwl.gemms_per_step
            out["roofline"]["flop_basis"] = (f"EXECUTED GEMMs: {g} per step for {len(cwl.specs)} specs (pairs one order apart along one "
                                             "dimension share a slab-packed GEMM, +4.8 % row tiles, when the library's probe "
                                             "measures the derived member within 3e-13 of its own GEMM: delta / gamma here); "
                                             f"avg_launch_ms = step / {g}; `value` counts all {len(cwl.specs)} specs")
            out["config"] = {"group_tolerance": 3e-13, "gemms_per_step": g,
                             "parity": "every spec within 1e-12 (normwise) of the reference's batch results, shared ones "
                                       "measured within 3e-13 of their own GEMM on a probe batch of domain corners / edges"}
            # the same step with no sharing (span 0), and with the sharing tolerance at the parity bar itself
            cwl.set_group_span(0)
            r2 = measure(cwl, args.steps, args.warmup, headline_mode)
            out["span0"] = {"value": rate(cwl, r2, args.steps), "ms_per_step": r2["elapsed"] / args.steps * 1e3,
                            "gemms_per_step": cwl.gemms_per_step, "note": "no sharing: one GEMM per spec (round 2's path)"}
            cwl.set_group_span(1)
            cwl.set_group_tolerance(1e-12)
            if cwl.gemms_per_step < g:
                r2 = measure(cwl, args.steps, args.warmup, headline_mode)
                out["tol_1e-12"] = {"value": rate(cwl, r2, args.steps), "ms_per_step": r2["elapsed"] / args.steps * 1e3,
                                    "gemms_per_step": cwl.gemms_per_step,
                                    "note": "pcx_bary_set_group_tolerance(h, 1e-12): price / vega share too (8.5e-13 from vega's own "
                                            "GEMM at the domain corners: no margin to the parity bar; opt-in)"}
            cwl.set_group_tolerance(3e-13)
            if rank != 0:
                return None
        if cwl.build_info:
            out["build"] = cwl.build_info
        if name == "tt10d" and world == 1 and os.environ.get("PCX_BENCH_TT10D_FULL", "1") != "0":
            try:
                out["full_batch_one_gpu"] = tt10d_full(cwl)
            except Exception as exc:                     # noqa: BLE001
                sys.stderr.write(f"bench.py: tt10d full batch failed: {type(exc).__name__}: {exc}\n")
                out["full_batch_one_gpu"] = {"error": f"{type(exc).__name__}: {exc}"}
        return out, cwl

    def tt10d_full(cwl, shards=8, steps=10, warm=3):
        """BASELINE config 5 whole on ONE GPU: 10^8 points x 10 doubles = 8 GB resident in HBM (288 GB per GPU), evaluated by
        one launch per step -- what DESIGN.md section 2 sizes the device-resident entry points for.  The batch is the 8 shards
        of the config's recipe (shard s: default_rng(99 + s), 12.5 M x 10), uploaded one after the other."""
        per = 12_500_000
        n = per * shards
        d_pts, d_out = ctypes.c_void_p(), ctypes.c_void_p()
        chk(lib.pcx_dev_malloc(dev, n * cwl.d * 8, ctypes.byref(d_pts)))
        chk(lib.pcx_dev_malloc(dev, n * 8, ctypes.byref(d_out)))
        try:
            t0 = time.perf_counter()
            for s_ in range(shards):
                blk = np.ascontiguousarray(uniform_points(cwl.domain, per, 99 + s_))
                chk(lib.pcx_memcpy_h2d(dev, ctypes.c_void_p(d_pts.value + s_ * per * cwl.d * 8),
                                       blk.ctypes.data_as(ctypes.c_void_p), blk.nbytes))
            fill_s = time.perf_counter() - t0
            st = cwl.stream()
            for _ in range(warm):
                cwl.launch(d_pts, n, d_out, st)
            a, b = new_event(), new_event()
            chk(lib.pcx_event_record(a, st))
            for _ in range(steps):
                cwl.launch(d_pts, n, d_out, st)
            chk(lib.pcx_event_record(b, st))
            chk(lib.pcx_stream_synchronize(st))
            ms = elapsed_ms(a, b) / steps
            tail = np.empty(1000)
            chk(lib.pcx_memcpy_d2h(dev, tail.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_out.value + (n - 1000) * 8), 8000))
            if not np.isfinite(tail).all():
                raise SystemExit("non-finite results in the 10^8-point batch")
            # shard 0's first rows must equal what the per-GPU companion batch (the same generator) gives
            head = np.empty(4096)
            chk(lib.pcx_memcpy_d2h(dev, head.ctypes.data_as(ctypes.c_void_p), d_out, head.nbytes))
            same = bool(np.array_equal(head, cwl.model.eval_batch(uniform_points(cwl.domain, per, 99)[:4096])))
            frac = cwl.flop_per_eval * n / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS
            return {"workload": "config 5 whole: 10^8 points x 10 dimensions (8 shards of 12.5 M, default_rng(99 + s)) resident on one GPU, one launch per step",
                    "points_per_step": n, "device_bytes": n * (cwl.d + 1) * 8, "steps": steps, "warmup": warm,
                    "ms_per_step": ms, "value": n / (ms * 1e-3), "unit": "point-evals/s", "roofline_frac": frac,
                    "host_generate_and_upload_s_not_timed": fill_s, "first_rows_equal_the_per_gpu_batch": same}
        finally:
            lib.pcx_dev_free(dev, d_pts)
            lib.pcx_dev_free(dev, d_out)
            free_events()

    # the metric names both interpolants and BASELINE.json lists the Greeks run as a config:
    # the default (barycentric) run also times config 3 (TT) and config 4 (Greeks) with the
    # same discipline and reports them beside the headline value, never mixed into it
    companions = {}
    if args.workload == "bary5d" and not args.no_companion:
        for name, field in (("greeks5d", "greeks"), ("tt5d", "tt"), ("tt10d", "tt10d")):
            try:
                got = companion(name)
            except Exception as exc:                     # noqa: BLE001
                # a companion must not take the headline line with it; with several ranks a failure in the
                # middle of a gather cannot be contained, so there it still ends the run
                if world > 1:
                    raise
                sys.stderr.write(f"bench.py: companion {name} failed: {type(exc).__name__}: {exc}\n")
                companions[field] = {"error": f"{type(exc).__name__}: {exc}"}
                continue
            if got is None:
                continue
            out, cwl = got
            if name == "greeks5d":
                # per-spec rates from separate single-spec launches (events around each launch)
                per = {}
                if world == 1:
                    pts = np.ascontiguousarray(cwl.points(rank))
                    d_pts, d_out = ctypes.c_void_p(), ctypes.c_void_p()
                    chk(lib.pcx_dev_malloc(dev, pts.nbytes, ctypes.byref(d_pts)))
                    chk(lib.pcx_dev_malloc(dev, cwl.points_per_gpu * 6 * 8, ctypes.byref(d_out)))
                    chk(lib.pcx_memcpy_h2d(dev, d_pts, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes))
                    st = cwl.stream()
                    for i, nm in enumerate(GREEK_NAMES):
                        a, b = new_event(), new_event()
                        cwl.launch(d_pts, cwl.points_per_gpu, d_out, st, which=i)
                        chk(lib.pcx_event_record(a, st))
                        for _ in range(5):
                            cwl.launch(d_pts, cwl.points_per_gpu, d_out, st, which=i)
                        chk(lib.pcx_event_record(b, st))
                        per[nm] = cwl.points_per_gpu * 5 / (elapsed_ms(a, b) / 1e3)
                    lib.pcx_dev_free(dev, d_pts)
                    lib.pcx_dev_free(dev, d_out)
                    free_events()
                    out["per_spec_point_evals_per_s"] = per
                    out["specs"] = GREEK_SPECS
            if world == 1 and not args.no_cpu_baseline:
                try:
                    if name == "tt5d":
                        out["cpu_baseline"] = cpu_baseline(cwl, 6.0)
                    out["end_to_end"] = end_to_end(cwl, reps=2)
                except Exception as exc:                 # noqa: BLE001
                    sys.stderr.write(f"bench.py: {name} baseline legs failed: {type(exc).__name__}: {exc}\n")
                    out["baseline_error"] = f"{type(exc).__name__}: {exc}"
            companions[field] = out
        if world == 1:
            try:
                companions["midsize"] = midsize_companion()
            except Exception as exc:                     # noqa: BLE001
                sys.stderr.write(f"bench.py: mid-size shapes companion failed: {type(exc).__name__}: {exc}\n")
                companions["midsize"] = {"error": f"{type(exc).__name__}: {exc}"}
            try:
                companions["c1"] = config1_companion()
            except Exception as exc:                     # noqa: BLE001
                sys.stderr.write(f"bench.py: config 1 companion failed: {type(exc).__name__}: {exc}\n")
                companions["c1"] = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0:
        line = {
            "metric": METRIC,
            "value": rate(wl, head, args.steps),
            "unit": "point-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["elapsed"] / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl.name, "points_per_gpu_per_step": n,
                       "evals_per_point": wl.evals_per_point,
                       "parallelism": f"batch-sharded x{world}, model replicated, one process per GPU"
                                      + {"none": "", "rccl": ", RCCL gather of the result blocks on rank 0 each step "
                                                             "(overlapping the next launch)",
                                         "rccl+d2h": ", RCCL gather + download on rank 0 each step",
                                         "d2h": ", every rank downloads its block into shared pinned host memory "
                                                "each step"}[headline_mode],
                       "gather": headline_mode},
            "roofline": roofline_of(wl, head),
        }
        line["roofline"]["avg_launch_ms_per_rank"] = head["avg_launch_ms_per_rank"]
        if wl.build_info:
            line["config"]["build"] = wl.build_info
        if multi:
            line["gather"] = gathers
            buses = [r_.get("pci_bus_id") for r_ in ranks_on]
            line["comm"] = {"backend": "rccl" if comm is not None else None,
                            "rccl_version": comm.rccl_version if comm is not None else None,
                            "rccl_error": rccl_error, "torch": "torch" in sys.modules,
                            "world": world,
                            "rccl_world_seen_by_every_rank": (sorted({r_.get("comm_world") for r_ in ranks_on}) == [world]
                                                              if comm is not None else None),
                            "distinct_gpus": len({b_ for b_ in buses if b_}),
                            "ranks": ranks_on}
            line["config"]["blocks_verified"] = head.get("blocks_verified")
        line.update(companions)
    if world == 1 and not args.no_cpu_baseline:
        def guarded(what, fn):
            # reported legs: a failure here (say, no C compiler for the oracle on this box) must not cost the line
            try:
                return fn()
            except Exception as exc:                     # noqa: BLE001
                sys.stderr.write(f"bench.py: {what} failed: {type(exc).__name__}: {exc}\n")
                return {"error": f"{type(exc).__name__}: {exc}"}
        e2e = guarded("end_to_end", lambda: end_to_end(wl))
        cpu = guarded("cpu_baseline", lambda: cpu_baseline(wl, 10.0))
        if rank == 0:
            line["end_to_end"] = e2e
            line["cpu_baseline"] = cpu
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())

    lib.pcx_stream_destroy(copy_stream)
    if comm is not None:
        comm.close()
    if group is not None:
        if boot_group is not None and not stuck_init:
            boot_group.close()
        group.close()
    if stuck_init:                            # a thread is still inside librccl: leave without joining it
        if rank == 0 and boot_group is not None:          # ... and its rendezvous directory is nobody's to close
            shutil.rmtree(boot_group.directory, ignore_errors=True)
        sys.stderr.flush()
        os._exit(0)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="bary5d", choices=["bary5d", "greeks5d", "tt5d", "tt10d"])
    ap.add_argument("--points", type=int, default=0, help="query points per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the CPU baseline and the host-pointer end_to_end legs")
    ap.add_argument("--no-companion", action="store_true",
                    help="bary5d only: skip the Greeks (config 4), TT (config 3) and 10-D TT (config 5) companions")
    ap.add_argument("--gather", default="auto", choices=["auto", "none", "rccl", "rccl+d2h", "d2h"],
                    help="what each timed step of the headline does with the result blocks (N > 1)")
    ap.add_argument("--variant", type=int, default=0,
                    help="barycentric kernel: 0 auto, 1 rows, 2 MFMA 16x16x4, 3 MFMA 4x4x4_4b")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_children(args.gpus)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
