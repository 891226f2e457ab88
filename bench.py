#!/usr/bin/env python3
"""bench.py -- headline benchmark: 5-D Black-Scholes barycentric point-evals/sec on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N = 1   : single process, no torch -- the C ABI (libpcx_hip.so) does everything.
  N > 1   : launched by torch.distributed.run, one rank per GPU; torch is plumbing only
            (rendezvous, barrier, max-over-ranks, the RCCL gather of the result blocks).

A "step" is one pass of the hot path over one batch already resident in HBM:
  workload bary5d (default, BASELINE.json configs[1]): 5-D Black-Scholes n=11^5 full
  tensor, 10^6 fp64 query points per GPU (seed 99 + rank, column-wise uniform), value
  spec.  Other workloads (--workload greeks5d | tt5d | tt10d) are the parity-test
  configs, runnable for profiling; they are not the headline line.

Weak scaling: every rank owns its own 10^6-point batch; value = all points of all ranks
/ max-over-ranks wall time of the K steps (barrier + device sync on both sides).  With
N > 1 each step ends with the gather of the N result blocks on rank 0 (the path's only
collective, inside the timed region).

Also on the JSON line:
  roofline     dominant kernel (k_bary_mfma): algorithmic flop per launch / mean launch
               duration from HIP events recorded on the launch stream, vs the FP64 MFMA peak.
  cpu_baseline the CPU oracle (C restatement of the reference, OpenMP over all host
               cores) timed on a bounded sample of the same workload, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import functions as F  # noqa: E402  (analytic Black-Scholes + the seed-99 point recipe)
from pychebyshev_amd import ChebyshevApproximation, ChebyshevTT, _lib  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X FP64 matrix peak (= FP64 vector peak), AMD spec;
                                  # v_mfma_f64_16x16x4_f64 at 64 cycles/SIMD x 1024 SIMDs x 2.4 GHz
HBM_PEAK_GBS = 8000.0


def cpu_threads() -> int:
    """Threads for the CPU baseline: the box's per-GPU CPU share (16), or fewer."""
    return max(1, min(os.cpu_count() or 1, int(os.environ.get("PCX_CPU_THREADS", "16"))))


# ----------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------
def bs5d_tensor() -> np.ndarray:
    info = ChebyshevApproximation.nodes(5, F.BS5_DOMAIN, F.BS5_NODES)
    vals = np.array([F.bs_5d(list(p)) for p in info["full_grid"]])
    return vals.reshape(info["shape"])


class Workload:
    name = ""
    d = 0
    points_per_gpu = 0
    evals_per_point = 1          # point-evals per query point per step
    flop_per_eval = 0.0          # algorithmic flop per point-eval
    bytes_per_eval = 0.0         # algorithmic HBM bytes per point-eval
    kernel = ""

    def points(self, rank: int) -> np.ndarray:
        raise NotImplementedError

    def launch(self, d_pts, n, d_out, stream):
        raise NotImplementedError


class Bary5D(Workload):
    name = "5D Black-Scholes n=11^5 full-tensor barycentric, 1M fp64 queries per GPU"
    d = 5
    flop_per_eval = 354310.0     # 177,155 FMA: 161051 + 14641 + 1331 + 121 + 11
    bytes_per_eval = 48.0        # 5 x 8 in + 8 out
    kernel = "k_bary_mfma<31,2,false>"

    def __init__(self, n_points, specs=((0, 0, 0, 0, 0),)):
        self.points_per_gpu = n_points
        self.specs = [list(s) for s in specs]
        self.evals_per_point = len(self.specs)
        self.model = ChebyshevApproximation.from_values(bs5d_tensor(), 5, F.BS5_DOMAIN, F.BS5_NODES)
        self.model.to_device()
        self.m = self.model._model()
        self.spec_arrays = [_lib.i32(s) for s in self.specs]

    def points(self, rank):
        return F.bs5_query_points(self.points_per_gpu, seed=99 + rank)

    def stream(self):
        st = ctypes.c_void_p()
        _lib.check(self.m.lib.pcx_bary_stream(self.m.handle, ctypes.byref(st)), self.m.lib)
        return st

    def launch(self, d_pts, n, d_out, stream):
        for i, s in enumerate(self.spec_arrays):
            out_i = ctypes.c_void_p(d_out.value + i * n * 8)
            _lib.check(self.m.lib.pcx_bary_eval_batch_dev(self.m.handle, d_pts, n, _lib.p_i32(s), out_i, stream),
                       self.m.lib)

    def oracle_rate(self, seconds=12.0):
        import oracle
        om = oracle.BaryModel(self.model.nodes, self.model.weights, self.model.diff_matrices,
                              self.model.tensor_values)
        pts = F.bs5_query_points(self.points_per_gpu, seed=99)
        oracle.set_num_threads(cpu_threads())
        probe = 4000
        oracle.bary_eval_batch(om, pts[:probe], self.specs[0])          # thread start-up
        probe = 20000
        t0 = time.perf_counter()
        oracle.bary_eval_batch(om, pts[:probe], self.specs[0])
        rate = probe / (time.perf_counter() - t0)
        sample = int(min(len(pts), max(probe, rate * seconds / len(self.specs))))
        t0 = time.perf_counter()
        for s in self.specs:
            oracle.bary_eval_batch(om, pts[:sample], s)
        dt = time.perf_counter() - t0
        # the reference's own algorithm shape (Python loop of NumPy matvecs), one core
        npn = 3000
        t0 = time.perf_counter()
        oracle.bary_eval_batch_numpy(om, pts[:npn], self.specs[0])
        self.numpy_loop = {"value": npn / (time.perf_counter() - t0), "unit": "point-evals/s", "cores": 1,
                           "sample": f"first {npn} points, per-point NumPy loop as in the reference"}
        return sample * len(self.specs) / dt, oracle.num_threads(), \
            f"first {sample} of the 10^6 seed-99 points x {len(self.specs)} spec(s), {dt:.1f} s"


class TTWork(Workload):
    def __init__(self, n_points, kind):
        self.points_per_gpu = n_points
        self.build_info = None
        if kind == "tt5d":
            # config 3 = TT-Cross build (max_rank 8, seed 42) + batched queries: the model is built
            # here, through the Python callback and the device-side cross steps, and timed
            g = np.load(os.path.join(ROOT, "tests", "golden", "g4_tt_bs5d.npz"))
            built = ChebyshevTT(F.bs_5d, 5, F.BS5_DOMAIN, F.BS5_NODES, max_rank=8)
            t0 = time.perf_counter()
            built.build(verbose=False, seed=42)
            self.build_info = {"method": "cross", "seconds": time.perf_counter() - t0,
                               "tt_ranks": list(built.tt_ranks), "unique_evals": int(built.total_build_evals),
                               "reference_ranks": [int(v) for v in g["r8_ranks"]],
                               "reference_unique_evals": int(g["r8_evals"])}
            cores = built._coeff_cores
            self.domain = F.BS5_DOMAIN
            self.name = "5D Black-Scholes ChebyshevTT ranks [1,8,8,8,6,1] eval_batch"
            self.flop_per_eval, self.bytes_per_eval = 4560.0, 48.0
            self.kernel = "k_tt_eval_wfirst<8,3,1>"
        else:
            rng = np.random.default_rng(16)
            ranks = [1] + [16] * 9 + [1]
            cores = [rng.standard_normal((ranks[k], 11, ranks[k + 1])) / np.sqrt(ranks[k] * 11) for k in range(10)]
            self.domain = [[-1.0, 1.0]] * 10
            self.name = "10D synthetic rank-16 ChebyshevTT eval_batch"
            self.flop_per_eval, self.bytes_per_eval = 49920.0, 88.0
            self.kernel = "k_tt_eval_mfma<4,1,4>"
        self.d = len(cores)
        self.cores = cores
        self.model = ChebyshevTT.from_coeff_cores(cores, self.domain)
        self.model.to_device()
        self.t = self.model._dev()

    def points(self, rank):
        rng = np.random.default_rng(99 + rank)
        return np.column_stack([rng.uniform(lo, hi, self.points_per_gpu) for lo, hi in self.domain])

    def stream(self):
        st = ctypes.c_void_p()
        _lib.check(self.t.lib.pcx_tt_stream(self.t.handle, ctypes.byref(st)), self.t.lib)
        return st

    def launch(self, d_pts, n, d_out, stream):
        _lib.check(self.t.lib.pcx_tt_eval_batch_dev(self.t.handle, d_pts, n, d_out, stream), self.t.lib)

    def oracle_rate(self, seconds=10.0):
        import oracle
        pts = self.points(0)
        oracle.set_num_threads(cpu_threads())
        probe = min(len(pts), 200_000)
        oracle.tt_eval_batch(self.cores, self.domain, pts[:20000])      # thread start-up
        t0 = time.perf_counter()
        oracle.tt_eval_batch(self.cores, self.domain, pts[:probe])
        rate = probe / (time.perf_counter() - t0)
        sample = int(min(len(pts), max(probe, rate * seconds)))
        t0 = time.perf_counter()
        oracle.tt_eval_batch(self.cores, self.domain, pts[:sample])
        dt = time.perf_counter() - t0
        return sample / dt, oracle.num_threads(), f"first {sample} points of rank 0's batch, {dt:.1f} s"


def make_workload(name, n_points):
    if name == "bary5d":
        return Bary5D(n_points or 1_000_000)
    if name == "greeks5d":
        return Bary5D(n_points or 1_000_000, specs=F.GREEK_SPECS_5D[:6])
    if name in ("tt5d", "tt10d"):
        return TTWork(n_points or (10_000_000 if name == "tt5d" else 4_000_000), name)
    raise SystemExit(f"unknown workload {name}")


# ----------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="bary5d")
    ap.add_argument("--points", type=int, default=0, help="query points per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-companion", action="store_true",
                    help="bary5d only: skip the TT (config 3) measurement reported under \"tt\"")
    ap.add_argument("--variant", type=int, default=0,
                    help="barycentric kernel: 0 auto, 1 rows, 2 MFMA 16x16x4, 3 MFMA 4x4x4_4b")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # The contract is ONE JSON line on stdout.  RCCL/torch print banners to the C-level
    # stdout at init, so keep the real stdout aside and point fd 1 at stderr meanwhile.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    dist = torch = None
    # PCX_BENCH_FORCE_TORCH=1 exercises the multi-GPU plumbing (nccl group, shared stream,
    # gather) with a single rank, so that path can be rehearsed on a one-GPU box.
    if world > 1 or os.environ.get("PCX_BENCH_FORCE_TORCH") == "1":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    os.environ["PCX_DEVICE"] = str(local_rank)
    if world == 1:
        from pychebyshev_amd import _build
        if _build.needs_build():          # fresh checkout: the .so is git-ignored
            _build.build()
    lib = _lib.load()
    dev = local_rank

    def sync():
        if torch is not None:
            torch.cuda.synchronize()
        else:
            _lib.check(lib.pcx_device_synchronize(dev), lib)

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
            sync()

    tstream = None
    if torch is not None:
        # kernel and gather share one (non-default) stream, so the collective is ordered
        # behind the kernel without a host sync; a NULL stream would mean "the handle's own".
        tstream = torch.cuda.Stream()
        torch.cuda.synchronize()
        torch.cuda.set_stream(tstream)

    def measure(wl, steps, warmup):
        """W untimed + K timed steps of one workload, batch resident in HBM beforehand.
        Returns (wall seconds, max over ranks; per-step kernel milliseconds from HIP events)."""
        n = wl.points_per_gpu
        pts = np.ascontiguousarray(wl.points(rank))
        n_out = n * wl.evals_per_point
        if torch is None:
            d_pts, d_out = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.check(lib.pcx_dev_malloc(dev, pts.nbytes, ctypes.byref(d_pts)), lib)
            _lib.check(lib.pcx_dev_malloc(dev, n_out * 8, ctypes.byref(d_out)), lib)
            _lib.check(lib.pcx_memcpy_h2d(dev, d_pts, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes), lib)
            stream = wl.stream()
            gathered = t_out = None
        else:
            t_pts = torch.from_numpy(pts).cuda()
            # two result buffers: the RCCL gather of step i (on the collective's own stream)
            # overlaps the kernel of step i+1, which writes the other buffer
            t_outs = [torch.empty(n_out, dtype=torch.float64, device="cuda") for _ in range(2)]
            t_out = t_outs[0]
            d_pts = ctypes.c_void_p(t_pts.data_ptr())
            d_outs = [ctypes.c_void_p(t.data_ptr()) for t in t_outs]
            stream = ctypes.c_void_p(tstream.cuda_stream)
            gathered = [[torch.empty_like(t_out) for _ in range(world)] if rank == 0 else None
                        for _ in range(2)]
        pending = [None, None]
        count = [0]

        def step(events=None):
            slot = count[0] & 1
            count[0] += 1
            if dist is not None and pending[slot] is not None:
                pending[slot].wait()          # the launch stream waits for the gather that read this buffer
                pending[slot] = None
            out_ptr = d_out if torch is None else d_outs[slot]
            if events is not None:
                _lib.check(lib.pcx_event_record(events[0], stream), lib)
            wl.launch(d_pts, n, out_ptr, stream)
            if events is not None:
                _lib.check(lib.pcx_event_record(events[1], stream), lib)
            if dist is not None:
                # ordered behind the kernel (same current stream), not blocking the next launch
                pending[slot] = dist.gather(t_outs[slot], gathered[slot], dst=0, async_op=True)

        def drain():
            for i in (0, 1):
                if pending[i] is not None:
                    pending[i].wait()
                    pending[i] = None

        evs = []
        for _ in range(steps):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.check(lib.pcx_event_create(dev, ctypes.byref(a)), lib)
            _lib.check(lib.pcx_event_create(dev, ctypes.byref(b)), lib)
            evs.append((a, b))
        for _ in range(warmup):
            step()
        drain()
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(evs[i])
        drain()                               # every gather of the K timed steps has completed
        barrier()
        elapsed = time.perf_counter() - t0

        kernel_ms = []
        for a, b in evs:
            ms = ctypes.c_float()
            _lib.check(lib.pcx_event_elapsed_ms(a, b, ctypes.byref(ms)), lib)
            kernel_ms.append(ms.value)
            lib.pcx_event_destroy(a)
            lib.pcx_event_destroy(b)
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # sanity: the last step's results are finite
        if torch is None:
            got = np.empty(n_out)
            _lib.check(lib.pcx_memcpy_d2h(dev, got.ctypes.data_as(ctypes.c_void_p), d_out, n_out * 8), lib)
            lib.pcx_dev_free(dev, d_pts)
            lib.pcx_dev_free(dev, d_out)
        else:
            got = t_outs[(count[0] - 1) & 1].cpu().numpy()
        if not np.isfinite(got).all():
            raise SystemExit(f"non-finite results in the benchmark batch ({wl.name})")
        return elapsed, kernel_ms

    def roofline_of(wl, kernel_ms, workload_key):
        n = wl.points_per_gpu
        launches = wl.evals_per_point                     # kernel launches between the two events
        avg_launch_s = float(np.mean(kernel_ms)) / 1e3 / launches
        flop_per_launch = wl.flop_per_eval * n
        achieved = flop_per_launch / avg_launch_s / 1e12
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            try:
                rec = json.load(open(pmc_path)).get(workload_key)
                if rec and rec.get("points") == n:
                    traffic = rec["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        return {"bound": "mfma", "kernel": wl.kernel, "achieved": achieved,
                "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                "avg_launch_ms": avg_launch_s * 1e3,
                "algorithmic_flop_per_launch": flop_per_launch,
                "algorithmic_hbm_bytes_per_launch": wl.bytes_per_eval * n,
                "hbm_frac": wl.bytes_per_eval * n / avg_launch_s / 1e9 / HBM_PEAK_GBS,
                "traffic": traffic}

    wl = make_workload(args.workload, args.points)
    if args.variant and hasattr(wl, "m"):
        _lib.check(wl.m.lib.pcx_bary_set_kernel(wl.m.handle, args.variant), wl.m.lib)
    n = wl.points_per_gpu
    elapsed, kernel_ms = measure(wl, args.steps, args.warmup)

    # the metric names both interpolants: the default (barycentric) run also times the TT
    # half of it -- config 3's eval_batch, 10^7 points per GPU -- and reports it beside the
    # headline value (same timing discipline, same JSON line, never mixed into `value`)
    companion = None
    if args.workload == "bary5d" and not args.no_companion:
        cwl = make_workload("tt5d", 0)
        c_elapsed, c_ms = measure(cwl, args.steps, args.warmup)
        if rank == 0:
            companion = {"workload": cwl.name, "build": cwl.build_info, "points_per_gpu_per_step": cwl.points_per_gpu,
                         "value": float(cwl.points_per_gpu) * world * args.steps / c_elapsed,
                         "unit": "point-evals/s", "ms_per_step": c_elapsed / args.steps * 1e3,
                         "roofline": roofline_of(cwl, c_ms, "tt5d")}

    if rank == 0:
        total_evals = float(n) * wl.evals_per_point * world * args.steps
        value = total_evals / elapsed
        line = {
            "metric": "point-evals/sec, 5D Black-Scholes n=11^5 barycentric + TT, 1/2/4/8 GPU",
            "value": value,
            "unit": "point-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl.name, "points_per_gpu_per_step": n,
                       "evals_per_point": wl.evals_per_point,
                       "parallelism": f"batch-sharded x{world}, model replicated"
                                      + (", RCCL gather of results each step (overlapping the next launch)" if dist is not None else "")},
            "roofline": roofline_of(wl, kernel_ms, args.workload),
        }
        if getattr(wl, "build_info", None):
            line["config"]["build"] = wl.build_info
        if companion is not None:
            line["tt"] = companion
        if world == 1 and not args.no_cpu_baseline:
            rate, cores, sample = wl.oracle_rate()
            line["cpu_baseline"] = {"value": rate, "unit": "point-evals/s", "cores": cores,
                                    "kind": "port", "sample": sample,
                                    "host_cpus": os.cpu_count()}
            if getattr(wl, "numpy_loop", None):
                line["cpu_baseline"]["numpy_loop"] = wl.numpy_loop
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
