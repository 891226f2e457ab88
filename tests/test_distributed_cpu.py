"""World-size-2 run of the batch-sharding path on CPU (gloo): each rank evaluates its own
row block and the blocks are gathered on rank 0.  The per-rank evaluator here is the CPU
oracle (tests may use it); on a GPU node the evaluator is the HIP path and the backend is
nccl (= RCCL) -- the sharding / gather code is the same."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
    import torch.distributed as dist
    import functions as F
    import oracle
    from pychebyshev_amd.distributed import eval_sharded, shard_bounds, gather_results
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = np.load(os.path.join({root!r}, "tests", "golden", "g1_sincos2d.npz"))
    model = oracle.BaryModel([g["nodes0"], g["nodes1"]], [g["weights0"], g["weights1"]],
                             [g["diff0"], g["diff1"]], g["tensor"])
    for n in (1001, 7, 1, 0, 4096):
        pts = np.random.default_rng(3).uniform(-1, 1, (n, 2))
        full = eval_sharded(lambda block: oracle.bary_eval_batch(model, block, [0, 0]) if len(block) else np.empty(0),
                            pts)
        if rank == 0:
            want = oracle.bary_eval_batch(model, pts, [0, 0]) if n else np.empty(0)
            assert full.shape == (n,) and np.array_equal(full, want), n
        else:
            assert full is None
    lo, hi = shard_bounds(1001, rank, world)
    assert (lo, hi) == ((0, 501) if rank == 0 else (501, 1001))
    dist.barrier()
    dist.destroy_process_group()
    sys.stdout.write("rank-%d-ok\\n" % rank)
    sys.stdout.flush()
""")


def test_sharded_eval_and_gather_world_size_2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", env["MASTER_PORT"], str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "rank-0-ok" in res.stdout and "rank-1-ok" in res.stdout
