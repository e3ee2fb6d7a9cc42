"""N>1 path on CPU: world_size-2 gloo run of the sharding + all-gather helper.  The solver is
injected (the oracle stands in for the HIP backend here -- the helper itself is backend-agnostic);
the sharded result must equal the single-process result bitwise."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, N, B, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as O
    from boundplanner_amd import scenes
    from boundplanner_amd.distributed import solve_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = scenes.make_batch(B, N, 99, O.fk_batch, randomize_sets=True)
    fn = lambda x0, lbx, ubx, p: O.solve_batch(N, x0, lbx, ubx, p, nthreads=1)
    r = solve_sharded(fn, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], dist=dist)
    if rank == 0:
        q.put({k: np.asarray(v) for k, v in r.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_equals_single_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from boundplanner_amd import scenes
    from boundplanner_amd.distributed import shard_bounds
    assert [shard_bounds(7, 2, r) for r in range(2)] == [(0, 4), (4, 7)]
    assert [shard_bounds(8, 4, r) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    N, B = 6, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, 29531, N, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    batch = scenes.make_batch(B, N, 99, O.fk_batch, randomize_sets=True)
    ref = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], nthreads=1)
    for k in ("x", "f", "iters", "status", "viol"):
        assert np.array_equal(got[k], ref[k]), k
