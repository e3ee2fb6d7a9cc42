"""N>1 path on CPU, world_size-2 gloo runs:
  * the sharding + all-gather helper (`distributed.solve_sharded`) with an injected solver: the sharded result must equal
    the single-process result bitwise;
  * bench.py's own schedule (`batch_stream.BatchStream`: several handles in flight, retire + all_gather_into_tensor of the
    solution blocks) with injected asynchronous handles;
  * `bench.py --gpus N` refuses to print a line when fewer than N GPUs are visible.
The oracle stands in for the HIP backend here -- the helpers themselves are backend-agnostic."""
import os
import subprocess
import sys
import threading

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


def _run_world(target, args, world=2, timeout=300):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=timeout)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_sharded_equals_single_process(free_port):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from boundplanner_amd import scenes
    from boundplanner_amd.distributed import shard_bounds
    assert [shard_bounds(7, 2, r) for r in range(2)] == [(0, 4), (4, 7)]
    assert [shard_bounds(8, 4, r) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    N, B = 6, 7
    got = _run_world(_worker, (free_port, N, B))
    batch = scenes.make_batch(B, N, 99, O.fk_batch, randomize_sets=True)
    ref = O.solve_batch(N, batch["x0"], batch["lbx"], batch["ubx"], batch["p"], nthreads=1)
    for k in ("x", "f", "iters", "status", "viol"):
        assert np.array_equal(got[k], ref[k]), k


class _FakeHandle:
    """Asynchronous stand-in for a solver handle: x = x0 + rowsum(p) * (1 + rank) computed by a worker thread."""

    def __init__(self, rank):
        self.rank, self.t, self.n_active, self.log = rank, None, 0, []

    def solve_async(self, n, d, o):
        assert self.t is None, "one solve in flight per handle"
        self.n_active = n
        self.log.append(n)

        def work():
            o["x"][:n] = d["x0"] + d["p"].sum(dim=1, keepdim=True) * (1 + self.rank)
            self.n_active = 0
        self.t = threading.Thread(target=work)
        self.t.start()

    def wait(self):
        self.t.join()
        self.t = None

    def active(self):
        return self.n_active

    def last_kernel_ms(self):
        return 1.0


def _stream_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from boundplanner_amd.batch_stream import BatchStream
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, n_w, M, depth, nd = 5, 12, 2, 3, 4
    g = torch.Generator().manual_seed(100 + rank)
    inputs = {"x0": torch.rand((nd * B, n_w), generator=g, dtype=torch.float64),
              "p": torch.rand((nd * B, 7), generator=g, dtype=torch.float64)}
    outs = [{"x": torch.zeros((M * B, n_w), dtype=torch.float64)} for _ in range(depth)]
    gathered = torch.zeros((world * M * B, n_w), dtype=torch.float64)
    seen = []

    class Rec(BatchStream):
        def retire(self, j):
            m = self.busy[j]
            super().retire(j)
            if m:
                seen.append((m, self.gathered[:self.world * m * self.B].clone()))
    hs = [_FakeHandle(rank) for _ in range(depth)]
    st = Rec(hs, outs, inputs, B, merge=M, gate=1.0, dist=dist, gathered=gathered)
    st.run(5)          # calls of 2, 2, 1 batches
    st.run(2)          # continues round-robin: one more call of 2 batches
    all_inputs = [None] * world
    dist.all_gather_object(all_inputs, {k: v.numpy() for k, v in inputs.items()})
    if rank == 0:
        q.put(dict(seen=[(m, g_.numpy()) for m, g_ in seen], inputs=all_inputs, calls=st.calls, kms=len(st.kernel_ms),
                   log=[h.log for h in hs]))
    dist.barrier()
    dist.destroy_process_group()


def test_batch_stream_gathers_every_call_gloo(free_port):
    """bench.py's schedule: every retired call delivers, on every rank, the solution blocks of all ranks in rank order."""
    import torch
    world, B, M, nd = 2, 5, 2, 4
    got = _run_world(_stream_worker, (free_port,), world=world)
    assert got["calls"] == 4 and got["kms"] == 4
    assert got["log"] == [[10, 10], [10], [5]]                 # handle 0 took calls 0 and 3, handle 1 call 1, handle 2 call 2
    sizes = [m for m, _ in got["seen"]]
    assert sorted(sizes) == [1, 2, 2, 2]
    # retire order: run(5) drains calls 0, 1, 2 in order; run(2) then retires call 3
    slots = [0, 2, 0, 2]                                       # (call * M) % nd: calls 0..3 start at batch slot 0, 2, 0, 2
    for c, (m, g) in enumerate(got["seen"]):
        s = slots[c]
        for r in range(world):
            x0 = got["inputs"][r]["x0"][s * B:(s + m) * B]
            p = got["inputs"][r]["p"][s * B:(s + m) * B]
            want = (torch.from_numpy(x0) + torch.from_numpy(p).sum(dim=1, keepdim=True) * (1 + r)).numpy()   # the handle's arithmetic
            assert np.array_equal(g[r * m * B:(r + 1) * m * B], want), (c, r)


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` on a box with fewer than 2 GPUs exits non-zero and prints no JSON line."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0
    assert "n_gpus" not in r.stdout and "refusing" in r.stderr
    # a launcher that provides a different world size than --gpus is refused too
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "n_gpus" not in r.stdout
