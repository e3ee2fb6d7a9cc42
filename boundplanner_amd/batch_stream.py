"""A stream of independent batches through a few solver handles, with the all-gather of the solutions.

This is the schedule `bench.py` runs (BASELINE.json configs[2]/[3]): the instances of a batch are
independent, so every rank (one process per GPU) solves its own contiguous shard and there is NO collective on the
data path; the only exchange is the all-gather of the solution block `x` that `north_star` asks for (RCCL over xGMI on
GPUs, gloo in the CPU tests).  Several handles are kept in flight so that the launch-bound straggler tail of one solver
call overlaps the bulk of the next.

The handles are duck-typed (`solve_async(n, inputs, outputs)`, `wait()`, `active()`, `last_kernel_ms()`), so the same
code runs with the HIP backend (`HipHandle`) and with an injected solver in `tests/test_distributed.py`.
"""
import time


class HipHandle:
    """Adapter of one `HipBoundMPC` handle: device tensors in, device tensors out, asynchronous on the handle's stream."""

    def __init__(self, backend):
        self.be = backend

    def solve_async(self, n, d, o):
        self.be.solve_dev_async(n, d["x0"].data_ptr(), d["lbx"].data_ptr(), d["ubx"].data_ptr(), d["p"].data_ptr(),
                                o["x"].data_ptr(), o["f"].data_ptr(), o["iters"].data_ptr(), o["status"].data_ptr(),
                                o["viol"].data_ptr())

    def wait(self):
        self.be.wait()

    def active(self):
        return self.be.active()

    def last_kernel_ms(self):
        return self.be.last_kernel_ms()


class BatchStream:
    """`handles`: solver handles used round-robin; `outs[j]`: output tensors of handle j (x [M*B, n_w], f, viol, iters,
    status [M*B]); `inputs`: tensors [nd*B, .] holding `nd` distinct batches back to back (nd a multiple of `merge`);
    `gathered`: [world*M*B, n_w] or None; `dist`: initialised torch.distributed module or None;
    `sync_gather`: callable that makes the gather's read of `outs[j]["x"]` complete (the next solve on the handle
    overwrites it)."""

    def __init__(self, handles, outs, inputs, B, merge=1, gate=1.0, dist=None, gathered=None, sync_gather=None):
        self.h, self.outs, self.inputs, self.B = handles, outs, inputs, B
        self.depth, self.M, self.gate = len(handles), merge, gate
        self.dist, self.gathered, self.sync_gather = dist, gathered, sync_gather
        self.world = dist.get_world_size() if dist is not None else 1
        self.nd = inputs["x0"].shape[0] // B
        assert self.nd % merge == 0 or self.nd == 1, "distinct batches must fill whole solver calls"
        self.busy = [0] * self.depth          # batches of the solve in flight on each handle
        self.kernel_ms = []                   # HIP-event time of each retired call / its batches
        self.calls = 0

    def retire(self, j):
        """Wait for the solve in flight on handle j; all-gather its solutions."""
        if not self.busy[j]:
            return
        self.h[j].wait()
        m, self.busy[j] = self.busy[j], 0
        self.kernel_ms.append(self.h[j].last_kernel_ms() / m)
        if self.world > 1:
            n = m * self.B
            self.dist.all_gather_into_tensor(self.gathered[:self.world * n], self.outs[j]["x"][:n])
            if self.sync_gather is not None:
                self.sync_gather()

    def run(self, nbatches):
        """`nbatches` batches, `merge` per solver call (the last call takes what is left)."""
        left = nbatches
        while left > 0:
            m = min(self.M, left)
            j = self.calls % self.depth
            self.retire(j)
            # start the next call when the solves in flight have left their bulk phase (most of their instances
            # finished): the launch-latency-bound tail of one solve then runs beside the throughput-bound bulk of the next
            while any(self.busy[q] and self.h[q].active() > self.gate * self.busy[q] * self.B for q in range(self.depth)):
                time.sleep(0.0005)
            s = (self.calls * self.M) % self.nd if self.nd > 1 else 0
            d = {k: v[s * self.B:(s + m) * self.B] for k, v in self.inputs.items()}
            self.h[j].solve_async(m * self.B, d, self.outs[j])
            self.busy[j] = m
            self.calls += 1
            left -= m
        self.drain()

    def drain(self):
        for q in range(self.depth):
            self.retire((self.calls + q) % self.depth)


class HostBatchStream:
    """The same schedule with the transfers inside ("incl. H2D of inputs and D2H of x, iters, status"): every
    solver call's x0, lbx, ubx, p come from PINNED host memory by asynchronous copies, its x, f, iters, status, viol go back to
    pinned host memory when the call has retired.  Uploads and downloads have a stream each and every handle two sets of staging
    buffers: the inputs of its NEXT call are fetched while the current one is being solved, and the outputs of the previous call
    leave while the next one runs (until round 4's last session the copies sat on the handle's own stream, between its calls).  torch provides the pinned buffers, the copy streams and events and wraps the handles' streams
    (plumbing only)."""

    def __init__(self, backends, host_inputs, B, merge, dev):
        import torch
        self.torch, self.be, self.B, self.M, self.dev = torch, backends, B, merge, dev
        self.depth = len(backends)
        self.src = host_inputs                               # pinned [nd * B, .] tensors
        self.nd = host_inputs["x0"].shape[0] // B
        n = merge * B
        n_w, n_p = host_inputs["x0"].shape[1], host_inputs["p"].shape[1]
        f64, i32 = torch.float64, torch.int32
        self.streams = [torch.cuda.ExternalStream(b.stream(), device=dev) for b in backends]
        # one stream for all uploads, one for all downloads: copies are served in the order the calls need them (with a copy stream per
        # handle the first calls of a run all waited for the sum of their uploads)
        self.h2d, self.d2h = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        self.d_in = [[{k: torch.empty((n, n_p if k == "p" else n_w), dtype=f64, device=dev) for k in ("x0", "lbx", "ubx", "p")} for _ in range(2)]
                     for _ in backends]
        self.d_out = [[dict(x=torch.empty((n, n_w), dtype=f64, device=dev), f=torch.empty(n, dtype=f64, device=dev), viol=torch.empty(n, dtype=f64, device=dev),
                            iters=torch.empty(n, dtype=i32, device=dev), status=torch.empty(n, dtype=i32, device=dev)) for _ in range(2)] for _ in backends]
        self.h_out = [{k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in o[0].items()} for o in self.d_out]
        self.ev_in = [[torch.cuda.Event() for _ in range(2)] for _ in backends]       # inputs of the set have arrived
        self.ev_out = [[None, None] for _ in backends]                               # outputs of the set have left (None: never used)
        self.fetched = [[None, None] for _ in backends]                              # global call index whose inputs the set holds
        self.busy = [0] * self.depth                         # batches of the call in flight on the handle
        self.par = [0] * self.depth                          # staging set of the call in flight / of the next call
        self.calls = 0

    def _fetch(self, j, p, g, m):
        """inputs of global call g (m batches) into staging set p of handle j, on the handle's copy stream"""
        s = (g * self.M) % self.nd
        n = m * self.B
        with self.torch.cuda.stream(self.h2d):
            for k, v in self.d_in[j][p].items():
                v[:n].copy_(self.src[k][s * self.B:s * self.B + n], non_blocking=True)
            self.ev_in[j][p].record()
        self.fetched[j][p] = g

    def _retire(self, j):
        if not self.busy[j]:
            return
        self.be[j].wait()                                    # every kernel of the call has been enqueued and has finished
        n = self.busy[j] * self.B
        p = self.par[j]
        with self.torch.cuda.stream(self.d2h):
            for k, v in self.d_out[j][p].items():
                self.h_out[j][k][:n].copy_(v[:n], non_blocking=True)
            ev = self.torch.cuda.Event()
            ev.record()
        self.ev_out[j][p] = ev
        self.busy[j] = 0
        self.par[j] = 1 - p

    def run(self, nbatches):
        ncalls = (nbatches + self.M - 1) // self.M
        g0 = self.calls
        for c in range(ncalls):
            g = g0 + c
            m = min(self.M, nbatches - c * self.M)
            j = g % self.depth
            self._retire(j)
            p = self.par[j]
            if self.fetched[j][p] != g:
                self._fetch(j, p, g, m)                      # (the first calls of a run: nothing was fetched ahead)
            st = self.streams[j]
            st.wait_event(self.ev_in[j][p])                  # enqueued before the handle's worker thread launches anything
            if self.ev_out[j][p] is not None:
                st.wait_event(self.ev_out[j][p])             # the outputs this set held two calls ago have left
            i, o = self.d_in[j][p], self.d_out[j][p]
            n = m * self.B
            self.be[j].solve_dev_async(n, i["x0"].data_ptr(), i["lbx"].data_ptr(), i["ubx"].data_ptr(), i["p"].data_ptr(), o["x"].data_ptr(),
                                       o["f"].data_ptr(), o["iters"].data_ptr(), o["status"].data_ptr(), o["viol"].data_ptr())
            self.busy[j] = m
            self.calls += 1
            # this handle's next call of the run: its inputs travel while this one is being solved (the other staging set: its
            # last reader, the handle's previous call, has retired)
            c2 = c + self.depth
            if c2 < ncalls:
                self._fetch(j, 1 - p, g0 + c2, min(self.M, nbatches - c2 * self.M))
        for q in range(self.depth):
            self._retire((self.calls + q) % self.depth)
        for st in self.streams + [self.h2d, self.d2h]:
            st.synchronize()
