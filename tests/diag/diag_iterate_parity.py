"""Diagnostic (not a test): the iterate-for-iterate table of tests/iterate_parity_lib.py for one workload, and both sides' decision
records of the instances that are furthest apart while in step.   python tests/diag/diag_iterate_parity.py [N B seed rnd K]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401
import iterate_parity_lib as IP, oracle_lib as O
from boundplanner_amd import scenes
from boundplanner_amd.solver import HipBoundMPC
N, B, seed, rnd, K = (int(x) for x in (sys.argv[1:6] + ["20", "1024", "8192", "1", "12"][len(sys.argv) - 1:]))
bes = {}
get = lambda k: bes.setdefault(k, HipBoundMPC(N, max_iter=k))
full = scenes.make_batch(max(B, 8192 if seed == 8192 else B), N, seed, get(1).fk, randomize_sets=bool(rnd))
batch = {k: v[:B] for k, v in full.items() if hasattr(v, "shape") and v.shape[:1] == full["x0"].shape[:1]}
ks = list(range(1, K + 1))
rows, left_at, reason = IP.table(N, batch, get, O, ks)
for r in rows:
    print(r)
a = (batch["x0"], batch["lbx"], batch["ubx"], batch["p"])
np.set_printoptions(linewidth=250, precision=6)
for i in rows[-2]["worst_in_step"]:
    print("instance", i, "left_at", left_at[i], reason[i])
    for k in ks:
        h = get(k).solve_batch(*(v[i:i + 1] for v in a)); ih = get(k).inst_state(1)[0]
        o = O.solve_batch_info(N, *(v[i:i + 1] for v in a), max_iter=k)
        print(k, "rel %.2e" % IP.rel_diff(N, h["x"], o["x"])[0], "hip", ih[1:], "\n" + " " * 14 + "or ", o["info"][0][1:])
