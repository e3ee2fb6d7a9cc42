"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference):
    python tests/golden/gen/gen_golden.py [kin] [nlp6] [nlp10] [nlp15] [nlp20] [nlp30]

* kin.npz      : outputs of the reference's serialized CasADi kinematics (RobotModel/*.ca)
                 at 32 joint configurations (+ d(J dq)/dq by complex step through jacobian.ca).
* nlp_N*.npz   : f, g, grad f, J_g (full, N=6) or directional derivatives (N=10, 20) of the NLP
                 built by the reference's casadi_ocp_formulation.setup_optimization_problem,
                 evaluated numerically through the casadi shim (complex step, h=1e-30).
The fixtures are data (inputs + expected outputs); no reference source is copied.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.abspath(os.path.join(HERE, ".."))
sys.path.insert(0, HERE)

import ref_eval as RE  # noqa: E402
from ca_tape import load_all  # noqa: E402


def gen_kin():
    tapes = load_all()
    rng = np.random.default_rng(7)
    qs = [np.zeros(7), np.array([0, 0, 0, -np.pi / 2, 0, np.pi / 2, 0.0])]
    lo = np.array([-2.967, -2.094, -2.967, -2.094, -2.967, -2.094, -3.054])
    while len(qs) < 32:
        qs.append(rng.uniform(lo, -lo))
    qs = np.array(qs)
    dqs = rng.normal(size=qs.shape)
    out = {"q": qs, "dq": dqs}
    out["fk_pos"] = np.array([tapes["fk_pos"](q).ravel() for q in qs])
    out["hom_trans"] = np.array([tapes["hom_trans"](q) for q in qs])
    out["jacobian"] = np.array([tapes["jacobian"](q) for q in qs])
    out["fk_pos_col"] = np.array(
        [[tapes[f"fk_pos_col_{i}"](q).ravel() for i in range(6)] for q in qs]
    )
    G = np.zeros((len(qs), 6, 7))
    for n, (q, dq) in enumerate(zip(qs, dqs)):
        for i in range(7):
            qc = q.astype(complex)
            qc[i] += 1e-30j
            G[n, :, i] = (tapes["jacobian"](qc) @ dq).imag / 1e-30
    out["dvdq"] = G
    np.savez_compressed(os.path.join(OUT, "kin.npz"), **out)
    print("kin.npz written")


SPLITS = lambda N: [[0, N, N, N, N], [0, 3, N, N, N], [0, 2, 5, N, N], [0, 1, 2, N, N], [0, 4, N, N, N]]


def random_point(N, rng, variant):
    w = rng.normal(size=RE.n_w(N)) * 0.3
    p = rng.normal(size=RE.N_P) * 0.3
    p[0:5] = SPLITS(N)[variant % 5]
    p[220:231] = np.abs(p[220:231]) + 0.01          # weights > 0
    p[231] = 0.2 + 0.5 * rng.random()               # phi_max
    return w, p


def gen_nlp_full(N, npts, seed):
    rng = np.random.default_rng(seed)
    W, P, F, G, GR, JAC = [], [], [], [], [], []
    for i in range(npts):
        t0 = time.time()
        w, p = random_point(N, rng, i)
        f, g, lbg, ubg = RE.eval_fg(N, w, p)
        gr, jac = RE.eval_derivs(N, w, p)
        W.append(w); P.append(p); F.append(f); G.append(g); GR.append(gr); JAC.append(jac)
        print(f"N={N} point {i} done in {time.time()-t0:.1f}s", flush=True)
    np.savez_compressed(
        os.path.join(OUT, f"nlp_N{N}.npz"),
        N=N, w=np.array(W), p=np.array(P), f=np.array(F), g=np.array(G),
        grad_f=np.array(GR), jac_g=np.array(JAC), lbg=lbg, ubg=ubg,
    )
    print(f"nlp_N{N}.npz written")


def gen_nlp_dir(N, npts, ndir, seed):
    rng = np.random.default_rng(seed)
    W, P, F, G, R, DF, DG = [], [], [], [], [], [], []
    for i in range(npts):
        w, p = random_point(N, rng, i)
        f, g, lbg, ubg = RE.eval_fg(N, w, p)
        rs, dfs, dgs = [], [], []
        for _ in range(ndir):
            r = rng.normal(size=RE.n_w(N))
            df, dg = RE.eval_dir(N, w, p, r)
            rs.append(r); dfs.append(df); dgs.append(dg)
        W.append(w); P.append(p); F.append(f); G.append(g)
        R.append(rs); DF.append(dfs); DG.append(dgs)
        print(f"N={N} point {i} done", flush=True)
    np.savez_compressed(
        os.path.join(OUT, f"nlp_N{N}.npz"),
        N=N, w=np.array(W), p=np.array(P), f=np.array(F), g=np.array(G),
        r=np.array(R), df=np.array(DF), dg=np.array(DG), lbg=lbg, ubg=ubg,
    )
    print(f"nlp_N{N}.npz written")


if __name__ == "__main__":
    what = sys.argv[1:] or ["kin", "nlp6", "nlp10", "nlp15", "nlp20", "nlp30"]
    if "kin" in what:
        gen_kin()
    if "nlp6" in what:
        gen_nlp_full(6, 5, 6)
    if "nlp10" in what:
        gen_nlp_dir(10, 8, 3, 10)
    if "nlp15" in what:
        gen_nlp_dir(15, 6, 2, 15)        # the reference's default horizon (util_functions.py:49)
    if "nlp20" in what:
        gen_nlp_dir(20, 8, 3, 20)
    if "nlp30" in what:
        gen_nlp_dir(30, 4, 2, 30)        # BASELINE configs[4]
