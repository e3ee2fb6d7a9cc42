"""Plan phase of the plan-then-track API: a path of convex collision-free sets from the start to the goal,
via points inside their intersections, and the orientation schedule along it.

Mirrors the reference's BoundPlanner (/root/reference/bound_planner/BoundPlanner/BoundPlanner.py): constructor arguments and
attributes (:26-127), add_obstacle_reps (:134-152), plan_convex_set_path (:174-584), compute_via_points (:586-743),
check_intersection (:745-772), set_intersection (:774-787), add_edges (:789-896).  The graph of sets and the graph of
set intersections are plain Python structures here; the sub-problems the reference hands to qpOASES / Clarabel / cdd / IPOPT
are solved by planner_opt.py.  Outputs -- (p_via, r_via, bp1_list, sets_via) -- are what BoundMPC.update / MPCNode.
update_reference take (boundplanner_with_mpc_example.py:102-135).  Host code: it runs once per (re)plan.

Behaviour of the reference that callers rely on, kept (each cited where it happens): the stored intersection sets shrink by
1 mm every time via points are computed from them; `fixed_mid` is always true for sampled sets; the replanning start set is
grown along the MPC horizon.
"""
import copy
import heapq

import numpy as np
from scipy.optimize import linprog
from scipy.spatial.transform import Rotation as R

from . import planner_opt as PO
from .convex_set_finder import ConvexSetFinder
from .params import normalize_set_size


def _gram_schmidt(v, b):
    return b - (v @ b) * v


def _shortest_path(adj, src, dst):
    """Dijkstra on {node: {neighbour: weight}}; ties are broken by insertion order of the heap (node ids)."""
    dist, prev, seen = {src: 0.0}, {}, set()
    heap = [(0.0, src)]
    while heap:
        d, u = heapq.heappop(heap)
        if u in seen:
            continue
        seen.add(u)
        if u == dst:
            break
        for v, w in adj.get(u, {}).items():
            nd = d + w
            if v not in dist or nd < dist[v]:
                dist[v], prev[v] = nd, u
                heapq.heappush(heap, (nd, v))
    if dst not in seen:
        raise RuntimeError("(PosPath) no path between the start and the end set")
    path = [dst]
    while path[-1] != src:
        path.append(prev[path[-1]])
    return path[::-1]


class BoundPlanner:
    def __init__(self, obstacles=(), e_p_max=0.5, obs_size_increase=0.08, workspace_max=(1.0, 1.0, 1.2),
                 workspace_min=(-1.0, -1.0, 0.0), seed=None):
        self.replanning = False
        self.sets_via_prev = []
        self.obs_size_increase = obs_size_increase
        self.w_size, self.c_fit, self.w_bias = 0.1, 1.0, 0.01
        self.rng = np.random.default_rng(seed)
        self.max_set_size = 20
        self.workspace_max, self.workspace_min = list(workspace_max), list(workspace_min)
        self.length_ee = 0.05
        self.max_iters, self.nr_optimized, self.nr_free_mid, self.max_samples = 20, 10, 5, 500
        self.e_p_max = e_p_max
        self.obs, self.obs_sets, self.obs_sets_orig, self.obs_points_sets = [], [], [], []
        self.obs_points = np.empty((0, 3))
        self.add_obstacle_reps(obstacles)
        self.set_finder = ConvexSetFinder(self.obs_sets, self.obs_points_sets, self.workspace_max, self.workspace_min)
        self.verbose = False

    def _log(self, msg):
        if self.verbose:
            print(msg)

    @staticmethod
    def make_box(lb, ub):
        return [np.concatenate((np.eye(3), -np.eye(3))), np.concatenate((np.asarray(ub, float), -np.asarray(lb, float)))]

    def add_obstacle_reps(self, obstacles, update=False, reset=False):
        """Boxes [xmin, ymin, zmin, xmax, ymax, zmax] -> halfspace sets inflated by obs_size_increase + their vertices."""
        if reset:
            self.obs, self.obs_sets, self.obs_sets_orig, self.obs_points_sets = [], [], [], []
            self.obs_points = np.empty((0, 3))
        for ob in obstacles:
            ob = np.asarray(ob, float)
            box = self.make_box(ob[:3], ob[3:])
            grown = [box[0].copy(), box[1] + self.obs_size_increase]
            pts = PO.polytope_vertices(grown[0], grown[1])
            self.obs_sets_orig.append(box)
            self.obs_points = np.concatenate((self.obs_points, pts))
            self.obs_points_sets.append(pts)
            self.obs_sets.append(grown)
        self.obs_sets = normalize_set_size(self.obs_sets)
        if update:
            self.set_finder.obs_sets = [[np.asarray(a, float), np.asarray(b, float)] for a, b in self.obs_sets]
            self.set_finder.obs_points_sets = list(self.obs_points_sets)

    # ---------------------------------------------------------------------------------------------------- graph pieces
    def set_intersection(self, set1, set2, tol=0.0):
        inter = [np.concatenate((set1[0], set2[0])), np.concatenate((set1[1], set2[1]))]
        res = linprog(np.zeros(3), A_ub=inter[0], b_ub=inter[1] - tol, bounds=(None, None))
        return res.x, inter, bool(res.success)

    def check_intersection(self, a_set, b_set, l_ee, sample):
        """Does the end-effector offset fit into the set at some orientation along the rotation (20 samples)?"""
        b_shrunk = b_set - 0.001
        for i in range(20):
            om = i / 19
            if PO.fits(a_set, b_shrunk, PO.rodrigues(self.omega_normed, self.omega_norm * om) @ l_ee):
                return True, np.concatenate((sample, [om]))
        return False, np.concatenate((sample, [0]))

    def _new_vertex(self, G, cset, q_ellipse, p_mid, name):
        vid = len(G["v"])
        G["v"].append(dict(cset=cset, name=name, size=1.0 / np.linalg.det(q_ellipse), q_ellipse=q_ellipse, p_mid=p_mid,
                           a_set=np.array(cset[0]), b_set=np.array(cset[1])))
        return vid

    def _new_inter(self, G, **kw):
        iid = len(G["i"])
        G["i"].append(dict(kw))
        G["adj"][iid] = {}
        return iid

    def add_edges(self, id_new, G, end, start):
        """Intersections of the new set with every known set become nodes of the intersection graph; two intersection nodes
        that share a set are joined by an edge weighted with the distance between their projection points."""
        connected = False
        set_new = G["v"][id_new]["cset"]
        for idc, vert in enumerate(G["v"]):
            if idc == id_new:
                continue
            p_int, set_inter, hit = self.set_intersection(vert["cset"], set_new, tol=0.01)
            if not hit:
                continue
            fit, via = self.check_intersection(set_inter[0], set_inter[1], self.l_ee, p_int)
            inter_id = self._new_inter(G, cset=set_inter, name=f"Interset {len(G['i'])}", id0=idc, id1=id_new, set0=vert["cset"],
                                       set1=set_new, conn_to_start=False, conn_to_end=False, p_proj=None, p_via=via)
            me = G["i"][inter_id]
            self.nr_inter_set += 2
            for eid, edge in enumerate(G["i"]):
                c1 = edge["id0"] == idc or edge["id1"] == idc
                c2 = edge["id0"] == id_new or edge["id1"] == id_new
                if eid == inter_id or not (c1 or c2):
                    continue
                size = vert["size"] if c1 else G["v"][id_new]["size"]
                self.nr_edges += 2
                target = edge["p_proj"] if edge["p_proj"] is not None else end
                if me["p_proj"] is None:
                    me["p_proj"] = PO.project_polytope(set_inter[0], set_inter[1], target)
                dist = np.linalg.norm(me["p_proj"] - target)
                cs = me["conn_to_start"] or edge["conn_to_start"]
                ce = me["conn_to_end"] or edge["conn_to_end"]
                me["conn_to_start"] = edge["conn_to_start"] = cs
                me["conn_to_end"] = edge["conn_to_end"] = ce
                connected = bool(cs and ce)          # (the verdict of the LAST edge processed, BoundPlanner.py:881-884)
                cost = dist * (1 + self.w_size * np.tanh(0.25 - np.cbrt(size))) + self.w_bias + (0.0 if fit else self.c_fit)
                G["adj"][inter_id][eid] = cost
                G["adj"][eid][inter_id] = cost
        return connected

    # ---------------------------------------------------------------------------------------------------- via points
    def compute_via_points(self, path, start, end, G, with_rot=False, p_via_guess=None):
        x0 = np.empty(0)
        sets_inter = []
        for nid in path[1:-1]:
            node = G["i"][nid]
            sets_inter.append(node["cset"])          # the node's own list: shrunk and padded in place (BoundPlanner.py:593-600, 645)
            x0 = np.concatenate((x0, node["p_proj"], [0.5]))
            rows = np.linalg.norm(node["cset"][0], axis=1) > 1e-4
            node["cset"][1][rows] -= 0.001
        sets, sets_via, q_ell, p_mid, w_size = [], [], [], [], []
        last = None
        for k, nid in enumerate(path):
            node = G["i"][nid]
            if k == 0:
                a_set, b_set = node["cset"]
                last = node["id0"]
                v = G["v"][last]
                w_size.append(v["size"])
            else:
                nxt = node["id0"] if node["id0"] != last else (node["id1"] if node["id1"] != last else None)
                if nxt is not None:
                    v = G["v"][nxt]
                    a_set, b_set = v["cset"]
                    w_size.append(v["size"])
                    last = nxt
            sets.append([a_set, b_set]); sets_via.append([a_set, b_set])
            q_ell.append(v["q_ellipse"]); p_mid.append(v["p_mid"])
        w_size = 1 - np.cbrt(w_size)
        sets_inter = normalize_set_size(sets_inter, self.max_set_size)
        sets_via = normalize_set_size(sets_via, self.max_set_size)
        nr_via = len(sets_inter)
        S = self.max_set_size
        if with_rot:
            params = np.concatenate((start, end, self.l_ee, self.omega_normed, [self.omega_norm], w_size))
            for i in range(nr_via):
                params = np.concatenate((params, sets_inter[i][0].T.flatten(), sets_inter[i][1]))
            for i in range(nr_via + 1):
                params = np.concatenate((params, sets_via[i][0].T.flatten(), sets_via[i][1]))
            x0 = np.concatenate((x0, 0.5 * np.ones(S * nr_via)))
            sol_x, ok = self.solve_via_rot(nr_via, x0, params)
            self.via_rot_success = ok
            if not ok:
                self._log("(PosOpt) ERROR No convergence in via point rot optimization")
        sets_out, p_via, omega_via = [], [np.asarray(start, float)], [0.0]
        for i in range(nr_via):
            if with_rot:
                step = 4 + S
                cand, om = sol_x[step * i:step * i + 3], sol_x[step * i + 3]
            else:
                cand, om = x0[4 * i:4 * i + 3], x0[4 * i + 3]
            if np.linalg.norm(cand - p_via[-1]) > 1e-4:
                p_via.append(np.array(cand)); omega_via.append(float(om)); sets_out.append(sets[i])
            if with_rot and self.replanning and i == 0:
                # the first segment is extended backwards so that the MPC horizon lies on the new path (BoundPlanner.py:706-729)
                a0, b0 = sets_out[0]
                dp0 = p_via[1] - p_via[0]
                dp0 = dp0 / np.linalg.norm(dp0)
                lin = linprog(np.ones(1), A_ub=(a0 @ dp0)[:, None], b_ub=b0 - a0 @ p_via[0], bounds=(None, None))
                phi_h = min(float(np.min(dp0 @ (np.asarray(self.p_horizon) - p_via[0]).T)), -0.5)
                self.replanning_phi = max(-phi_h, 0.0)
                self.replanning_linprog_phi = float(lin.x[0]) if lin.x is not None else None
                p_via[0] = p_via[0] - self.replanning_phi * dp0
        p_via.append(np.asarray(end, float)); omega_via.append(1.0); sets_out.append(sets[-1])
        return np.array(p_via), p_via, omega_via, sets_out, q_ell, p_mid

    def solve_via_rot(self, nr_via, x0, params):
        """The via-point / rotation NLP (IPOPT in the reference); a hook so that a test can count its calls."""
        return PO.via_rot_problem(nr_via, self.max_set_size, x0, params)

    # ---------------------------------------------------------------------------------------------------- the planner
    def _push_out_of_obstacles(self, p):
        for a, b in self.obs_sets:
            viol = a @ p - b
            if not np.any(viol > 0):
                k = int(np.argmax(viol))
                p -= (viol[k] - self.obs_size_increase) * a[k]
        return p

    def plan_convex_set_path(self, start, end, r0, r1, replanning=False, p_horizon=(), first_sample=None, new_obs=False):
        """-> (p_via list, r_via list of 3x3, bp1_list, sets_via as [A (15x3), b (15)] padded with (0, 10) rows)."""
        start, end = np.array(start, float), np.array(end, float)
        r0, r1 = np.asarray(r0, float), np.asarray(r1, float)
        sampled_first = False
        self.replanning, self.replanning_phi, self.p_horizon = replanning, 0.0, p_horizon
        self.nr_sets = self.nr_edges = self.nr_inter_set = 0
        end = self._push_out_of_obstacles(end)
        self.omega = R.from_matrix(r1 @ r0.T).as_rotvec()
        self.omega_norm = np.linalg.norm(self.omega)
        self.omega_normed = self.omega / self.omega_norm if self.omega_norm > 1e-6 else np.array([0, 0, 1.0])
        self.l_ee = r0 @ np.array([-self.length_ee, 0, 0])
        self.l_ee_end = r1 @ np.array([-self.length_ee, 0, 0])
        G = {"v": [], "i": [], "adj": {}}
        fnd = self.set_finder

        # ---- start set
        collision = False
        if replanning:
            h_idx = 1
            for s in self.sets_via_prev:
                start_in = np.max(s[0] @ start - s[1]) < 1e-8
                out = np.where(~(np.max(s[0] @ np.array(p_horizon).T - s[1][:, None], axis=0) < 1e-8))[0]
                if out.shape[0] > 0:
                    if out[0] != 0 and start_in:
                        h_idx = max(h_idx, out[0] - 1)
                elif start_in:
                    h_idx = len(p_horizon) - 1
                    break
            if new_obs:
                h_idx = 1
            self.p_horizon_max = p_horizon[h_idx]
            a_set, b_set, q_start, mid_start, collision = fnd.find_set_collision_avoidance(start, self.p_horizon_max, True)
        else:
            a_set, b_set, q_start, mid_start = fnd.find_set_around_point(start, fixed_mid=True)
            if np.max(a_set @ (start + self.l_ee) - b_set) > 1e-8:
                a_set, b_set, q_start, mid_start, collision = fnd.find_set_collision_avoidance(start, start + self.l_ee, True)
        if collision:
            if new_obs:
                start = self._push_out_of_obstacles(start)
                a_set, b_set, q_start, mid_start = fnd.find_set_around_point(start, fixed_mid=True)
            else:       # no new set could be grown: keep the last set of the previous plan
                a_set, b_set = copy.deepcopy(self.sets_via_prev[-1][0]), copy.deepcopy(self.sets_via_prev[-1][1])
                mid_start, q_start = start, np.eye(3)
        a_set, b_set = PO.reduce_ineqs(a_set, b_set)
        set_start = [a_set, b_set]
        v0 = self._new_vertex(G, set_start, q_start, mid_start, "Vertex start")
        self._new_inter(G, cset=set_start, name="Vertex start", id0=v0, id1=v0, set0=set_start, set1=set_start, conn_to_start=True,
                        conn_to_end=False, p_proj=start, edge=None, p_via=np.concatenate((start, [0.0])))
        self.nr_sets += 1
        connected = self.add_edges(v0, G, end, start)

        if np.max(a_set @ end - b_set) < 1e-8 and np.max(a_set @ (end + self.l_ee_end) - b_set) < 1e-8:
            # the goal (and the end-effector offset there) is inside the start set: one segment
            omega_via = [0.0, 1.0]
            r_via = [R.from_rotvec(x * self.omega).as_matrix() @ r0 for x in omega_via]
            sets_normed = normalize_set_size([[a_set, b_set]], 15)
            self.sets_via_prev = sets_normed.copy()
            self.graph = G
            return [start, end], r_via, [np.array([0, 0, 1.0])], sets_normed

        # ---- end set
        a_set, b_set, q_end, mid_end, _ = fnd.find_set_collision_avoidance(end, end + self.l_ee_end, True)
        a_set, b_set = PO.reduce_ineqs(a_set, b_set)
        set_end = [a_set, b_set]
        v1 = self._new_vertex(G, set_end, q_end, mid_end, "Vertex end")
        self._new_inter(G, cset=set_end, name="Vertex end", id0=v1, id1=v1, set0=set_end, set1=set_end, conn_to_start=False,
                        conn_to_end=True, p_proj=end, edge=None, p_via=np.concatenate((end, [1.0])))
        self.nr_sets += 1
        connected = self.add_edges(v1, G, end, start) or connected

        # ---- grow the graph until the via points of the shortest path stop moving
        j = nr_samples = 0
        p_via_old = None
        while True:
            via_sample = False
            if connected:
                path = _shortest_path(G["adj"], 0, 1)
                p_via, p_via_list, omega_via, sets_via, _, _ = self.compute_via_points(path, start, end, G)
                if p_via_old is not None and p_via_old.shape == p_via.shape and np.linalg.norm(p_via_old - p_via) < 1e-4:
                    break
                samples, via_sample = p_via_list[1:-1], True
                p_via_old = np.copy(p_via)
            elif not sampled_first and first_sample is not None:
                samples = [np.asarray(first_sample, float)]
            else:
                for _try in range(self.max_samples + 1):
                    sample = self.rng.uniform(self.workspace_min, self.workspace_max, 3)
                    blocked = any(np.max(a @ sample - b) < 1e-3 for a, b in self.obs_sets)
                    known = any(np.max(v["a_set"] @ sample - v["b_set"]) < 1e-3 for v in G["v"])
                    if not blocked and not known:
                        break
                else:
                    raise RuntimeError("(PosPath) Could not find collision-free sample")
                samples = [sample]
                nr_samples += 1
                if nr_samples > self.max_iters:
                    raise RuntimeError("(PosPath) Exceeded max iterations")
            for sample in samples:
                j += 1
                optimize = nr_samples < self.nr_optimized
                # (the reference builds a one-element tuple here, which is always true: BoundPlanner.py:503-505)
                a_set, b_set, q_ell, p_mid = fnd.find_set_around_point(sample, fixed_mid=True, optimize=optimize)
                a_set, b_set = PO.reduce_ineqs(a_set, b_set)
                sampled_first = True
                d_known = min(np.linalg.norm(q_ell - v["q_ellipse"]) + np.linalg.norm(p_mid - v["p_mid"]) for v in G["v"])
                if d_known > 0.01:
                    vid = self._new_vertex(G, [a_set, b_set], q_ell, p_mid, f"Vertex {j}")
                    self.nr_sets += 1
                    connected = self.add_edges(vid, G, end, start) or connected

        # ---- final via points with the rotation schedule
        p_via, p_via_list, omega_via, sets_via, _, _ = self.compute_via_points(path, start, end, G, with_rot=True, p_via_guess=p_via_list)
        self.sets_via_prev = sets_via.copy()
        bp1_list = []
        for i in range(len(p_via) - 1):
            dp = p_via[i + 1] - p_via[i]
            dp = dp / np.linalg.norm(dp)
            b1 = _gram_schmidt(dp, np.array([0, 0, 1.0]))
            bp1_list.append(b1 / np.linalg.norm(b1))
        r_via = [R.from_rotvec(x * self.omega).as_matrix() @ r0 for x in omega_via]
        r_via[0] = R.from_rotvec(-self.replanning_phi * self.omega).as_matrix() @ r0
        self.graph = G
        return p_via_list, r_via, bp1_list, normalize_set_size(sets_via, 15)
