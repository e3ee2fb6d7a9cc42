"""Robot tables: the kinematic chain, limits and collision-sphere radii of the arms the reference supports
(RobotModel/RobotModel.py:10-54: `USE_IIWA` selects iiwa.urdf or gen3_arm.urdf; same joint / frame names in both files).

A table is what `include/boundmpc.h` calls `bmpc_robot`: URDF <origin xyz rpy> of joint_1..joint_7 (revolute about local z),
the fixed joints to `end_effector_link` and `link4_col_link`, URDF <limit>s (RobotModel.py:44-48 removes the +-10 rad limits of
the Gen3's continuous joints), acceleration / jerk limits (BoundMPC.py:182-186) and `col_joint_sizes` (RobotModel.py:37-40).
`table_from_urdf` builds one from any URDF with those names; IIWA14 and GEN3 are the two files of the reference as data
(tests/golden/gen/gen_robot_tables.py checks them against the files)."""
import ctypes
import xml.etree.ElementTree as ET

import numpy as np

PI_2, PI_1 = 1.5707963267948966, 3.141592653589793
BIG = 1e20

IIWA14 = dict(
    name="iiwa14",
    joint_xyz=[[0, 0, 0.1525], [0, 0, 0.2075], [0, 0.2325, 0], [0, 0, 0.1875], [0, 0.2125, 0], [0, 0, 0.1875], [0, 0.0796, 0]],
    joint_rpy=[[0, 0, 0], [PI_2, 0, PI_1], [PI_2, 0, PI_1], [PI_2, 0, 0], [-PI_2, PI_1, 0], [PI_2, 0, 0], [-PI_2, PI_1, 0]],
    ee_xyz=[0, 0, 0.21], ee_rpy=[0, -1.575, -1.575], link4_col_xyz=[0, 0.3, 0],
    q_lower=[-2.9670597283903604, -2.0943951023931953, -2.9670597283903604, -2.0943951023931953, -2.9670597283903604,
             -2.0943951023931953, -3.0543261909900763],
    q_upper=[2.9670597283903604, 2.0943951023931953, 2.9670597283903604, 2.0943951023931953, 2.9670597283903604,
             2.0943951023931953, 3.0543261909900763],
    dq_max=[10.0] * 7, ddq_max=5.0, u_max=35.0, col_joint_sizes=[0.09, 0.12, 0.09, 0.10, 0.07, 0.09, 0.075])

GEN3 = dict(
    name="gen3",
    joint_xyz=[[0, 0, 0.15643], [0, 0.005375, -0.12838], [0, -0.21038, -0.006375], [0, 0.006375, -0.21038],
               [0, -0.20843, -0.006375], [0, 0.00017505, -0.10593], [0, -0.10593, -0.00017505]],
    joint_rpy=[[3.1416, 2.7629E-18, -4.9305E-36], [1.5708, 2.1343E-17, -1.1102E-16], [-1.5708, 1.2326E-32, -2.9122E-16],
               [1.5708, -6.6954E-17, -1.6653E-16], [-1.5708, 2.2204E-16, -6.373E-17], [1.5708, 9.2076E-28, -8.2157E-15],
               [-1.5708, -5.5511E-17, 9.6396E-17]],
    ee_xyz=[0, 0, -0.20], ee_rpy=[0, 1.570796326794895, 1.570796326794895], link4_col_xyz=[0, -0.1, 0.0],
    q_lower=[-BIG, -2.24, -BIG, -2.57, -BIG, -2.09, -BIG], q_upper=[BIG, 2.24, BIG, 2.57, BIG, 2.09, BIG],
    dq_max=[1.3963, 1.3963, 1.3963, 1.3963, 1.2218, 1.2218, 1.2218], ddq_max=5.0, u_max=35.0,
    col_joint_sizes=[0.09, 0.09, 0.06, 0.06, 0.06, 0.06, 0.075])


class BmpcRobot(ctypes.Structure):
    _fields_ = [("joint_xyz", ctypes.c_double * 21), ("joint_rpy", ctypes.c_double * 21), ("ee_xyz", ctypes.c_double * 3),
                ("ee_rpy", ctypes.c_double * 3), ("link4_col_xyz", ctypes.c_double * 3), ("q_lower", ctypes.c_double * 7),
                ("q_upper", ctypes.c_double * 7), ("dq_max", ctypes.c_double * 7), ("ddq_max", ctypes.c_double),
                ("u_max", ctypes.c_double), ("col_joint_sizes", ctypes.c_double * 7)]


def to_struct(table):
    r = BmpcRobot()
    for k in ("joint_xyz", "joint_rpy", "ee_xyz", "ee_rpy", "link4_col_xyz", "q_lower", "q_upper", "dq_max", "col_joint_sizes"):
        a = np.asarray(table[k], float).ravel()
        getattr(r, k)[:] = a.tolist()
    r.ddq_max, r.u_max = float(table["ddq_max"]), float(table["u_max"])
    return r


def from_struct(r, name="custom"):
    t = dict(name=name, ddq_max=r.ddq_max, u_max=r.u_max)
    for k, shape in (("joint_xyz", (7, 3)), ("joint_rpy", (7, 3)), ("ee_xyz", (3,)), ("ee_rpy", (3,)), ("link4_col_xyz", (3,)),
                     ("q_lower", (7,)), ("q_upper", (7,)), ("dq_max", (7,)), ("col_joint_sizes", (7,))):
        t[k] = np.array(list(getattr(r, k))).reshape(shape).tolist()
    return t


def table_from_urdf(path_or_text, col_joint_sizes, continuous_unlimited=True, ddq_max=5.0, u_max=35.0, name="urdf"):
    """Robot table from a URDF with the reference's names: revolute joints joint_1..joint_7 (axis 0 0 1), fixed joints whose
    children are end_effector_link and link4_col_link.  `continuous_unlimited`: joints whose limits are +-10 rad or wider
    are treated as unlimited (RobotModel.py:46-48)."""
    text = open(path_or_text).read() if "<" not in path_or_text else path_or_text
    root = ET.fromstring(text)
    joints = {j.get("name"): j for j in root.iter("joint")}
    by_child = {j.find("child").get("link"): j for j in root.iter("joint") if j.find("child") is not None}

    def origin(j):
        o = j.find("origin")
        f = lambda s: [float(v) for v in (s or "0 0 0").split()]
        return f(o.get("xyz") if o is not None else None), f(o.get("rpy") if o is not None else None)

    t = dict(name=name, joint_xyz=[], joint_rpy=[], q_lower=[], q_upper=[], dq_max=[], ddq_max=ddq_max, u_max=u_max,
             col_joint_sizes=list(col_joint_sizes))
    parent_of = {}
    for i in range(1, 8):
        j = joints[f"joint_{i}"]
        ax = [float(v) for v in j.find("axis").get("xyz").split()]
        if j.get("type") not in ("revolute", "continuous") or ax != [0.0, 0.0, 1.0]:
            raise ValueError(f"joint_{i}: revolute about local z expected")
        xyz, rpy = origin(j)
        t["joint_xyz"].append(xyz); t["joint_rpy"].append(rpy)
        lim = j.find("limit")
        lo, hi = float(lim.get("lower", -BIG)), float(lim.get("upper", BIG))
        if j.get("type") == "continuous" or (continuous_unlimited and lo <= -10.0 and hi >= 10.0):
            lo, hi = -BIG, BIG
        t["q_lower"].append(lo); t["q_upper"].append(hi); t["dq_max"].append(float(lim.get("velocity")))
        parent_of[j.find("child").get("link")] = i
    ee, l4 = by_child["end_effector_link"], by_child["link4_col_link"]
    if parent_of.get(ee.find("parent").get("link")) != 7 or parent_of.get(l4.find("parent").get("link")) != 4:
        raise ValueError("end_effector_link must hang off joint_7's link and link4_col_link off joint_4's link")
    t["ee_xyz"], t["ee_rpy"] = origin(ee)
    t["link4_col_xyz"] = origin(l4)[0]
    return t


def chain_fk(table, q):
    """Plain numpy forward kinematics of a table (URDF semantics: T = T_parent * Trans(xyz) * Rz(y) Ry(p) Rx(r) * Rz(q)):
    end-effector position / rotation and the six collision points.  Independent of the HIP kernels and of the oracle."""
    def rot(rpy):
        r, p, y = rpy
        cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
        return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                         [-sp, cp * sr, cp * cr]])
    R, t = np.eye(3), np.zeros(3)
    origins = []
    for i in range(7):
        t = t + R @ np.asarray(table["joint_xyz"][i], float)
        R = R @ rot(table["joint_rpy"][i])
        origins.append(t.copy())
        c, s = np.cos(q[i]), np.sin(q[i])
        R = R @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
        if i == 3:
            l4 = t + R @ np.asarray(table["link4_col_xyz"], float)
    ee_pos = t + R @ np.asarray(table["ee_xyz"], float)
    ee_rot = R @ rot(table["ee_rpy"])
    return ee_pos, ee_rot, np.array(origins[2:7] + [l4])
