"""Build-container check (nothing is generated): the robot tables of boundplanner_amd/robots.py and of
boundplanner_amd/csrc/bmpc_robot.hpp equal the reference's URDF files number for number.
    python tests/golden/gen/gen_robot_tables.py"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
sys.path.insert(0, ROOT)
from boundplanner_amd import robots  # noqa: E402

REF = "/root/reference/bound_planner/RobotModel/"
for tab, f in ((robots.IIWA14, "iiwa.urdf"), (robots.GEN3, "gen3_arm.urdf")):
    t = robots.table_from_urdf(REF + f, tab["col_joint_sizes"])
    for k in ("joint_xyz", "joint_rpy", "ee_xyz", "ee_rpy", "link4_col_xyz", "q_lower", "q_upper", "dq_max"):
        assert np.array_equal(np.asarray(t[k], float), np.asarray(tab[k], float)), (f, k)
    print(f, "== robots." + tab["name"].upper())
