"""boundplanner_amd: MI355X-native batched solver for the BoundMPC receding-horizon step.

The package holds only what that path needs: csrc/ (HIP kernels + the C-ABI library
libboundmpc_hip.so), a ctypes binding (solver.py) and the host-side mirror of the reference's
BoundMPC / ReferencePath / RobotModel interfaces.  See DESIGN.md.
"""
