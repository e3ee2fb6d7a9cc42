"""URDF-backed stub of the few `pinocchio` entry points RobotModel.__init__ touches
(golden-vector generation only; numeric FK is served by the .ca tapes instead)."""
import xml.etree.ElementTree as ET

import numpy as np

LOCAL_WORLD_ALIGNED = 2


class _Model:
    def __init__(self, path):
        root = ET.parse(str(path)).getroot()
        self.joint_names = []
        self.frames = []
        lo, hi, vel = [], [], []
        for j in root.findall("joint"):
            self.frames.append(j.get("name"))
            self.frames.append(j.find("child").get("link"))
            if j.get("type") == "revolute":
                self.joint_names.append(j.get("name"))
                lim = j.find("limit")
                lo.append(float(lim.get("lower")))
                hi.append(float(lim.get("upper")))
                vel.append(float(lim.get("velocity")))
        self.lowerPositionLimit = np.array(lo)
        self.upperPositionLimit = np.array(hi)
        self.velocityLimit = np.array(vel)

    def getFrameId(self, name):
        return self.frames.index(name)

    def getJointId(self, name):
        return self.joint_names.index(name) + 1

    def createData(self):
        raise RuntimeError("numeric pinocchio FK is not available in the stub")


def buildModelsFromUrdf(path, package_dirs=None):
    return _Model(path), None, None
