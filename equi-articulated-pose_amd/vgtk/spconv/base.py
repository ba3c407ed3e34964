"""Containers passed between conv blocks (reference: vgtk/vgtk/spconv/base.py:L4-47).

xyz [B,3,P], feats [B,C,P,A], anchors [A,3,3], pose [B,P,4,4]."""
from vgtk.point3d import PointSet


class SphericalPointCloud():
    def __init__(self, xyz, feats, anchors):
        self._xyz = PointSet(xyz)
        self._feats = feats
        self._anchors = anchors

    @property
    def xyz(self):
        return self._xyz.data

    @property
    def feats(self):
        return self._feats

    @property
    def anchors(self):
        return self._anchors


class SphericalPointCloudPose(SphericalPointCloud):
    def __init__(self, xyz, feats, anchors, pose):
        super().__init__(xyz, feats, anchors)
        self._pose = pose

    @property
    def pose(self):
        return self._pose
