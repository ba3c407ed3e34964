"""vgtk.point3d.PointSet -- thin wrapper around a [(b,) 3|4, n] coordinate tensor
(reference: vgtk/vgtk/point3d/base.py).  Only what SphericalPointCloud needs."""
import torch


class PointSet():
    def __init__(self, p):
        self._p = p

    @property
    def is_hom(self):
        return self._p.shape[-2] == 4

    @property
    def n_batch(self):
        return self._p.shape[0]

    @property
    def n_point(self):
        return self._p.shape[-1]

    @property
    def device(self):
        return self._p.device

    @property
    def data(self):
        return self._p

    def to_hom(self):
        if self.is_hom:
            return PointSet(self._p)
        ones = torch.ones(self.n_batch, 1, self.n_point, device=self.device, dtype=self._p.dtype)
        return PointSet(torch.cat((self._p, ones), dim=-2))

    def from_hom(self):
        return PointSet(self._p if not self.is_hom else self._p[..., :3, :])
