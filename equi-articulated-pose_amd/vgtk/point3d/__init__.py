from .base import PointSet  # noqa: F401
