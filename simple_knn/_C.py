"""`from simple_knn._C import distCUDA2` (scene/gaussian_model.py:20,134)."""
from event_3dgs_amd.knn import distCUDA2  # noqa: F401
