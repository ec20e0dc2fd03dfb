"""mcm_amd — MI355X-native Maximum Concept Matching scorer.

Host side of the hot path `get_ood_scores_clip` (reference utils/detection_util.py:209-249):
Python over the C ABI of libmcm_hip.so (include/mcm.h).  There is no CPU fallback: the
engine raises if the HIP library is missing.
"""
from .config import CHECKPOINTS, TEST_GEOMETRIES, ClipGeometry, geometry  # noqa: F401

__all__ = ["CHECKPOINTS", "TEST_GEOMETRIES", "ClipGeometry", "geometry"]
