"""MI355X-native scan-to-map ICP path for open3d_slam / libpointmatcher (C ABI: include/o3s_icp.h)."""
from .icp import (ICP, IcpConfig, IcpStats, ConvergenceError, TransformationError, InvalidModuleType, HipError,  # noqa: F401
                  compute_batch)
from .dense_map import DenseMap  # noqa: F401
from .submap import ProcessedScan, Submap  # noqa: F401
from .submap_collection import SubmapCollection  # noqa: F401

__all__ = ["ICP", "IcpConfig", "IcpStats", "ConvergenceError", "TransformationError", "InvalidModuleType", "HipError", "compute_batch", "Submap", "ProcessedScan", "SubmapCollection", "DenseMap"]
