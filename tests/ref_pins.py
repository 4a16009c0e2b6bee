"""Vectors the reference's own tests hold, kept as data for both the oracle tests (CPU) and the parity tests (GPU)."""
import numpy as np

# utest/ui/Transformations.cpp:133-147 (TEST(Transformation, RigidTransformationParameterCheck)): the reference's non-orthogonal
# 3-D matrix (Eigen's comma initialiser fills row by row) — checkParameters is false (|1 - det| = 1.98e-3 > 1e-3) and
# compute / inPlaceCompute throw TransformationError (TransformationsImpl.cpp:73-74, 98-113).
REF_T3D_NOT_RIGID = np.array([[0.99935116, 0.13669771, 0.03436585, 1.71138524],
                              [-0.02633967, 0.99326295, -0.04907545, -0.10860933],
                              [-0.03615132, 0.04400287, 0.99820427, -0.04454497],
                              [0.0, 0.0, 0.0, 1.0]], np.float32)
