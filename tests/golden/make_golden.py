"""Regenerates the data fixtures in tests/golden/ from the reference's own test data files.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The fixtures are DATA (inputs + expected outputs held by the reference's tests), not source:
  * car_cloud400 / car_cloud401: libpointmatcher/examples/data/car_cloud400.csv (x,y,z,nx,ny,nz; header) and car_cloud401.csv (x y z; no header), loaded by
    libpointmatcher/utest/utest.cpp:74-77 as ref3D / data3D.
  * validT3d: the expected transform hard-coded at libpointmatcher/utest/utest.cpp:85-89, tolerance
    0.1 m / 0.1 rad (libpointmatcher/utest/utest.h:83-84).
"""
import os

import numpy as np

REF = "/root/reference/libpointmatcher/examples/data/"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_csv(name):
    with open(REF + name) as f:
        first = f.readline()
    has_header = any(c.isalpha() and c not in "eE" for c in first)
    delim = "," if "," in first else None
    a = np.loadtxt(REF + name, delimiter=delim, skiprows=1 if has_header else 0, dtype=np.float64)
    return a.astype(np.float32)


def main():
    c400 = load_csv("car_cloud400.csv")
    c401 = load_csv("car_cloud401.csv")
    validT3d = np.array([[0.982304, 0.166685, -0.0854066, 0.0446816],
                         [-0.150189, 0.973488, 0.172524, 0.191998],
                         [0.111899, -0.156644, 0.981296, -0.0356313],
                         [0, 0, 0, 1]], np.float32)
    np.savez_compressed(os.path.join(HERE, "car_clouds.npz"), ref3D=c400, data3D=c401, validT3d=validT3d)
    print("car_cloud400", c400.shape, "car_cloud401", c401.shape)


if __name__ == "__main__":
    main()
