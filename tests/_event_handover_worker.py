"""Worker of tests/test_gpu_parity.py::test_device_resident_inputs_handed_over_with_an_event_equal_the_host_path (GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from open3d_slam_advanced_rss_2024_public_amd import ICP, IcpConfig  # noqa: E402
from open3d_slam_advanced_rss_2024_public_amd import synthetic as syn  # noqa: E402

sp = syn.make_scan_pair(6000, 50000, 0.1, seed=12)
dev = torch.device("cuda", 0)
producer = torch.cuda.Stream(dev)
with torch.cuda.stream(producer):
    ref = torch.ones((sp.map_xyz.shape[0], 4), dtype=torch.float32, device=dev)
    ref[:, :3] = torch.from_numpy(np.ascontiguousarray(sp.map_xyz, np.float32)).to(dev, non_blocking=True)
    refn = torch.from_numpy(np.ascontiguousarray(sp.map_normals, np.float32)).to(dev, non_blocking=True).contiguous()
    rd = torch.ones((sp.scan_xyz.shape[0], 4), dtype=torch.float32, device=dev)
    rd[:, :3] = torch.from_numpy(np.ascontiguousarray(sp.scan_xyz, np.float32)).to(dev, non_blocking=True)
    rdn = torch.from_numpy(np.ascontiguousarray(sp.scan_normals, np.float32)).to(dev, non_blocking=True).contiguous()
    ev = torch.cuda.Event()
    ev.record(producer)
g = ICP(IcpConfig())
g.wait_event(ev.cuda_event)
assert g.init_reference_dev_async(ref.data_ptr(), refn.data_ptr(), ref.shape[0])
g.set_reading_dev(rd.data_ptr(), rdn.data_ptr(), rd.shape[0])
T_dev = g.compute_resident(sp.T_init)

host = ICP(IcpConfig())
assert host.init_reference(sp.map_xyz, sp.map_normals)
T_host = host.compute(sp.scan_xyz, sp.scan_normals, sp.T_init)
assert np.array_equal(T_dev, T_host)
n = host.stats.iterations
assert g.stats.iterations == n
assert np.array_equal(g.stats.trace_limit[:n].view(np.uint32), host.stats.trace_limit[:n].view(np.uint32))
assert np.array_equal(g.stats.trace_kept[:n], host.stats.trace_kept[:n])
print("handover ok", n)
