"""Python mirror of cpp/o3s_submap_collection.hpp (the host bookkeeping of o3d_slam::SubmapCollection,
open3d_slam/src/SubmapCollection.cpp:94-247, over device-resident submaps): used by the tests to check the compiled header
step by step and by tools/mapping_loop.py.  No compute here — every cloud operation is a call into the C-ABI library."""
import numpy as np

from . import cloud_ops as co
from .submap import ProcessedScan, Submap


class SubmapCollection:
    """SubmapCollection::insertScan / updateActiveSubmap (SubmapCollection.cpp:94-247) restated over the Python mirror — the
    same steps as cpp/o3s_submap_collection.hpp, resident scans in a ring of numScansOverlap + 1 objects."""

    def __init__(self, radius, min_num, max_points, overlap, map_voxel, map_builder_cropper, submap_factory=None, scan_factory=None):
        """map_builder_cropper: (kind, p0[, p1, p2]) as for cloud_ops.croppingVolumeFactory.  submap_factory / scan_factory:
        stand-ins for the device-resident objects (insertProcessed / __len__ / computeSubmapCenter) — the CPU tests of the
        switching rules use them; the default is the real thing."""
        self.radius, self.min_num, self.max_points, self.overlap = radius, min_num, max_points, overlap
        self.map_voxel, self.cropper = map_voxel, tuple(map_builder_cropper)
        self._new_submap = submap_factory or (lambda: Submap(self.map_voxel, co.croppingVolumeFactory(*self.cropper)))
        scan_factory = scan_factory or ProcessedScan
        self.maps, self.ids, self.parents, self.origins, self.centers = [], [], [], [], []
        self.active, self.next_id, self.merged, self.force = 0, 0, 0, False
        self.edges = set()
        self.buffer, self.free = [], [scan_factory() for _ in range(overlap + 1)]
        self.finished, self.finished_queue, self.switched = [], [], False
        self.create(np.zeros(3))

    def create(self, origin):
        self.maps.append(self._new_submap())
        self.ids.append(self.next_id)
        self.parents.append(self.active)
        self.next_id += 1
        self.origins.append(np.array(origin, np.float64))
        self.centers.append(None)
        self.active = len(self.maps) - 1
        self.merged = 0

    def centre(self, i):
        return self.centers[i] if self.centers[i] is not None else self.origins[i]

    @staticmethod
    def dist(a, b):
        d = a - b
        return np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])

    def scan_for_next(self):
        return self.free[-1]

    def adjacent(self, a, b):
        return a == b or (min(a, b), max(a, b)) in self.edges

    def update_active(self, p0):
        if self.force:
            self.create(p0)
            self.force = False
            return
        if self.merged < self.min_num:
            return
        closest = 0
        for i in range(1, len(self.maps)):
            if self.dist(p0, self.centre(i)) < self.dist(p0, self.centre(closest)):
                closest = i
        active = self.active
        if len(self.maps[active]) > self.max_points:
            self.force = True
        if self.dist(p0, self.centre(closest)) < self.radius:
            if closest == active:
                return
            if self.adjacent(self.ids[closest], self.ids[active]):
                self.active = closest
            elif self.dist(p0, self.centre(active)) > self.radius:
                self.create(p0)
        else:
            self.create(p0)

    def insert(self, ps, T, stamp):
        self.switched = False
        prev = self.active
        assert self.free and self.free[-1] is ps
        self.free.pop()
        self.buffer.append((ps, T.copy(), stamp))
        while len(self.buffer) > self.overlap:
            self.free.append(self.buffer.pop(0)[0])
        self.update_active(T[:3, 3].copy())
        if prev != self.active:
            self.switched = True
            self.maps[prev].insertProcessed(ps, T)
            self.centers[prev] = self.maps[prev].computeSubmapCenter()
            self.finished.append((prev, stamp))
            self.finished_queue.append((prev, stamp))
            self.merged = 0
            a, b = self.ids[prev], self.ids[self.active]
            self.edges.add((min(a, b), max(a, b)))
            while self.buffer:
                q, Tq, _ = self.buffer.pop(0)
                self.maps[self.active].insertProcessed(q, Tq)
                self.free.append(q)
            assert len(self.maps[self.active]) > 0
        else:
            self.maps[self.active].insertProcessed(ps, T)
        self.merged += 1

    def pop_finished(self):
        """SubmapCollection::popFinishedSubmapIds (:53-55)."""
        out, self.finished_queue = self.finished_queue, []
        return out
