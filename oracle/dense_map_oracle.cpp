// dense_map_oracle.cpp — CPU restatement of o3d_slam::VoxelizedPointCloud and of the dense-map space carving.
// TEST INFRASTRUCTURE ONLY; see dense_map_oracle.h.  Sequential, fp64, no FMA contraction (-ffp-contract=off).
#include "dense_map_oracle.h"

#include <array>
#include <cmath>
#include <map>
#include <set>
#include <vector>

namespace {
using Key = std::array<int32_t, 3>;
struct KeyLess {  // ascending (z, y, x): the order the device map reports its voxels in
  bool operator()(const Key& a, const Key& b) const {
    if (a[2] != b[2]) return a[2] < b[2];
    if (a[1] != b[1]) return a[1] < b[1];
    return a[0] < b[0];
  }
};
// AggregatedVoxel (O3S/include/open3d_slam/Voxel.hpp:38-57); colours are not carried
struct Voxel {
  int num = 0;
  double p[3] = {0, 0, 0}, n[3] = {0, 0, 0};
};
// getVoxelIdx(p, InverseVoxelSize) (VoxelHashMap.hpp:48-51): the reciprocal form
inline Key key_recip(const double* p, double inv) {
  return {(int32_t)std::floor(p[0] * inv), (int32_t)std::floor(p[1] * inv), (int32_t)std::floor(p[2] * inv)};
}
// getVoxelIdx(p, voxelSize) (VoxelHashMap.hpp:53-56): the dividing form
inline Key key_div(const double* p, double voxel) {
  return {(int32_t)std::floor(p[0] / voxel), (int32_t)std::floor(p[1] / voxel), (int32_t)std::floor(p[2] / voxel)};
}
// getVoxelsWithinPointNeighborhood (VoxelHashMap.cpp:13-46)
template <typename F>
void neighbourhood(const double* p, double radius, double voxel, F&& emit) {
  const Key centre = key_div(p, voxel);
  if (radius <= 0.0) {
    emit(centre);
    return;
  }
  bool centre_added = false;
  for (double dx = -radius; dx <= radius; dx += voxel) {
    for (double dy = -radius; dy <= radius; dy += voxel) {
      for (double dz = -radius; dz <= radius; dz += voxel) {
        const double t[3] = {p[0] + dx, p[1] + dy, p[2] + dz};
        const Key k = key_div(t, voxel);
        // getVoxelCenter: key * voxel + voxel * 0.5 (VoxelHashMap.hpp:70-72)
        const double ex = t[0] - ((double)k[0] * voxel + voxel * 0.5), ey = t[1] - ((double)k[1] * voxel + voxel * 0.5),
                     ez = t[2] - ((double)k[2] * voxel + voxel * 0.5);
        if (std::sqrt((ex * ex + ey * ey) + ez * ez) <= radius) {
          emit(k);
          if (k == centre) centre_added = true;
        }
      }
    }
  }
  if (!centre_added) emit(centre);
}
}  // namespace

struct orc_dense_map {
  double voxel = 0.25, inv = 4.0;
  std::map<Key, Voxel, KeyLess> vox;
};

extern "C" {

orc_dense_map* orc_dense_create(double voxel_size) {
  orc_dense_map* m = new orc_dense_map();
  m->voxel = voxel_size;
  m->inv = 1.0 / voxel_size;
  return m;
}
void orc_dense_destroy(orc_dense_map* m) { delete m; }
int64_t orc_dense_size(const orc_dense_map* m) { return (int64_t)m->vox.size(); }

void orc_dense_insert(orc_dense_map* m, const double* pts, const double* normals, int64_t N) {
  for (int64_t i = 0; i < N; ++i) {
    Voxel& v = m->vox[key_recip(pts + 3 * i, m->inv)];
    for (int d = 0; d < 3; ++d) v.p[d] += pts[3 * i + d];
    ++v.num;
    if (normals)
      for (int d = 0; d < 3; ++d) v.n[d] += normals[3 * i + d];
  }
}

int64_t orc_dense_to_point_cloud(const orc_dense_map* m, double* pts, double* normals, int32_t* keys, int32_t* counts) {
  int64_t o = 0;
  for (const auto& kv : m->vox) {
    const Voxel& v = kv.second;
    if (v.num <= 0) continue;
    for (int d = 0; d < 3; ++d) {
      pts[3 * o + d] = v.p[d] / (double)v.num;
      if (normals) normals[3 * o + d] = v.n[d] / (double)v.num;
      if (keys) keys[3 * o + d] = kv.first[d];
    }
    if (counts) counts[o] = v.num;
    ++o;
  }
  return o;
}

void orc_dense_transform(orc_dense_map* m, const double* T) {
  // Isometry3d * Vector3d = translation + linear * v, the 3-term products summed left to right
  auto apply = [&](double* s) {
    const double x = s[0], y = s[1], z = s[2];
    for (int r = 0; r < 3; ++r) s[r] = T[12 + r] + ((T[r] * x + T[4 + r] * y) + T[8 + r] * z);
  };
  for (auto it = m->vox.begin(); it != m->vox.end();) {
    if (it->second.num > 0) {
      apply(it->second.n);
      apply(it->second.p);
      ++it;
    } else {
      it = m->vox.erase(it);
    }
  }
}

int64_t orc_remove_duplicate_points(const double* pts, int64_t N, double voxel_size, uint8_t* keep) {
  const double inv = 1.0 / voxel_size;
  std::set<Key, KeyLess> seen;
  int64_t n = 0;
  for (int64_t i = 0; i < N; ++i) {
    keep[i] = seen.insert(key_recip(pts + 3 * i, inv)).second ? 1 : 0;
    n += keep[i];
  }
  return n;
}

int64_t orc_voxels_within_neighborhood(const double* p3, double radius, double voxel_size, int32_t* keys, int64_t cap) {
  int64_t n = 0;
  neighbourhood(p3, radius, voxel_size, [&](const Key& k) {
    if (n < cap)
      for (int d = 0; d < 3; ++d) keys[3 * n + d] = k[d];
    ++n;
  });
  return n;
}

int64_t orc_dense_carve(orc_dense_map* m, const double* scan, int64_t N, const double* s, double radius, double max_length,
                        double truncation) {
  if (m->vox.empty() || N == 0) return 0;
  std::vector<uint8_t> keep((size_t)N);
  orc_remove_duplicate_points(scan, N, m->voxel, keep.data());
  const double step = 2.0 * radius;
  if (!(step > 0.0)) return 0;  // the reference would never leave its while loop
  std::set<Key, KeyLess> to_remove;
  for (int64_t i = 0; i < N; ++i) {
    if (!keep[i]) continue;
    const double dx = scan[3 * i] - s[0], dy = scan[3 * i + 1] - s[1], dz = scan[3 * i + 2] - s[2];
    const double length = std::sqrt((dx * dx + dy * dy) + dz * dz);
    if (!(length > 0.0)) continue;  // NaN direction: no voxel can be addressed
    const double ux = dx / length, uy = dy / length, uz = dz / length;
    double distance = 0.0;
    const double max_path = std::max(step, std::min(length - truncation, max_length));
    while (distance < max_path) {
      const double c[3] = {distance * ux + s[0], distance * uy + s[1], distance * uz + s[2]};
      neighbourhood(c, radius, m->voxel, [&](const Key& k) {
        if (m->vox.count(k)) to_remove.insert(k);
      });
      distance += step;
    }
  }
  for (const Key& k : to_remove) m->vox.erase(k);
  return (int64_t)to_remove.size();
}

}  // extern "C"
