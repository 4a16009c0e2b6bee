/*
 * o3s_rccl.h — C ABI of libo3dslam_icp_rccl.so: the RCCL side of the one-pair-sharded ICP mode for hosts that do not
 * run PyTorch (the reference's host is C++/catkin).  It supplies an o3s_allreduce_fn (include/o3s_icp.h) that is a plain
 * in-place ncclAllReduce(sum) on the stream the ICP kernels run on — three small, latency-bound collectives per
 * iteration over xGMI: int32 x R x 2048 and x 8192 (trim selection levels 1 and 2, LPM/Matches.cpp:61-87) and float64 x (128 +
 * 34 x 128 + 34 x blocks): level 3 together with the raw moments of the kept pairs, from which every rank forms the means
 * (LPM/ErrorMinimizers/PointToPlane.cpp:263-264) and the 6x6 normal equations with their right-hand side (PointToPlane.cpp:283-306;
 * include/o3s_icp.h, o3s_icp_shard_configure).  One process per GPU, one communicator per process.
 *
 * Kept out of libo3dslam_icp_hip.so so that the single-GPU library does not depend on librccl.
 *
 *   rank 0:      o3s_rccl_unique_id(id);  ... send the 128 bytes to the other ranks over any host channel ...
 *   every rank:  o3s_rccl_create(id, rank, world, device, &comm);
 *                o3s_icp_shard_configure(icp, rank, world, n_total, o3s_rccl_allreduce, comm, NULL);
 *                o3s_icp_set_reading(icp, <this rank's slice>);  o3s_icp_compute_resident(icp, T_init, T_out, &stats);
 */
#ifndef O3S_RCCL_H
#define O3S_RCCL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define O3S_RCCL_ID_BYTES 128

typedef struct o3s_rccl o3s_rccl;

/* All functions return 0 on success; o3s_rccl_last_error() describes the last failure on this thread. */
int o3s_rccl_unique_id(uint8_t id[O3S_RCCL_ID_BYTES]);
int o3s_rccl_create(const uint8_t id[O3S_RCCL_ID_BYTES], int32_t rank, int32_t world, int device, o3s_rccl** out);
void o3s_rccl_destroy(o3s_rccl* c);
/* Signature-compatible with o3s_allreduce_fn; `user` is the o3s_rccl*.  dtype: 0 = int32, 1 = float64. */
int o3s_rccl_allreduce(void* user, void* dev_ptr, int64_t byte_offset, int64_t count, int32_t dtype, void* hip_stream);
/* Number of collectives issued through this communicator so far. */
int64_t o3s_rccl_collectives(const o3s_rccl* c);
const char* o3s_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* O3S_RCCL_H */
