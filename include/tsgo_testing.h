/* tsgo_testing.h — entry points that exist ONLY in libtsgo_hip_testing.so (built with -DTSGO_TESTING, toyslam_amd/build.py).
 *
 * The shipped libtsgo_hip.so and graph_optimizer contain neither these symbols nor the test hooks / research variables of
 * toyslam_amd/csrc/host/knobs.h (TSGO_INJECT_AMG_FAILURE, TSGO_FORCE_HOST_SLOW, TSGO_FORCE_PACED, TSGO_SYM_DECLINE, TSGO_HIER_*,
 * TSGO_AGG*, TSGO_HOST_PRODUCTS, ...).  Everything else — kernels, host code, the C ABI of tsgo.h — is the same source. */
#ifndef TSGO_TESTING_H
#define TSGO_TESTING_H

#include "tsgo.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The same sharded path among handles of ONE process (one thread per handle; they may share a device): the all-reduces
 * go through host memory instead of RCCL.  For tests on a box with a single GPU, where RCCL refuses two ranks on one
 * device — it is what lets `world` = 2, 3 run the device kernels' ownership rules there.  The group outlives its handles. */
typedef struct tsgo_local_group tsgo_local_group;
int tsgo_local_group_create(int32_t world, tsgo_local_group** out);
void tsgo_local_group_destroy(tsgo_local_group* group);
int tsgo_comm_init_local(tsgo_optimizer* opt, tsgo_local_group* group);

#ifdef __cplusplus
}
#endif
#endif /* TSGO_TESTING_H */
