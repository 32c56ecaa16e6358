// TEST-ONLY stand-in for the slice of <rccl/rccl.h> that sharded_batch_hip.hpp calls (see ../hip/hip_runtime_api.h).  ncclAllGather keeps the
// property the aligner's design leans on: it is ENQUEUED (returns at once, the rank's contribution is taken in stream order) and completes --
// when its stream is synchronised -- only after EVERY rank of the communicator has enqueued the same collective; ncclCommAbort releases
// whoever waits.
#pragma once
#include <cstddef>
#include "../hip/hip_runtime_api.h"
extern "C" {
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclInternalError = 3 };
typedef struct fakeComm* ncclComm_t;
typedef enum { ncclChar = 0 } ncclDataType_t;
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist);
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclCommAbort(ncclComm_t comm);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
}
