// TEST-ONLY stand-in for the slice of <hip/hip_runtime_api.h> that riv-slam_amd/cpp/sharded_batch_hip.hpp calls, so that the aligner's HOST
// logic -- worker threads, queues, slots, late gathers, abort -- runs on the CPU box under ThreadSanitizer / AddressSanitizer
// (tests/test_sanitizers.py).  Never shipped, never on the include path of a product build.  Semantics kept: streams are in-order
// queues with an executor thread each (they make progress on their own, like device queues), events mark a position in a stream, "device"
// memory is host memory.  Implemented in tests/cpp/fake/fake_backend.cpp.
#pragma once
#include <cstddef>
extern "C" {
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
typedef struct fakeStream* hipStream_t;
typedef struct fakeEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
hipError_t hipSetDevice(int device);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemset(void* p, int v, size_t bytes);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
}
