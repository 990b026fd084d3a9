// Link-time stand-in for the HIP runtime, for the CPU-only concurrency tests of libmovba's HOST side (tests/hipstub/):
// streams are worker threads that run queued closures in order, events order streams, "device memory" is host memory.
// Test infrastructure only: the product is never built against this header.
#pragma once
#include <cstddef>
#include <cstdint>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
struct FakeStream;
struct FakeEvent;
typedef FakeStream *hipStream_t;
typedef FakeEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2 };

hipError_t hipGetDeviceCount(int *n);
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
hipError_t hipDeviceGetAttribute(int *value, hipDeviceAttribute_t attr, int device);
hipError_t hipSetDevice(int d);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned flags);
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s);
hipError_t hipMemset(void *dst, int value, size_t bytes);

// the fake device's side of a stream: queue a closure behind everything queued so far (fake_device.cpp)
#include <functional>
void fake_enqueue(hipStream_t s, std::function<void()> f);
