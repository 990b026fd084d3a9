// The stand-in runtime of hip/hip_runtime.h (this directory): one worker thread per stream, FIFO; events as generation
// counters under a mutex.  Everything a real device would do asynchronously happens on these threads, so ThreadSanitizer sees
// every hand-off between the API's threads and "the device" that is not ordered by a stream, an event or an atomic.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

struct FakeStream {
    std::mutex m;
    std::condition_variable cv, idle_cv;
    std::deque<std::function<void()>> q;
    bool running = false, quit = false;
    std::thread th;
    FakeStream()
    {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [&] { return quit || !q.empty(); });
                if (q.empty()) return;
                std::function<void()> f = std::move(q.front());
                q.pop_front();
                running = true;
                lk.unlock();
                f();
                lk.lock();
                running = false;
                if (q.empty()) idle_cv.notify_all();
            }
        });
    }
    ~FakeStream()
    {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv.notify_all();
        th.join();
    }
};

struct FakeEvent {
    std::mutex m;
    std::condition_variable cv;
    uint64_t recorded = 0, completed = 0;
};

void fake_enqueue(hipStream_t s, std::function<void()> f)
{
    { std::lock_guard<std::mutex> lk(s->m); s->q.push_back(std::move(f)); }
    s->cv.notify_all();
}

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *value, hipDeviceAttribute_t, int) { *value = 256; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "fake hip error"; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new FakeStream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s)
{
    std::unique_lock<std::mutex> lk(s->m);
    s->idle_cv.wait(lk, [&] { return s->q.empty() && !s->running; });
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = new FakeEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new FakeEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    uint64_t gen;
    { std::lock_guard<std::mutex> lk(e->m); gen = ++e->recorded; }
    fake_enqueue(s, [e, gen] { { std::lock_guard<std::mutex> lk(e->m); if (e->completed < gen) e->completed = gen; } e->cv.notify_all(); });
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    uint64_t gen;
    { std::lock_guard<std::mutex> lk(e->m); gen = e->recorded; }
    fake_enqueue(s, [e, gen] { std::unique_lock<std::mutex> lk(e->m); e->cv.wait(lk, [&] { return e->completed >= gen; }); });
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e)
{
    std::unique_lock<std::mutex> lk(e->m);
    const uint64_t gen = e->recorded;
    e->cv.wait(lk, [&] { return e->completed >= gen; });
    return hipSuccess;
}
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t bytes) { return posix_memalign(p, 256, bytes ? bytes : 256) == 0 ? hipSuccess : hipErrorUnknown; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) { std::memcpy(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t s)
{
    fake_enqueue(s, [=] { std::memcpy(dst, src, bytes); });
    return hipSuccess;
}
hipError_t hipMemset(void *dst, int value, size_t bytes) { std::memset(dst, value, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s)
{
    fake_enqueue(s, [=] { std::memset(dst, value, bytes); });
    return hipSuccess;
}
