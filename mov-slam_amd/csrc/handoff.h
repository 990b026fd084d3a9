// Words that travel between workgroups INSIDE a launch (cdna_hip_programming.md Guideline 16, write-through form; first
// row of MI355X_MICROARCH.md's hand-off table): every such word is a GLOBAL agent-scope access — stores leave the XCD's L2
// at once (`global_store ... sc1`), loads bypass the CU's L1 (`global_load ... sc1`), never flat_, never through the scalar
// cache.  Producer: sc1 stores, `s_waitcnt vmcnt(0)` in every storing wave, then ONE lane signals (an sc1 flag store or an
// agent-scope atomic add); consumer: learns it from an sc1 poll or from the value its own add returned, and reads the bytes
// with sc1 loads only.  dense_persist.hip keeps its own copies of the 8-byte forms (same instructions).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace movba {

typedef __attribute__((address_space(1))) long long hx_g_i64;
typedef __attribute__((address_space(1))) unsigned hx_g_u32;

__device__ __forceinline__ double hx_ld_f64(const double *p)
{
    return __longlong_as_double(__hip_atomic_load((const hx_g_i64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void hx_st_f64(double *p, double v)
{
    __hip_atomic_store((hx_g_i64 *)p, __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int hx_ld_i32(const int *p) { return (int)__hip_atomic_load((const hx_g_u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hx_st_i32(int *p, int v) { __hip_atomic_store((hx_g_u32 *)p, (unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned hx_ld_u32(const unsigned *p) { return __hip_atomic_load((const hx_g_u32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void hx_st_u32(unsigned *p, unsigned v) { __hip_atomic_store((hx_g_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// returning agent-scope add: the caller's later loads are ordered behind the returned value by its use
__device__ __forceinline__ unsigned hx_add_u32(unsigned *p, unsigned v) { return __hip_atomic_fetch_add((hx_g_u32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the storing wave's drain in front of its signal (inline asm: invisible to the pass that would drop a builtin wait)
__device__ __forceinline__ void hx_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 16-byte sc1 accesses through a buffer descriptor (aux 16 = sc1), for payloads read or written two doubles at a time
typedef unsigned int hx_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t hx_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double2 hx_ld_f64x2(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    const hx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
    return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}
__device__ __forceinline__ void hx_st_f64x2(__amdgpu_buffer_rsrc_t r, unsigned byte_off, double a, double b)
{
    const hx_u32x4 v = { (unsigned)__double2loint(a), (unsigned)__double2hiint(a), (unsigned)__double2loint(b), (unsigned)__double2hiint(b) };
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, 16);
}

// 8-byte sc1 load through a descriptor (element index in doubles)
__device__ __forceinline__ double hx_ld_f64(__amdgpu_buffer_rsrc_t r, unsigned idx)
{
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(idx * 8u), 0, 16);
    return __hiloint2double((int)v.y, (int)v.x);
}
constexpr unsigned long long kHxWaitTicks = 2000000ull;      // bound of every device-side wait: 20 ms of the 100 MHz clock

// ---- a double handed over as ONE self-announcing 16-byte record (value, tag, check word): one sc1 store of one lane, no
// drain, no flag; the consumer's lane polls the record itself until tag and check word fit (MI355X_MICROARCH.md's
// "handoff-1to1" against "handoff-flag": 0.8 us against 1.3 and more).  A 16-byte store of one lane is one request to one
// cache line; the check word is there so that a torn read could only ever cause another look.  Tags are never 0 and never
// repeat within one run of a window; the records are zeroed when a run starts.
constexpr unsigned kHxTagSalt = 0x9e3779b9u;
__device__ __forceinline__ void hx_st_tagged(__amdgpu_buffer_rsrc_t r, unsigned rec, double v, unsigned tag)
{
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const hx_u32x4 q = { lo, hi, tag, tag ^ lo ^ hi ^ kHxTagSalt };
    __builtin_amdgcn_raw_buffer_store_b128(q, r, (int)(rec * 16u), 0, 16);
}
__device__ __forceinline__ bool hx_ld_tagged(__amdgpu_buffer_rsrc_t r, unsigned rec, unsigned tag, double &v)
{
    const hx_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(rec * 16u), 0, 16);
    v = __hiloint2double((int)q.y, (int)q.x);
    return q.z == tag && q.w == (tag ^ q.x ^ q.y ^ kHxTagSalt);
}

}  // namespace movba
