// Pieces shared by the two forms of the direct solver (dense_solve.hip: one launch per block column; dense_persist.hip: one
// launch for the whole solve): tile geometry, the fp64 MFMA tile product and the single-wave panel sweeps.
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "device_types.h"

namespace movba {
namespace dense {

constexpr int NB = kDenseNB;
constexpr int LD = NB + 1;              // LDS row stride: 49 doubles, conflict-free for the MFMA operand reads (rows x k)
constexpr int kStepThreads = 256;
constexpr int kBackThreads = 1024;
constexpr int kBackStepsFrom = 12;      // block columns beyond which the back substitution takes one launch per block row

typedef double dbl4 __attribute__((ext_vector_type(4)));
// the sweeps' progress counter, typed as an LDS word so that its volatile accesses are ds_ instructions (a generic pointer
// makes them flat_ accesses, which count on both memory counters)
typedef __attribute__((address_space(3))) volatile int lds_vint;

// (fast_rcp, device_math.h: the pivot loops below are latency chains with one reciprocal per link)
__device__ __forceinline__ size_t tile_off(int I, int J) { return ((size_t)I * (I + 1) / 2 + J) * (NB * NB); }

// global tile (row-major NB x NB) -> LDS image with row stride LD
__device__ __forceinline__ void load_tile(const double *__restrict__ g, double *sm, int tid)
{
    const double2 *g2 = reinterpret_cast<const double2 *>(g);
    for (int e = tid; e < NB * NB / 2; e += kStepThreads) {
        const double2 v = g2[e];
        const int r = (2 * e) / NB, c = (2 * e) - r * NB;
        sm[r * LD + c] = v.x; sm[r * LD + c + 1] = v.y;
    }
}

// One wave's share of  C -= A B^T  for 48 x 48 tiles: MFMA tile (mt, nt) of 16 x 16, k = 48 in 12 steps of 4.
// Operand maps of v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// C/D: col = lane & 15, row = (lane >> 4) + 4 reg.  A is negated on the way in, so D = C - A B^T.
__device__ __forceinline__ dbl4 tile_mfma(const double *As, const double *Bs, int mt, int nt, int lane, dbl4 c)
{
    const double *ap = As + (mt * 16 + (lane & 15)) * LD + (lane >> 4);
    const double *bp = Bs + (nt * 16 + (lane & 15)) * LD + (lane >> 4);
    double a[NB / 4], b[NB / 4];                    // (operands in registers first: 24 LDS reads in flight, then the chain)
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) { a[q] = -ap[4 * q]; b[q] = bp[4 * q]; }
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], c, 0, 0, 0);
    return c;
}

// The single-wave panel sweeps of k_chol_step (see there), K a compile-time constant so that a panel row stays in
// registers.  DiagSweep<0>::run factors D: pivot K publishes column K of D in T[K][.] (one ds_write, broadcast reads back:
// LDS operations of one wave execute in order, no barrier) and its reciprocal pivot in rinvb[K].  RowSweep<0>::run then
// solves a row of U against it from the same table: u_c -= (u_K / p_KK) T[K][c], no cross-lane traffic at all.
// (Both sweeps in one pass made the compiler sink the whole U chain behind the D chain and keep every broadcast value for
//  it: two thousand spilled registers.  The table in LDS is that hand-off done on purpose.)
template <int K>
struct DiagSweep {
    static __device__ __forceinline__ void run(double (&d)[NB], double *T, double *rinvb, double *pivb, lds_vint *prog, int lane, bool &bad)
    {
        double *cb = T + K * 64;
        cb[lane] = d[K];                            // column K of D, one value per lane
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double piv = cb[K];
        if (!(piv > 0.0) || !isfinite(piv)) { bad = true; piv = 1.0; }
        const double rinv = fast_rcp(piv);
        // column K and its reciprocal pivot are in LDS: the row sweep of the second wave may take pivot K (LDS operations of a
        // wave are performed in order, so whoever sees the counter sees what was written before it)
        if (lane == 0) { pivb[K] = piv; rinvb[K] = rinv; *prog = K + 1; }
        const double td = d[K] * rinv;
#pragma unroll
        for (int c = K + 1; c < NB; ++c) d[c] -= td * cb[c];
        __builtin_amdgcn_sched_barrier(0);
        DiagSweep<K + 1>::run(d, T, rinvb, pivb, prog, lane, bad);
    }
};
template <>
struct DiagSweep<NB> {
    static __device__ __forceinline__ void run(double (&)[NB], double *, double *, double *, lds_vint *, int, bool &) {}
};
template <int K>
struct RowSweep {
    static __device__ __forceinline__ void run(double (&u)[NB], const double *T, const double *rinvb, const lds_vint *prog)
    {
        // wait for pivot K of the factorisation running beside this sweep (bounded: the first wave always gets through
        // its 48 pivots, whatever their values)
        for (int guard = 0; *prog <= K && guard < (1 << 20); ++guard) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // (the table address is made to depend on u[K] through an opaque zero: left alone, the compiler hoists the reads of the
        //  whole table — 1 128 values, none of which depends on the u chain — to the top of the sweep and spills them)
        int z;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z) : "v"(__double2hiint(u[K])));
        const double *cb = T + K * 64 + z;
        const double tu = u[K] * rinvb[K + z];
#pragma unroll
        for (int c = K + 1; c < NB; ++c) u[c] -= tu * cb[c];
        __builtin_amdgcn_sched_barrier(0);
        RowSweep<K + 1>::run(u, T, rinvb, prog);
    }
};
template <>
struct RowSweep<NB> {
    static __device__ __forceinline__ void run(double (&)[NB], const double *, const double *, const lds_vint *) {}
};


}  // namespace dense
}  // namespace movba
