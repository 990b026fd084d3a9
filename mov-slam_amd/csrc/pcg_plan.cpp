// Host side of the k_pcg_rows plan: LDS size of the launch and the deal of block rows to the workgroup's waves (pure C++: shared
// by the product build and by the CPU-only concurrency tests, tests/hipstub).
#include "kernels.h"

namespace movba {

namespace {
constexpr int kNW = kPcgPlanWaves, kNC = kCoarseDim, kOwnBatch = kPcgPlanOwnBatch;
}

// `padded`: the mat-vec's pair sums by row in kOwnBatch zero-padded slots (+ a strip for the lanes without a pair and ten zeros)
size_t pcg_rows_lds_bytes(int nfree, int nrowent, bool padded)
{
    const size_t n = 6 * (size_t)nfree, npad = (n + 1) & ~(size_t)1;
    const size_t ypart = padded ? n * kOwnBatch + 5 * kOwnBatch + 6 + kOwnBatch : 6 * ((size_t)nrowent / 2 + 1 + kOwnBatch);
    const size_t solve = (2 * npad + 36 * (size_t)nfree + 2 * kNW + 2 + kNC * kNC / 2 + kNC + 32 * kNW + ypart + 36 * (size_t)nfree + 64 * kNW) * sizeof(double);
    const size_t coarse = ((size_t)kNC * kNC + 9 * kNC + 8) * sizeof(double) + 2 * sizeof(int32_t) * ((size_t)nrowent + 2);   // the second workgroup (coarse_level.h)
    return solve > coarse ? solve : coarse;
}

// k_band's LDS carve (band_kernel.hip): band, two vectors, the step's panel, a strip, the enumeration of the trailing blocks, two words
size_t band_lds_bytes(int nfree, int bw)
{
    const size_t n = 6 * (size_t)nfree, npad = (n + 1) & ~(size_t)1;
    const size_t ntri = ((size_t)bw * (bw + 1) / 2 + 1) & ~(size_t)1;
    return ((size_t)nfree * (bw + 1) * 36 + 2 * npad + (size_t)bw * 36 + 8 + 12) * sizeof(double) + (ntri + 4) * sizeof(int32_t);
}

// (8 waves x 58 rows of the sweep: 6 bw + 1 <= 464)
bool band_supported(int nfree, int bw) { return nfree >= 1 && bw >= 0 && bw <= 77 && band_lds_bytes(nfree, bw) <= 159 * 1024; }

// Deals block rows to the waves so that every wave gets about the same number of gather-list
// entries (the mat-vec work) and at most 10 block rows (60 owner lanes).  A wave holds 64 entry
// pairs in VGPRs; anything beyond that is flagged as overflow (read from an L2 copy of S).
bool pcg_rows_supported(int nfree, const int32_t *row_ptr, PcgParams *pp)
{
    if (nfree <= 0 || nfree > 10 * kNW) return false;
    const int nrowent = row_ptr[nfree];
    int maxrow = 0;
    for (int b = 0; b < nfree; ++b) maxrow = row_ptr[b + 1] - row_ptr[b] > maxrow ? row_ptr[b + 1] - row_ptr[b] : maxrow;
    pp->padded = 0;
    if (pcg_rows_lds_bytes(nfree, nrowent, false) > 159 * 1024) return false;
    const bool pad_fits = maxrow <= 2 * kOwnBatch && pcg_rows_lds_bytes(nfree, nrowent, true) <= 159 * 1024;
    // smallest per-wave entry budget for which a greedy fill (<= 10 rows per wave) needs <= kNW waves
    auto fill = [&](int cap, int32_t *out) {
        int b = 0, wv = 0;
        for (; wv < kNW && b < nfree; ++wv) {
            int e = b + 1;                                      // a wave always takes at least one row
            while (e < nfree && e - b < 10 && row_ptr[e + 1] - row_ptr[b] <= cap) ++e;
            if (out) out[wv + 1] = e;
            b = e;
        }
        if (out) for (; wv < kNW; ++wv) out[wv + 1] = nfree;
        return b == nfree;
    };
    if (nfree <= kNW) {
        // one keyframe per wave: the aggregates are single keyframes and the coarse level is the exact inverse (fresh mode)
        for (int wv = 0; wv <= kNW; ++wv) pp->wave_row0[wv] = wv < nfree ? wv : nfree;
        pp->overflow = 0;
        pp->padded = pad_fits ? 1 : 0;
        return true;
    }
    int lo = 1, hi = nrowent > 1 ? nrowent : 1;
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (fill(mid, nullptr)) hi = mid; else lo = mid + 1;
    }
    pp->wave_row0[0] = 0;
    if (!fill(lo, pp->wave_row0)) return false;
    pp->overflow = 0;
    for (int wv = 0; wv < kNW; ++wv)
        if (row_ptr[pp->wave_row0[wv + 1]] - row_ptr[pp->wave_row0[wv]] > 128) pp->overflow = 1;
    pp->padded = (pad_fits && !pp->overflow) ? 1 : 0;
    return true;
}

}  // namespace movba
