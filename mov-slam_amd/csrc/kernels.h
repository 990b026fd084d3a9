// Launch wrappers of the HIP kernels in kernels.hip (one stream, no host syncs inside).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace movba {

constexpr int kBandThreads = 512;       // k_band: one workgroup of 8 waves
constexpr int kPcgRowsThreads = 512;    // 8 waves: 2 per SIMD, 256 VGPRs per lane
constexpr int kPcgPlanWaves = kPcgRowsThreads / 64, kPcgPlanOwnBatch = 10;     // (pcg_plan.cpp / pcg_kernel.hip)

hipError_t configure_kernels(int unused);
hipError_t configure_pcg_rows();
hipError_t configure_struct_kernels();
hipError_t launch_struct_count(const StructDev &sd, hipStream_t s);
hipError_t launch_struct_scan(const StructDev &sd, hipStream_t s);
hipError_t launch_struct_counts_out(const StructDev &sd, int32_t *host_cnt_dev, int seq, hipStream_t s, const int32_t *basic_pe = nullptr, const int32_t *basic_info = nullptr);
// arrays in mapped host memory -> arena, read by the kernel itself (IngestArgs); every workgroup raises a.counter once
hipError_t launch_ingest(const IngestArgs &a, hipStream_t s);
int ingest_workgroups();
// the grouping pass on the device (BasicDev): k_basic_hist, then k_basic_scan (whose last workgroup numbers the keyframes)
hipError_t launch_basic(const BasicDev &bd, hipStream_t s);
bool struct_lds_fits(int nfree, int NP);
hipError_t launch_struct_fill(const StructDev &sd, hipStream_t s);
// structure pass beyond k_struct_pairs' reach (struct_sort.hip): counts by atomics, fill by a stable sort of the couples
hipError_t launch_couple_count(const StructDev &sd, int32_t *cnt_pt, hipStream_t s);
size_t sorted_fill_temp_bytes(int P, long long noff, int nfree);
hipError_t launch_sorted_fill(const StructDev &sd, const int32_t *cnt_pt, int32_t *off, unsigned *keys_in, unsigned *keys_out,
                              unsigned long long *vals_in, void *tmp, size_t tmp_bytes, long long noff, hipStream_t s);
hipError_t launch_slot_point(int32_t *slot, const int32_t *g_pose, const int32_t *base, const int32_t *g_point, int32_t *slot_point, int E, const int32_t *hx, int NP, hipStream_t s);
bool pcg_rows_supported(int nfree, const int32_t *row_ptr, PcgParams *pp);
hipError_t launch_pcg_rows(const DevWindow &w, int nrowent, const PcgParams &pp, int trial, hipStream_t s);

hipError_t launch_init(const DevWindow &w, hipStream_t s);
hipError_t launch_linearize(const DevWindow &w, hipStream_t s);
hipError_t launch_schur(const DevWindow &w, int mode, hipStream_t s);
hipError_t launch_lambda_init(const DevWindow &w, hipStream_t s);
hipError_t launch_backsub(const DevWindow &w, hipStream_t s);
hipError_t launch_finalize(const DevWindow &w, hipStream_t s);
// destinations of k_export in the host's pinned staging buffer (device view; null = not wanted), as 64-bit words
struct ExportDst { unsigned long long *poses, *points, *chi2, *outlier; };
hipError_t launch_export(const DevWindow &w, const ExportDst &d, hipStream_t s);

// batched launches over the concatenated windows of a BatchDev (movba_lba_run_batch)
hipError_t launch_init_batch(const BatchDev &b, int nblk, hipStream_t s);
hipError_t launch_point_batch(const BatchDev &b, int nblk, bool backsub, bool stereo, bool ldsp, size_t lds, hipStream_t s);
size_t point_lds_bytes_for(const DevWindow &w, bool backsub, bool ldsp);
int schur_blocks(const DevWindow &w);
hipError_t launch_schur_batch(const BatchDev &b, int nblk, int mode, bool stereo, hipStream_t s);
hipError_t launch_lambda_init_batch(const BatchDev &b, hipStream_t s);
hipError_t launch_finalize_batch(const BatchDev &b, int nblk, hipStream_t s);
hipError_t launch_pcg_rows_batch(const BatchDev &b, bool overflow, bool padded, size_t lds, int trial, hipStream_t s);
size_t pcg_rows_lds_bytes(int nfree, int nrowent, bool padded);
// single-workgroup banded factorisation (band_kernel.hip): LDS of a window of nfree keyframes whose reduced matrix has half
// bandwidth bw (in blocks); band_supported: the band fits one workgroup's LDS
size_t band_lds_bytes(int nfree, int bw);
bool band_supported(int nfree, int bw);
hipError_t launch_band(const DevWindow &w, int bw, hipStream_t s);
hipError_t launch_band_batch(const BatchDev &b, size_t lds, hipStream_t s);
hipError_t configure_band();

// direct solver (dense_solve.hip): assemble + one launch per block column + back substitution / pose update
hipError_t configure_dense_kernels();
hipError_t launch_dense_solve(const DevWindow &w, hipStream_t s);
// ... and in one launch (dense_persist.hip) when the static schedule fits (dense_plan.h)
hipError_t configure_dense_persist();
bool dense_persist_supported(const DensePlan &p);
hipError_t launch_dense_persist(const DevWindow &w, unsigned epoch, hipStream_t s);
size_t dense_tiles_doubles(int nfree);
int dense_ntile(int nfree);
// the point kernels stage every keyframe's rotation in LDS up to this many bytes; larger windows read them through L2
constexpr size_t kPointLdsLimit = 150 * 1024;
size_t point_lds_need(int NP, int nfree);

}  // namespace movba
