// Launch wrappers of the HIP kernels in kernels.hip (one stream, no host syncs inside).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace movba {

hipError_t configure_kernels(int unused);
size_t pcg_lds_bytes(int nfree);

hipError_t launch_init(const DevWindow &w, hipStream_t s);
hipError_t launch_linearize(const DevWindow &w, hipStream_t s);
hipError_t launch_schur(const DevWindow &w, int mode, hipStream_t s);
hipError_t launch_lambda_init(const DevWindow &w, hipStream_t s);
hipError_t launch_pcg(const DevWindow &w, const PcgParams &pp, hipStream_t s);
hipError_t launch_backsub(const DevWindow &w, hipStream_t s);
hipError_t launch_decide(const DevWindow &w, hipStream_t s);
hipError_t launch_finalize(const DevWindow &w, hipStream_t s);

}  // namespace movba
