// bmv_variants.hip -- the alignment kernel's variants with 5 to 8 words per lane and the two strip kernels, instantiated in a
// translation unit of their own: together with the ones bmv_api.hip instantiates they took four minutes in one compiler
// run; side by side they take two.
#include "bmv_kernels.hip.h"

namespace bmv {
template __global__ void bmv_align_kernel<1, 5, false>(Job);
template __global__ void bmv_align_kernel<1, 6, false>(Job);
template __global__ void bmv_align_kernel<1, 7, false>(Job);
template __global__ void bmv_align_kernel<1, 8, false>(Job);
template __global__ void bmv_align_kernel<1, 6, true>(Job);
// groups of 8..15 lanes with several words per lane (1-4 kbp reads): two trace-word pairs per lane in the traceback
template __global__ void bmv_align_kernel<2, 2, false>(Job);
template __global__ void bmv_align_kernel<2, 3, false>(Job);
template __global__ void bmv_align_kernel<2, 4, false>(Job);
template __global__ void bmv_align_kernel<2, 5, false>(Job);
// ... of 4..7 lanes (four pairs per lane) and of 2..3 lanes (eight): reads of a few hundred bases
template __global__ void bmv_align_kernel<4, 2, false>(Job);
template __global__ void bmv_align_kernel<4, 3, false>(Job);
template __global__ void bmv_align_kernel<4, 4, false>(Job);
template __global__ void bmv_align_kernel<8, 2, false>(Job);
template __global__ void bmv_align_kernel<8, 3, false>(Job);
template __global__ void bmv_align_kernel<1, 8, true>(Job);
// one alignment per lane: queries of up to 64 * CW bases
template __global__ void bmv_align_lane_kernel<1>(Job);
template __global__ void bmv_align_lane_kernel<2>(Job);
template __global__ void bmv_align_lane_kernel<3>(Job);
template __global__ void bmv_align_lane_kernel<4>(Job);
template __global__ void bmv_align_lane_kernel<5>(Job);
template __global__ void bmv_align_lane_kernel<6>(Job);
template __global__ void bmv_align_lane_kernel<7>(Job);
template __global__ void bmv_align_lane_kernel<8>(Job);
}  // namespace bmv
