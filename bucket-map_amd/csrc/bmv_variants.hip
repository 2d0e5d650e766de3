// bmv_variants.hip -- the alignment kernel's variants with 5 to 8 words per lane, the two strip kernels and the lane-per-alignment kernels, instantiated in a
// translation unit of their own: together with the ones bmv_api.hip instantiates they took four minutes in one compiler
// run; side by side they take two.
#include "bmv_kernels.hip.h"

namespace bmv {
// groups of 4 lanes and more: four columns of a traceback cell per lane
template __global__ void bmv_align_kernel<4, 5, false>(Job);
template __global__ void bmv_align_kernel<4, 6, false>(Job);
template __global__ void bmv_align_kernel<4, 7, false>(Job);
template __global__ void bmv_align_kernel<4, 8, false>(Job);
// groups of 2..7 lanes: eight columns per lane (two lanes to a cell)
template __global__ void bmv_align_kernel<8, 4, false>(Job);
template __global__ void bmv_align_kernel<8, 5, false>(Job);
template __global__ void bmv_align_kernel<8, 6, false>(Job);
template __global__ void bmv_align_kernel<8, 7, false>(Job);
template __global__ void bmv_align_kernel<8, 8, false>(Job);
// ... the whole wave, queries in strips
template __global__ void bmv_align_kernel<4, 6, true>(Job);
template __global__ void bmv_align_kernel<4, 8, true>(Job);
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
