// bmv_variants3.hip -- the lane-per-alignment kernels, instantiated in a translation unit of
// their own (declared `extern template` in bmv_api.hip): the alignment kernels in one compiler run took more than five minutes;
// side by side the slowest takes under two.
#include "bmv_kernels.hip.h"

namespace bmv {
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
