// bmv_variants2.hip -- the group kernel with 4 to 8 words per lane in groups of 2..7 lanes, instantiated in a translation unit of
// their own (declared `extern template` in bmv_api.hip): the alignment kernels in one compiler run took more than five minutes;
// side by side the slowest takes under two.
#include "bmv_kernels.hip.h"

namespace bmv {
// groups of 2..7 lanes: eight columns per lane (two lanes to a cell)
template __global__ void bmv_align_kernel<8, 4, false>(Job);
template __global__ void bmv_align_kernel<8, 5, false>(Job);
template __global__ void bmv_align_kernel<8, 6, false>(Job);
template __global__ void bmv_align_kernel<8, 7, false>(Job);
template __global__ void bmv_align_kernel<8, 8, false>(Job);
}  // namespace bmv
