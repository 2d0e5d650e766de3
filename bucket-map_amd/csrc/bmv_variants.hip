// bmv_variants.hip -- the group kernel with 5 to 8 words per lane in groups of 8 lanes and more, and the two strip kernels, instantiated in a translation unit of
// their own (declared `extern template` in bmv_api.hip): the alignment kernels in one compiler run took more than five minutes;
// side by side the slowest takes under two.
#include "bmv_kernels.hip.h"

namespace bmv {
// groups of 8 lanes and more: four columns of a traceback cell per lane
template __global__ void bmv_align_kernel<4, 5, false>(Job);
template __global__ void bmv_align_kernel<4, 6, false>(Job);
template __global__ void bmv_align_kernel<4, 7, false>(Job);
template __global__ void bmv_align_kernel<4, 8, false>(Job);
// ... the whole wave, queries in strips
template __global__ void bmv_align_kernel<4, 6, true>(Job);
template __global__ void bmv_align_kernel<4, 8, true>(Job);
}  // namespace bmv
