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
template __global__ void bmv_align_kernel<1, 8, true>(Job);
}  // namespace bmv
