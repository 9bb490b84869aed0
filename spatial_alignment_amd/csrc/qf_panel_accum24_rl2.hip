// panel_mfma_kernel<24, 1, MODE_ACCUM, 2> (256 < M <= 384): one instantiation per unit, the longest compile of the library
#include "qf_panel_kernel.hpp"

namespace gpsa {
template __global__ void panel_mfma_kernel<24, 1, MODE_ACCUM, 2> GPSA_PANEL_SIG;
}  // namespace gpsa
