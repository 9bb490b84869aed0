// panel_mfma_kernel<MODE_ACCUM>, M <= 256: explicit instantiations (see qf_panel_kernel.hpp)
#include "qf_panel_kernel.hpp"

namespace gpsa {
GPSA_PANEL_SHAPES(GPSA_PANEL_DEFINE, MODE_ACCUM)
}  // namespace gpsa
