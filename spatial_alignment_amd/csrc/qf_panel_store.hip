// panel_mfma_kernel<MODE_STORE>: explicit instantiations (see qf_panel_kernel.hpp)
#include "qf_panel_kernel.hpp"

namespace gpsa {
GPSA_PANEL_SHAPES(GPSA_PANEL_DEFINE, MODE_STORE)
}  // namespace gpsa
