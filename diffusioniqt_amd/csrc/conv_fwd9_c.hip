// conv_fwd9_kernel, (3,1,1) variants and the 256-voxel 3x3x3 variant (see conv_fwd9.hip)
#include "conv_fwd9_kernel.h"

namespace diqt {

int fwd9_launch_c(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream) {
    switch (g.variant) {
        case 1: return f9_launch<F9_333_256>(x, packed, bias, residual, y, g, lds, grid, stream);
        case 5: return f9_launch<F9_311_512>(x, packed, bias, residual, y, g, lds, grid, stream);
        case 6: return f9_launch<F9_311_256>(x, packed, bias, residual, y, g, lds, grid, stream);
    }
    set_error("conv3d_fwd(v9): no variant %d in this unit", g.variant);
    return DIQT_E_UNSUPPORTED;
}

}  // namespace diqt
