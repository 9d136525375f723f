// conv_fwd9_kernel, (1,3,3) variants (see conv_fwd9.hip)
#include "conv_fwd9_kernel.h"

namespace diqt {

int fwd9_launch_b(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream) {
    switch (g.variant) {
        case 2: return f9_launch<F9_133_A>(x, packed, bias, residual, y, g, lds, grid, stream);
        case 3: return f9_launch<F9_133_B>(x, packed, bias, residual, y, g, lds, grid, stream);
        case 4: return f9_launch<F9_133_C>(x, packed, bias, residual, y, g, lds, grid, stream);
    }
    set_error("conv3d_fwd(v9): no (1,3,3) variant %d", g.variant);
    return DIQT_E_UNSUPPORTED;
}

}  // namespace diqt
