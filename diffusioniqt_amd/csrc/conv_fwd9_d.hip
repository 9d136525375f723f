// conv_fwd9_kernel with the GroupNorm-backward epilogue, 512-voxel 3x3x3 variant (see conv_fwd9.hip)
#include "conv_fwd9_kernel.h"

namespace diqt {

int fwd9_launch_d(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream) {
    if (g.variant == 0) return f9_launch<F9_333_512, true>(x, packed, bias, residual, y, g, lds, grid, stream);
    set_error("conv3d_fwd(v9, GroupNorm-backward epilogue): no variant %d in this unit", g.variant);
    return DIQT_E_UNSUPPORTED;
}

}  // namespace diqt
