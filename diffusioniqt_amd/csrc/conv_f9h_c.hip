// conv_f9h_kernel: the per-frame (1,3,3) instantiations (pseudo-3D blocks of Unet3D, imagen_video.py:352-381).
#include "conv_f9h_kernel.h"

namespace diqt {
namespace h9 {

int launch_c(const void* x, const unsigned short* wp, const float* bias, const float* residual, void* y, const H9Geom& g, size_t lds,
             unsigned grid, int bf16, bool yHalf, void* stream) {
    switch (g.variant) {
        case 2: return launch_cfg<H9_133_A>(x, wp, bias, residual, y, g, lds, grid, bf16, yHalf, stream);
        case 3: return launch_cfg<H9_133_B>(x, wp, bias, residual, y, g, lds, grid, bf16, yHalf, stream);
        case 5: return launch_cfg<H9_133_D>(x, wp, bias, residual, y, g, lds, grid, bf16, yHalf, stream);
        default: return launch_cfg<H9_133_C>(x, wp, bias, residual, y, g, lds, grid, bf16, yHalf, stream);
    }
}

}  // namespace h9
}  // namespace diqt
