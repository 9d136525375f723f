// conv_f9h_kernel: the 256-voxel 3x3x3 instantiations (the 16^3 level of the C2 / C4 U-Nets under autocast).
#include "conv_f9h_kernel.h"

namespace diqt {
namespace h9 {

int launch_b(const void* x, const unsigned short* wp, const float* bias, const float* residual, void* y, const H9Geom& g, size_t lds,
             unsigned grid, int bf16, bool yHalf, void* stream) {
    return launch_cfg<H9_333_256>(x, wp, bias, residual, y, g, lds, grid, bf16, yHalf, stream);
}

}  // namespace h9
}  // namespace diqt
