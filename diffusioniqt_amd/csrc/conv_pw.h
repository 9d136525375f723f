// Internal interface of the K-resident pointwise forward kernel (conv_pw.hip), called from conv3d_fwd_one (conv_mfma.hip).
#pragma once
#include <stddef.h>

namespace diqt {

// Does conv1x1_k64_kernel take this product (y[rows][Cout] = x[rows][64] W^T, Cout >= 256, tensors < 2 GiB, aligned)?
bool pw64_ok(long long rows, int Cin, int Cout, const void* x, const void* packed, const void* y);
int pw64_launch(const float* x, const float* packed, const float* bias, const float* residual, float* y, long long rows, int Cout,
                int CoutPad, void* stream);

}  // namespace diqt
