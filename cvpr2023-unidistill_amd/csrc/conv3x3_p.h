// Persistent 32x32x16-MFMA 3x3 convolution (csrc/conv2d_p.hip), called from conv2d.hip's launcher.
#pragma once
#include <hip/hip_runtime.h>

bool ud_conv3x3_p_supported(int B, int H, int W, int Cin, int Cout);
// number of BatchNorm partial slices the kernel writes (work units: 8 rows x 16 pixels)
int ud_conv3x3_p_slices(int B, int H, int W);
int ud_conv3x3_p_launch(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int Cout, const float* bias,
                        const float* scale, const float* shift, const void* residual, int relu, int reverse_taps,
                        float* stats, hipStream_t stream);
