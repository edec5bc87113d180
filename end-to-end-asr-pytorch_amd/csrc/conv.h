// Internal (C++ linkage) entry points of the implicit-GEMM 3x3 convolution kernels in conv.hip, used by vgg.hip.
#pragma once
#include "las_common.h"

// out[P][N] = epi( conv3x3_pad1(in [.,T,F,C] channels-last, w [N][9C] tap-major) ); N in {64,128}, C % 32 == 0.
// epi 0: + bias, ReLU.  epi 1: zero where mask[P][N] <= 0 (mask may be NULL).
int las_conv3x3_fwd(int prec, const float* in, int T, int F, int C, long P, const float* w, int N, const float* bias,
                    const float* mask, int epi, float* out, hipStream_t st);
// dwr[Co][9C] = sum_p dy[p][co] * in[p + off(tap)][c]   (overwrites dwr); Co in {64,128}, C % 32 == 0.
int las_conv3x3_wgrad(int prec, const float* dy, int Co, const float* in, int T, int F, int C, long P, float* dwr,
                      hipStream_t st);
