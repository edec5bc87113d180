// VGG front-end of the Listener (reference src/asr.py:507-558): conv3x3(C_in->64)+ReLU, conv3x3(64->64)+ReLU,
// MaxPool2d(2), conv3x3(64->128)+ReLU, conv3x3(128->128)+ReLU, MaxPool2d(2), and the whole backward.
//
// Layout: activations are channels-last [B][T][F][C] (one (b,t,f) "pixel" = one GEMM row), so a 3x3 tap of a pixel is
// C contiguous floats.  conv2..4 (C_in = 64/128) run as implicit GEMMs (conv.hip): forward with the bias+ReLU
// epilogue, the data gradient as the same kernel on dY with tap-flipped transposed weights and the ReLU mask of the
// layer below as epilogue, the weight gradient as a pixel-split TN product with a gathered operand.  Weights are
// re-ordered once per call from the reference's [C_out][C_in][3][3] to [C_out][tap][C_in] (and [C_in][8-tap][C_out]).
// conv1 (C_in = 1..3, K = 9*C_in <= 27) is too thin for a gathered k-slab: its 32-wide zero-padded patch matrix is
// materialised (R1 x 32 floats) and multiplied on the plain GEMM.  Max-pool keeps a 2-bit argmax per output (first
// maximum in (t,f) scan order, as ATen); its backward is a gather fused with the ReLU mask.
#include "las_common.h"
#include "conv.h"

namespace {

struct VStr { long b, t, f, c; };          // element strides of a logical [B][T][F][C] view

constexpr int C1 = 64, C2 = 128;

// patches[row][tap*C + c] = in[b, t+kt-1, f+kf-1, c]  (0 outside the image; columns >= 9C are zero padding)
template <int VW>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ in, VStr s, int T, int F, int C, int Kp,
                                                     long total, float* __restrict__ col) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int kq = Kp / VW;
    const long row = idx / kq;
    const int kk = (int)(idx - row * kq) * VW;
    const int f = (int)(row % F);
    const long bt = row / F;
    const int t = (int)(bt % T);
    const long b = bt / T;
    const int tap = kk / C, c = kk - tap * C;
    const int kt = tap / 3, kf = tap - kt * 3;
    const int tt = t + kt - 1, ff = f + kf - 1;
    const bool ok = tap < 9 && tt >= 0 && tt < T && ff >= 0 && ff < F;
    const float* src = in + b * s.b + (long)(ok ? tt : t) * s.t + (long)(ok ? ff : f) * s.f + (long)(tap < 9 ? c : 0) * s.c;
    if (VW == 4) {
        float4 v = *(const float4*)src;
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *(float4*)(col + row * Kp + kk) = v;
    } else {
        const float v = *src;
        col[row * Kp + kk] = ok ? v : 0.f;
    }
}

// dx[b,t,f,c] = sum_taps dpatches[(b, t-(kt-1), f-(kf-1))][tap*C + c], optionally masked by relu_out > 0
template <int VW>
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, int T, int F, int C, int Kp,
                                                     long total, const float* __restrict__ relu_out,
                                                     float* __restrict__ dx, VStr so) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int cq = C / VW;
    const long row = idx / cq;
    const int c = (int)(idx - row * cq) * VW;
    const int f = (int)(row % F);
    const long bt = row / F;
    const int t = (int)(bt % T);
    const long b = bt / T;
    float acc[VW];
#pragma unroll
    for (int i = 0; i < VW; ++i) acc[i] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
        for (int kf = 0; kf < 3; ++kf) {
            const int ts = t - (kt - 1), fs = f - (kf - 1);
            const bool ok = ts >= 0 && ts < T && fs >= 0 && fs < F;
            const long r = ok ? row - (long)(kt - 1) * F - (kf - 1) : row;
            const float* p = dcol + r * Kp + (kt * 3 + kf) * C + c;
            if (VW == 4) {
                const float4 v = *(const float4*)p;
                if (ok) { acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w; }
            } else {
                const float v = *p;
                if (ok) acc[0] += v;
            }
        }
    }
    if (relu_out) {
#pragma unroll
        for (int i = 0; i < VW; ++i)
            if (!(relu_out[row * C + c + i] > 0.f)) acc[i] = 0.f;
    }
    float* o = dx + b * so.b + (long)t * so.t + (long)f * so.f + (long)c * so.c;
    if (VW == 4) *(float4*)o = make_float4(acc[0], acc[1], acc[2], acc[3]);
    else *o = acc[0];
}

// 2x2/2 max-pool over (t,f) of a channels-last tensor; floor in f (odd F drops its last column); T is even.
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ y, int T, int F, int C, long total,
                                                       float* __restrict__ out, VStr so, uint8_t* __restrict__ idx) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int T2 = T / 2, F2 = F / 2;
    const int c = (int)(i % C);
    long r = i / C;
    const int f2 = (int)(r % F2);
    r /= F2;
    const int t2 = (int)(r % T2);
    const long b = r / T2;
    const float* p = y + (((b * T + 2 * t2) * F) + 2 * f2) * C + c;
    const float v00 = p[0], v01 = p[C], v10 = p[(long)F * C], v11 = p[(long)F * C + C];
    float best = v00;
    int k = 0;
    if (v01 > best || v01 != v01) { best = v01; k = 1; }
    if (v10 > best || v10 != v10) { best = v10; k = 2; }
    if (v11 > best || v11 != v11) { best = v11; k = 3; }
    idx[i] = (uint8_t)k;
    out[b * so.b + (long)t2 * so.t + (long)f2 * so.f + (long)c * so.c] = best;
}

// d(pre-ReLU conv output)[b,t,f,c] = dpool[b,t/2,f/2,c] if this element won its window and is > 0, else 0
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ dpool, VStr so,
                                                       const uint8_t* __restrict__ idx, const float* __restrict__ y,
                                                       int T, int F, int C, long total, float* __restrict__ dy) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int T2 = T / 2, F2 = F / 2;
    const int c = (int)(i % C);
    long r = i / C;
    const int f = (int)(r % F);
    r /= F;
    const int t = (int)(r % T);
    const long b = r / T;
    const int t2 = t >> 1, f2 = f >> 1;
    float g = 0.f;
    if (f2 < F2 && t2 < T2) {
        const int k = idx[((b * T2 + t2) * F2 + f2) * C + c];
        if (k == (t & 1) * 2 + (f & 1) && y[i] > 0.f)
            g = dpool[b * so.b + (long)t2 * so.t + (long)f2 * so.f + (long)c * so.c];
    }
    dy[i] = g;
}

// Wr[co][tap*C + c] = w[co][c][tap]   (zero in the padding columns)
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, int Co, int C, int Kp,
                                                          float* __restrict__ wr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Co * Kp) return;
    const int co = i / Kp, k = i - co * Kp;
    const int tap = k / C, c = k - tap * C;
    wr[i] = tap < 9 ? w[((long)co * C + c) * 9 + tap] : 0.f;
}

// Wt[c][(8-tap)*Co + co] = w[co][c][tap]   (operand of the data-gradient convolution)
__global__ __launch_bounds__(256) void pack_weight_flipped_kernel(const float* __restrict__ w, int Co, int C,
                                                                  float* __restrict__ wt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Co * C * 9) return;
    const int co = i % Co, tp = (i / Co) % 9, c = i / (9 * Co);
    wt[i] = w[((long)co * C + c) * 9 + (8 - tp)];
}

// gw[co][c][tap] += dWr[co][tap*C + c]
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ dwr, int Co, int C, int Kp,
                                                           float* __restrict__ gw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Co * C * 9) return;
    const int tap = i % 9, c = (i / 9) % C, co = i / (9 * C);
    gw[i] += dwr[(long)co * Kp + tap * C + c];
}

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

int im2col(const float* in, VStr s, int T, int F, int C, int Kp, long rows, float* col, hipStream_t st) {
    if (rows == 0) return LAS_OK;
    if (s.c == 1 && C % 4 == 0 && Kp % 4 == 0 && s.f % 4 == 0 && s.t % 4 == 0 && s.b % 4 == 0 && ((uintptr_t)in & 15) == 0) {
        const long total = rows * (Kp / 4);
        hipLaunchKernelGGL(im2col_kernel<4>, dim3(nblk(total)), dim3(256), 0, st, in, s, T, F, C, Kp, total, col);
    } else {
        const long total = rows * Kp;
        hipLaunchKernelGGL(im2col_kernel<1>, dim3(nblk(total)), dim3(256), 0, st, in, s, T, F, C, Kp, total, col);
    }
    LAS_LAUNCH_OK();
    return LAS_OK;
}

int col2im(const float* dcol, int T, int F, int C, int Kp, long rows, const float* relu_out, float* dx, VStr so,
           hipStream_t st) {
    if (rows == 0) return LAS_OK;
    if (so.c == 1 && C % 4 == 0 && Kp % 4 == 0 && so.f % 4 == 0 && so.t % 4 == 0 && so.b % 4 == 0 && ((uintptr_t)dx & 15) == 0) {
        const long total = rows * (C / 4);
        hipLaunchKernelGGL(col2im_kernel<4>, dim3(nblk(total)), dim3(256), 0, st, dcol, T, F, C, Kp, total, relu_out, dx, so);
    } else {
        const long total = rows * C;
        hipLaunchKernelGGL(col2im_kernel<1>, dim3(nblk(total)), dim3(256), 0, st, dcol, T, F, C, Kp, total, relu_out, dx, so);
    }
    LAS_LAUNCH_OK();
    return LAS_OK;
}

VStr nhwc(int T, int F, int C) { return VStr{(long)T * F * C, (long)F * C, (long)C, 1}; }

struct Geo {
    las_vgg_dims d;
    int cin[4], cout[4], T[4], F[4];
    long rows[4];
    long wr_off[4], wt_off[4];
};

int geometry(int B, int T, int D, Geo* g) {
    const int rc = las_vgg_get_dims(B, T, D, &g->d);
    if (rc != LAS_OK) return rc;
    const las_vgg_dims& d = g->d;
    const int cin[4] = {d.C_in, C1, C1, C2}, cout[4] = {C1, C1, C2, C2};
    long off = 0;
    for (int i = 0; i < 4; ++i) {
        g->cin[i] = cin[i];
        g->cout[i] = cout[i];
        g->T[i] = i < 2 ? d.Tt : d.T2;
        g->F[i] = i < 2 ? d.F : d.F2;
        g->rows[i] = i < 2 ? d.R1 : d.R2;
        g->wr_off[i] = off;
        off += (long)cout[i] * d.Kp[i];
    }
    for (int i = 1; i < 4; ++i) {
        g->wt_off[i] = off;
        off += (long)cin[i] * 9 * cout[i];
    }
    g->wt_off[0] = -1;
    return LAS_OK;
}

}  // namespace

extern "C" int las_vgg_get_dims(int B, int T, int D, las_vgg_dims* d) {
    LAS_CHECK_ARG(d && B >= 1 && T >= 0 && D >= 1);
    if (D % 13 == 0) { d->C_in = D / 13; d->F = 13; }                 // asr.py:522-531 (MFCC first, then fbank)
    else if (D % 40 == 0) { d->C_in = D / 40; d->F = 40; }
    else return LAS_E_BADARG;
    d->Tt = T - T % 4;
    d->T2 = d->Tt / 2; d->F2 = d->F / 2;
    d->T4 = d->T2 / 2; d->F4 = d->F2 / 2;
    d->out_dim = C2 * d->F4;
    d->R1 = (int64_t)B * d->Tt * d->F;
    d->R2 = (int64_t)B * d->T2 * d->F2;
    d->R3 = (int64_t)B * d->T4 * d->F4;
    const int cin[4] = {d->C_in, C1, C1, C2};
    int64_t wr = 0, col = 0;
    for (int i = 0; i < 4; ++i) {
        d->Kp[i] = (9 * cin[i] + 31) / 32 * 32;
        wr += (int64_t)(i < 2 ? C1 : C2) * d->Kp[i] * (i == 0 ? 1 : 2);      // + the flipped copy of conv2..4
    }
    col = d->R1 * d->Kp[0];                                                   // only conv1 materialises patches
    d->wr_floats = wr;
    d->col_floats = col;
    return LAS_OK;
}

extern "C" int las_vgg_fwd(int prec, const float* x, int B, int T, int D, const las_vgg_params* p, las_vgg_state* s,
                           float* out, int time_major, void* stream) {
    LAS_CHECK_ARG(x && p && s && out);
    Geo g;
    int rc = geometry(B, T, D, &g);
    if (rc != LAS_OK) return rc;
    const las_vgg_dims& d = g.d;
    LAS_CHECK_ARG(s->y1 && s->y2 && s->p1 && s->y3 && s->y4 && s->idx1 && s->idx2 && s->col && s->wr);
    for (int i = 0; i < 4; ++i) LAS_CHECK_ARG(p->w[i] && p->b[i]);
    if (d.R3 == 0) return LAS_OK;
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < 4; ++i) {
        hipLaunchKernelGGL(pack_weight_kernel, dim3(nblk((long)g.cout[i] * d.Kp[i])), dim3(256), 0, st, p->w[i],
                           g.cout[i], g.cin[i], d.Kp[i], s->wr + g.wr_off[i]);
        LAS_LAUNCH_OK();
        if (i > 0) {
            hipLaunchKernelGGL(pack_weight_flipped_kernel, dim3(nblk((long)g.cout[i] * g.cin[i] * 9)), dim3(256), 0, st,
                               p->w[i], g.cout[i], g.cin[i], s->wr + g.wt_off[i]);
            LAS_LAUNCH_OK();
        }
    }
    const float* in[4] = {x, s->y1, s->p1, s->y3};
    float* outs[4] = {s->y1, s->y2, s->y3, s->y4};
    for (int i = 0; i < 4; ++i) {
        // the raw features are [B][T][C_in*F] with the delta channel outermost (view_input, asr.py:533-544)
        if (i == 0) {
            const VStr si = VStr{(long)T * D, (long)D, 1, (long)d.F};
            if ((rc = im2col(in[i], si, g.T[i], g.F[i], g.cin[i], d.Kp[i], g.rows[i], s->col, st)) != LAS_OK) return rc;
            rc = las_gemm(prec, 0, 1, (int)g.rows[i], g.cout[i], d.Kp[i], 1.f, s->col, d.Kp[i], 0, s->wr + g.wr_off[i],
                          d.Kp[i], 0, 0.f, outs[i], g.cout[i], 0, p->b[i], LAS_ACT_RELU, 1, stream);
        } else {
            rc = las_conv3x3_fwd(prec, in[i], g.T[i], g.F[i], g.cin[i], g.rows[i], s->wr + g.wr_off[i], g.cout[i], p->b[i],
                                 nullptr, 0, outs[i], st);
        }
        if (rc != LAS_OK) return rc;
        if (i == 1) {
            hipLaunchKernelGGL(pool_fwd_kernel, dim3(nblk(d.R2 * C1)), dim3(256), 0, st, s->y2, d.Tt, d.F, C1, d.R2 * C1,
                               s->p1, nhwc(d.T2, d.F2, C1), s->idx1);
            LAS_LAUNCH_OK();
        }
    }
    // (B,128,T/4,F/4) -> (B,T/4,128*F/4) (asr.py:554-557), written batch- or time-major
    const long od = d.out_dim;
    const VStr so = time_major ? VStr{od, (long)B * od, 1, (long)d.F4} : VStr{(long)d.T4 * od, od, 1, (long)d.F4};
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(nblk(d.R3 * C2)), dim3(256), 0, st, s->y4, d.T2, d.F2, C2, d.R3 * C2, out, so,
                       s->idx2);
    LAS_LAUNCH_OK();
    return LAS_OK;
}

extern "C" int las_vgg_bwd(int prec, const float* x, const float* dout, int B, int T, int D, int time_major,
                           const las_vgg_state* s, const las_vgg_grads* gr, float* dx, void* stream) {
    LAS_CHECK_ARG(x && dout && s && gr);
    Geo g;
    int rc = geometry(B, T, D, &g);
    if (rc != LAS_OK) return rc;
    const las_vgg_dims& d = g.d;
    LAS_CHECK_ARG(s->y1 && s->y2 && s->p1 && s->y3 && s->y4 && s->idx1 && s->idx2 && s->col && s->wr && s->dwr && s->ga && s->gb);
    for (int i = 0; i < 4; ++i) LAS_CHECK_ARG(gr->dw[i] && gr->db[i]);
    hipStream_t st = (hipStream_t)stream;
    if (dx) LAS_HIP(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)B * T * D, st));
    if (d.R3 == 0) return LAS_OK;
    const long od = d.out_dim;
    const VStr so = time_major ? VStr{od, (long)B * od, 1, (long)d.F4} : VStr{(long)d.T4 * od, od, 1, (long)d.F4};
    // pool2 backward (+ ReLU mask of conv4's output) -> d pre-activation of conv4
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(nblk(d.R2 * C2)), dim3(256), 0, st, dout, so, s->idx2, s->y4, d.T2, d.F2, C2,
                       d.R2 * C2, s->ga);
    LAS_LAUNCH_OK();
    const float* in[4] = {x, s->y1, s->p1, s->y3};
    float* dy = s->ga;
    float* other = s->gb;
    for (int i = 3; i >= 0; --i) {
        const int Co = g.cout[i], Ci = g.cin[i], Kp = d.Kp[i];
        const long R = g.rows[i];
        if ((rc = las_colsum(dy, Co, (int)R, Co, 1.f, gr->db[i], stream)) != LAS_OK) return rc;
        if (i == 0) {
            const VStr si = VStr{(long)T * D, (long)D, 1, (long)d.F};
            if ((rc = im2col(in[0], si, g.T[0], g.F[0], Ci, Kp, R, s->col, st)) != LAS_OK) return rc;
            rc = las_gemm(prec, 1, 0, Co, Kp, (int)R, 1.f, dy, Co, 0, s->col, Kp, 0, 0.f, s->dwr, Kp, 0, nullptr,
                          LAS_ACT_NONE, 1, stream);
        } else {
            rc = las_conv3x3_wgrad(prec, dy, Co, in[i], g.T[i], g.F[i], Ci, R, s->dwr, st);
        }
        if (rc != LAS_OK) return rc;
        hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(nblk((long)Co * Ci * 9)), dim3(256), 0, st, s->dwr, Co, Ci, Kp, gr->dw[i]);
        LAS_LAUNCH_OK();
        if (i == 0) {
            if (!dx) break;
            rc = las_gemm(prec, 0, 0, (int)R, Kp, Co, 1.f, dy, Co, 0, s->wr + g.wr_off[0], Kp, 0, 0.f, s->col, Kp, 0, nullptr,
                          LAS_ACT_NONE, 1, stream);
            if (rc != LAS_OK) return rc;
            rc = col2im(s->col, g.T[0], g.F[0], Ci, Kp, R, nullptr, dx, VStr{(long)T * D, (long)D, 1, (long)d.F}, st);
            if (rc != LAS_OK) return rc;
            break;
        }
        // data gradient: the same convolution on dY with the flipped weights; conv4 / conv2 inputs are ReLU outputs of
        // the layer below (mask fused), conv3's input is pool1's output (routed through pool1 and conv2's ReLU next)
        rc = las_conv3x3_fwd(prec, dy, g.T[i], g.F[i], Co, R, s->wr + g.wt_off[i], Ci, nullptr, i == 2 ? nullptr : in[i], 1,
                             other, st);
        if (rc != LAS_OK) return rc;
        if (i == 2) {
            hipLaunchKernelGGL(pool_bwd_kernel, dim3(nblk(d.R1 * C1)), dim3(256), 0, st, other, nhwc(d.T2, d.F2, C1), s->idx1,
                               s->y2, d.Tt, d.F, C1, d.R1 * C1, dy);
            LAS_LAUNCH_OK();
            continue;                                   // dy now holds d pre-activation of conv2
        }
        float* tmp = dy; dy = other; other = tmp;
    }
    return LAS_OK;
}
