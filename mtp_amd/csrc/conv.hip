// The layers around the DCNv3 core in InternImage (reference Multi-Task_Pretrain/backbone/intern_image.py "II",
// ops_dcnv3/modules/dcnv3.py "DCNM"), channels-last, gfx950.  All HBM-bound element / gather kernels; the contractions
// themselves (3x3 convolutions as im2col + GEMM, the Linear layers) run on the MFMA GEMMs of gemm*.hip.
//   im2col3x3 / col2im3x3    StemLayer II:239-276 and DownsampleLayer II:279-300: Conv2d(k=3, s=2, p=1)
//   conv3x3_pack / _unpack   (Cout, Cin, 3, 3) f32 master weight <-> the GEMM's [Cout][(kh, kw, c)] images
//   dwconv3x3 fwd / dx / dw  the depth-wise 3x3 of DCNM:262-272
//   softmax_groups           softmax over the P sampling points of each group, DCNM:341-342
//   scale_residual           x + drop_path(gamma * z), II:424-426 (layer scale + post-norm branch)
//   pack_rows_padded         Linear weights whose row count is not a multiple of 8 (mask head, 9 * 12 = 108 rows)
#include "common.h"

namespace {

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return bf16_bits_to_f32(p->bits); }

// ---------------------------------------------------------------------------------------------------------------------
// cols[(n, ho, wo)][(kh * 3 + kw) * Cin + c] = x[n][ho * s + kh - 1][wo * s + kw - 1][c] (0 outside, 0 for columns >= 9 Cin).
// The source is addressed through element strides, so the NCHW f32 image (stem) and channels-last maps both fit.
template <typename Tx, typename Tc>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const Tx* __restrict__ x, int64_t sN, int64_t sH, int64_t sW, int64_t sC, Tc* __restrict__ cols,
                                                       int H, int W, int Cin, int Ho, int Wo, int stride, int Kp, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int k = (int)(idx % Kp);
    const int64_t pix = idx / Kp;
    const int wo = (int)(pix % Wo), ho = (int)((pix / Wo) % Ho);
    const int64_t n = pix / ((int64_t)Wo * Ho);
    float v = 0.f;
    if (k < 9 * Cin) {
        const int tap = k / Cin, c = k - tap * Cin, kh = tap / 3, kw = tap - 3 * kh;
        const int h = ho * stride + kh - 1, w = wo * stride + kw - 1;
        if (h >= 0 && h < H && w >= 0 && w < W) v = ld1(x + n * sN + h * sH + w * sW + c * sC);
    }
    Elem<Tc>::store(cols + idx, v);
}

// dx[n][h][w][c] (+)= sum over the taps (kh, kw) whose output position (ho, wo) exists: dcols[(n, ho, wo)][(kh * 3 + kw) * Cin + c]
template <typename Tc>
__global__ __launch_bounds__(256) void col2im3x3_kernel(const Tc* __restrict__ dcols, float* __restrict__ dx, int64_t sN, int64_t sH, int64_t sW, int64_t sC,
                                                       int H, int W, int Cin, int Ho, int Wo, int stride, int Kp, int accumulate, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % Cin);
    const int64_t pix = idx / Cin;
    const int w = (int)(pix % W), h = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    float s = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hn = h + 1 - kh;
        if (hn < 0 || hn % stride) continue;
        const int ho = hn / stride;
        if (ho >= Ho) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wn = w + 1 - kw;
            if (wn < 0 || wn % stride) continue;
            const int wo = wn / stride;
            if (wo >= Wo) continue;
            s += ld1(dcols + ((n * Ho + ho) * Wo + wo) * Kp + (kh * 3 + kw) * Cin + c);
        }
    }
    float* p = dx + n * sN + h * sH + w * sW + c * sC;
    *p = accumulate ? *p + s : s;
}

// The same two gathers with 8 channels per lane (round 5): channels-last bf16 maps with Cin % 8 == 0 (the downsample convolutions and the stem's second one) -- one 16-byte load
// and one 16-byte store per lane instead of eight 2-byte pairs with their own index arithmetic (the element-wise kernels ran at 1.1 TB/s: 0.2 ms per call at InternImage-XL's stem).
__global__ __launch_bounds__(256) void im2col3x3_v8_kernel(const bf16_t* __restrict__ x, int64_t sN, int64_t sH, int64_t sW, bf16_t* __restrict__ cols,
                                                          int H, int W, int Cin, int Ho, int Wo, int stride, int Kp, int64_t total8) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total8) return;
    const int K8 = Kp >> 3;
    const int k = (int)(idx % K8) << 3;
    const int64_t pix = idx / K8;
    const int wo = (int)(pix % Wo), ho = (int)((pix / Wo) % Ho);
    const int64_t n = pix / ((int64_t)Wo * Ho);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (k < 9 * Cin) {
        const int tap = k / Cin, c = k - tap * Cin, kh = tap / 3, kw = tap - 3 * kh;
        const int h = ho * stride + kh - 1, w = wo * stride + kw - 1;
        if (h >= 0 && h < H && w >= 0 && w < W) v = *reinterpret_cast<const uint4*>(x + n * sN + h * sH + w * sW + c);
    }
    *reinterpret_cast<uint4*>(cols + pix * Kp + k) = v;
}
__global__ __launch_bounds__(256) void col2im3x3_v8_kernel(const bf16_t* __restrict__ dcols, float* __restrict__ dx, int64_t sN, int64_t sH, int64_t sW,
                                                          int H, int W, int Cin, int Ho, int Wo, int stride, int Kp, int accumulate, int64_t total8) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total8) return;
    const int C8 = Cin >> 3;
    const int c = (int)(idx % C8) << 3;
    const int64_t pix = idx / C8;
    const int w = (int)(pix % W), h = (int)((pix / W) % H);
    const int64_t n = pix / ((int64_t)W * H);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hn = h + 1 - kh;
        if (hn < 0 || hn % stride) continue;
        const int ho = hn / stride;
        if (ho >= Ho) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int wn = w + 1 - kw;
            if (wn < 0 || wn % stride) continue;
            const int wo = wn / stride;
            if (wo >= Wo) continue;
            float t[8];
            load8(dcols + ((n * Ho + ho) * Wo + wo) * Kp + (kh * 3 + kw) * Cin + c, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += t[e];     // same tap order as the element-wise kernel: bit-identical sums
        }
    }
    float* p = dx + n * sN + h * sH + w * sW + c;
    if (accumulate) {
        float o[8];
        load8(p, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = o[e] + s[e];
    }
    store8(p, s);
}

// w2[o][(kh * 3 + kw) * Cin + c] = w[o][c][kh][kw], zero beyond 9 Cin; w2t = its transpose [Kp][Cout]
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_pack_kernel(const float* __restrict__ w, T* __restrict__ w2, T* __restrict__ w2t, int Cout, int Cin, int Kp) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)Cout * Kp) return;
    const int k = (int)(idx % Kp), o = (int)(idx / Kp);
    float v = 0.f;
    if (k < 9 * Cin) {
        const int tap = k / Cin, c = k - tap * Cin;
        v = w[((int64_t)o * Cin + c) * 9 + tap];
    }
    if (w2) Elem<T>::store(w2 + idx, v);
    if (w2t) Elem<T>::store(w2t + (int64_t)k * Cout + o, v);
}
__global__ __launch_bounds__(256) void conv3x3_unpack_kernel(const float* __restrict__ dw2, float* __restrict__ dw, int Cout, int Cin, int Kp) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)Cout * Cin * 9) return;
    const int tap = (int)(idx % 9), c = (int)((idx / 9) % Cin), o = (int)(idx / (9 * (int64_t)Cin));
    dw[idx] = dw2[(int64_t)o * Kp + tap * Cin + c];
}

// w (R, C) f32 -> wp [Rp][C] (rows >= R zero) and wpt [C][Rp] (columns >= R zero)
template <typename T>
__global__ __launch_bounds__(256) void pack_rows_padded_kernel(const float* __restrict__ w, T* __restrict__ wp, T* __restrict__ wpt, int R, int C, int Rp) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)Rp * C) return;
    const int c = (int)(idx % C), r = (int)(idx / C);
    const float v = r < R ? w[(int64_t)r * C + c] : 0.f;
    if (wp) Elem<T>::store(wp + idx, v);
    if (wpt) Elem<T>::store(wpt + (int64_t)c * Rp + r, v);
}

// ---------------------------------------------------------------------------------------------------------------------
// depth-wise 3x3, stride 1, padding 1, channels-last; thread = (pixel, 4 channels).  FLIP: the data gradient (the same sum with the
// taps mirrored, no bias), written to an f32 map, optionally accumulating.
template <typename T, typename To, bool FLIP>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, To* __restrict__ y,
                                                       int H, int W, int C, int accumulate, int64_t total) {
    // 32-bit index arithmetic (the host checks total < 2^31): five 64-bit divisions were ~600 of this thread's ~700 instructions
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (uint32_t)total) return;
    const uint32_t c4 = (uint32_t)C >> 2;
    const uint32_t pix32 = idx / c4;
    const int c = 4 * (int)(idx - pix32 * c4);
    const uint32_t row = pix32 / (uint32_t)W, n32 = row / (uint32_t)H;
    const int wx = (int)(pix32 - row * (uint32_t)W), h = (int)(row - n32 * (uint32_t)H);
    const int64_t pix = pix32, n = n32;
    float4 acc = (bias && !FLIP) ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int hh = h + kh - 1;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ww = wx + kw - 1;
            const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
            const float4 v = load4(x + ((n * H + (ok ? hh : h)) * W + (ok ? ww : wx)) * C + c);      // unconditional load on a clamped address
            const int tap = FLIP ? (2 - kh) * 3 + (2 - kw) : kh * 3 + kw;
            const float m = ok ? 1.f : 0.f;
            acc.x += m * v.x * w[(c + 0) * 9 + tap];
            acc.y += m * v.y * w[(c + 1) * 9 + tap];
            acc.z += m * v.z * w[(c + 2) * 9 + tap];
            acc.w += m * v.w * w[(c + 3) * 9 + tap];
        }
    }
    To* o = y + pix * C + c;
    if (accumulate) {
        const float4 p = load4(o);
        acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    store4(o, acc);
}

// bf16 input, 8 consecutive pixels of a row per lane (round 5; W % 8 == 0): the 3 x 10 input window is loaded once (30 eight-byte loads, all in flight) and the lane's 36 weights
// as nine float4, then the ten columns are unpacked once each and accumulated into the outputs they touch -- 8.25 loads per output quad where dwconv3x3_kernel issues 45.
// (Different summation order from the one-pixel kernel: column-major over the window instead of tap-major; both are exact to f32 rounding.)
template <typename To, bool FLIP>
__global__ __launch_bounds__(256) void dwconv3x3_p8_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, To* __restrict__ y,
                                                          int H, int W, int C, int accumulate, int64_t total) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (uint32_t)total) return;
    const uint32_t c4 = (uint32_t)C >> 2, W8 = (uint32_t)W >> 3;
    const uint32_t grp = idx / c4;
    const int c = 4 * (int)(idx - grp * c4);
    const uint32_t row = grp / W8, n32 = row / (uint32_t)H;
    const int wx0 = 8 * (int)(grp - row * W8), h = (int)(row - n32 * (uint32_t)H);
    const int64_t n = n32;
    float wf[36];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(w + (int64_t)c * 9 + 4 * j);      // (c % 4 == 0: 16-byte aligned)
        wf[4 * j] = t.x; wf[4 * j + 1] = t.y; wf[4 * j + 2] = t.z; wf[4 * j + 3] = t.w;
    }
    uint2 raw[3][10];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int hh = h + r - 1;
        const bool rok = hh >= 0 && hh < H;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const int ww = wx0 + j - 1;
            const bool ok = rok && ww >= 0 && ww < W;
            const uint2 t = *reinterpret_cast<const uint2*>(x + ((n * H + (ok ? hh : h)) * W + (ok ? ww : wx0)) * C + c);      // unconditional load on a clamped address
            raw[r][j] = ok ? t : make_uint2(0u, 0u);
        }
    }
    float4 acc[8];
    const float4 b0 = (bias && !FLIP) ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = b0;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint2 t = raw[r][j];
            const float v0 = bf16_bits_to_f32(t.x & 0xffffu), v1 = bf16_bits_to_f32(t.x >> 16), v2 = bf16_bits_to_f32(t.y & 0xffffu), v3 = bf16_bits_to_f32(t.y >> 16);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int p = j - kw;
                if (p < 0 || p >= 8) continue;
                const int tap = FLIP ? (2 - r) * 3 + (2 - kw) : r * 3 + kw;
                acc[p].x += v0 * wf[tap]; acc[p].y += v1 * wf[9 + tap]; acc[p].z += v2 * wf[18 + tap]; acc[p].w += v3 * wf[27 + tap];
            }
        }
    }
    To* o = y + ((n * H + h) * W + wx0) * (int64_t)C + c;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        float4 a = acc[p];
        if (accumulate) {
            const float4 q = load4(o + (int64_t)p * C);
            a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
        }
        store4(o + (int64_t)p * C, a);
    }
}

// weight / bias gradient partials: part[block][c * 9 + tap] = sum over the block's pixels of dy[pix][c] x[pix + tap][c],
// part[block][9 C + c] = sum dy[pix][c].  Block = 256 threads = (64 channel quads) x (4 pixel lanes); reduced through LDS.
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_dw_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ part,
                                                          int H, int W, int C, int64_t pixels, int64_t pix_per_block) {
    __shared__ float red[4][64][40];
    const int cq = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = 4 * ((int)blockIdx.x * 64 + cq);
    const bool cok = c < C;
    const int cc = cok ? c : 0;
    float acc[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i] = 0.f;
    const int64_t p0 = (int64_t)blockIdx.y * pix_per_block;
    int64_t p1 = p0 + pix_per_block;
    p1 = p1 < pixels ? p1 : pixels;
    // (wx, h) of the lane's first pixel once, then stepped: two 64-bit divisions per pixel were most of the loop's instructions
    int wx = (int)((p0 + pl) % W), h = (int)(((p0 + pl) / W) % H);
    for (int64_t pix = p0 + pl; pix < p1; pix += 4, wx += 4) {
        while (wx >= W) {
            wx -= W;
            h = h + 1 == H ? 0 : h + 1;
        }
        const float4 g = load4(dy + pix * C + cc);
        acc[36] += g.x; acc[37] += g.y; acc[38] += g.z; acc[39] += g.w;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h + kh - 1;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ww = wx + kw - 1;
                const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
                const float4 v = load4(x + (pix + (ok ? (int64_t)(kh - 1) * W + (kw - 1) : 0)) * C + cc);
                const float m = ok ? 1.f : 0.f;
                const int tap = kh * 3 + kw;
                acc[tap] += m * g.x * v.x; acc[9 + tap] += m * g.y * v.y; acc[18 + tap] += m * g.z * v.z; acc[27 + tap] += m * g.w * v.w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 40; ++i) red[pl][cq][i] = acc[i];
    __syncthreads();
    if (pl == 0 && cok) {
        float* out = part + (int64_t)blockIdx.y * (10 * (int64_t)C);
#pragma unroll
        for (int i = 0; i < 40; ++i) {
            const float s = (red[0][cq][i] + red[1][cq][i]) + (red[2][cq][i] + red[3][cq][i]);
            if (i < 36) out[(c + i / 9) * 9 + i % 9] = s;
            else out[9 * (int64_t)C + c + (i - 36)] = s;
        }
    }
}

// the same partials with P consecutive pixels of a row per step (round 5; bf16, W % P == 0, pix_per_block % P == 0): P gradient loads + the 3 x (P + 2) window -- 22 loads for
// P = 4 pixels (the kernel above: 10 per pixel), every one in flight before the first multiply.  (P = 8 needs 256 registers: one wave per SIMD.)
template <int P>
__global__ __launch_bounds__(256) void dwconv3x3_dw_px_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ part,
                                                             int H, int W, int C, int64_t pixels, int64_t pix_per_block) {
    __shared__ float red[4][64][40];
    const int cq = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const int c = 4 * ((int)blockIdx.x * 64 + cq);
    const bool cok = c < C;
    const int cc = cok ? c : 0;
    float acc[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) acc[i] = 0.f;
    const int64_t p0 = (int64_t)blockIdx.y * pix_per_block;
    int64_t p1 = p0 + pix_per_block;
    p1 = p1 < pixels ? p1 : pixels;
    for (int64_t pix = p0 + P * pl; pix < p1; pix += 4 * P) {      // (pix % P == 0 and W % P == 0: the P pixels lie in one row)
        const int wx0 = (int)(pix % W), h = (int)((pix / W) % H);
        uint2 gr[P], raw[3][P + 2];
#pragma unroll
        for (int q = 0; q < P; ++q) gr[q] = *reinterpret_cast<const uint2*>(dy + (pix + q) * C + cc);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hh = h + r - 1;
            const bool rok = hh >= 0 && hh < H;
#pragma unroll
            for (int jj = 0; jj < P + 2; ++jj) {
                const int ww = wx0 + jj - 1;
                const bool ok = rok && ww >= 0 && ww < W;
                const uint2 t = *reinterpret_cast<const uint2*>(x + (pix + (ok ? (int64_t)(r - 1) * W + (jj - 1) : 0)) * C + cc);
                raw[r][jj] = ok ? t : make_uint2(0u, 0u);
            }
        }
        float g[P][4];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            g[q][0] = bf16_bits_to_f32(gr[q].x & 0xffffu); g[q][1] = bf16_bits_to_f32(gr[q].x >> 16);
            g[q][2] = bf16_bits_to_f32(gr[q].y & 0xffffu); g[q][3] = bf16_bits_to_f32(gr[q].y >> 16);
            acc[36] += g[q][0]; acc[37] += g[q][1]; acc[38] += g[q][2]; acc[39] += g[q][3];
        }
#pragma unroll
        for (int jj = 0; jj < P + 2; ++jj) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const uint2 t = raw[r][jj];
                const float v0 = bf16_bits_to_f32(t.x & 0xffffu), v1 = bf16_bits_to_f32(t.x >> 16), v2 = bf16_bits_to_f32(t.y & 0xffffu), v3 = bf16_bits_to_f32(t.y >> 16);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int q = jj - kw;
                    if (q < 0 || q >= P) continue;
                    const int tap = r * 3 + kw;
                    acc[tap] += g[q][0] * v0; acc[9 + tap] += g[q][1] * v1; acc[18 + tap] += g[q][2] * v2; acc[27 + tap] += g[q][3] * v3;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 40; ++i) red[pl][cq][i] = acc[i];
    __syncthreads();
    if (pl == 0 && cok) {
        float* out = part + (int64_t)blockIdx.y * (10 * (int64_t)C);
#pragma unroll
        for (int i = 0; i < 40; ++i) {
            const float sum = (red[0][cq][i] + red[1][cq][i]) + (red[2][cq][i] + red[3][cq][i]);
            if (i < 36) out[(c + i / 9) * 9 + i % 9] = sum;
            else out[9 * (int64_t)C + c + (i - 36)] = sum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// softmax over the P points of each (row, group): logits [rows][ld] -> probabilities [rows][G * P]; thread = (row, group)
// PC = the point count at compile time (9 / 8: InternImage's 3 x 3 with / without the centre; 0 = run-time P <= 32): with a run-time P the
// per-thread array is indexed dynamically and lives in scratch memory
template <typename T, int PC>
__global__ __launch_bounds__(256) void softmax_groups_fwd_kernel(const T* __restrict__ logits, int64_t ld, T* __restrict__ prob, int G, int P_, int64_t total) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;      // (total < 2^31, checked by the host)
    if (idx >= (uint32_t)total) return;
    const int P = PC ? PC : P_;
    const uint32_t row = idx / (uint32_t)G;
    const int g = (int)(idx - row * (uint32_t)G);
    const T* in = logits + (int64_t)row * ld + g * P;
    float v[PC ? PC : 32];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        v[i] = ld1(in + i);
        m = fmaxf(m, v[i]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        v[i] = __expf(v[i] - m);
        s += v[i];
    }
    const float inv = 1.f / s;
    T* out = prob + (int64_t)idx * P;
#pragma unroll
    for (int i = 0; i < P; ++i) Elem<T>::store(out + i, v[i] * inv);
}
// dlogits[rows][ld] = p (dprob - sum p dprob); columns G * P .. ld are zeroed (they feed a GEMM as padded contraction columns)
template <typename T, int PC>
__global__ __launch_bounds__(256) void softmax_groups_bwd_kernel(const T* __restrict__ prob, const float* __restrict__ dprob, T* __restrict__ dlogits, int64_t ld,
                                                                int G, int P_, int64_t total) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= (uint32_t)total) return;
    const int P = PC ? PC : P_;
    const uint32_t row32 = idx / (uint32_t)G;
    const int g = (int)(idx - row32 * (uint32_t)G);
    const int64_t row = row32;
    const T* p = prob + (int64_t)idx * P;
    const float* d = dprob + (int64_t)idx * P;
    float pv[PC ? PC : 32], dv[PC ? PC : 32], dot = 0.f;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        pv[i] = ld1(p + i);
        dv[i] = d[i];
        dot += pv[i] * dv[i];
    }
    T* out = dlogits + row * ld + g * P;
#pragma unroll
    for (int i = 0; i < P; ++i) Elem<T>::store(out + i, pv[i] * (dv[i] - dot));
    if (g == G - 1)
        for (int i = G * P; i < ld; ++i) Elem<T>::store(dlogits + row * ld + i, 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------
// out32 = x32 + s[sample] * gamma * z  (+ an ACT-dtype copy for the next GEMM / depth-wise conv);  thread = 4 channels of a row
template <typename T>
__global__ __launch_bounds__(256) void scale_residual_fwd_kernel(const float* __restrict__ x, const T* __restrict__ z, const float* __restrict__ gamma,
                                                                const float* __restrict__ sample_scale, int rows_per_sample, float* __restrict__ out,
                                                                T* __restrict__ out_act, int C, int64_t total) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;        // 32-bit index arithmetic (the host checks total < 2^31): 64-bit divisions cost ~150 instructions each
    if (idx >= (uint32_t)total) return;
    const uint32_t c4 = (uint32_t)C >> 2, row32 = idx / c4;
    const int c = 4 * (int)(idx - row32 * c4);
    const int64_t row = row32;
    const float s = sample_scale ? sample_scale[row32 / (uint32_t)rows_per_sample] : 1.f;
    const float4 xv = *reinterpret_cast<const float4*>(x + row * C + c), zv = load4(z + row * C + c), gv = *reinterpret_cast<const float4*>(gamma + c);
    const float4 o = make_float4(xv.x + s * gv.x * zv.x, xv.y + s * gv.y * zv.y, xv.z + s * gv.z * zv.z, xv.w + s * gv.w * zv.w);
    *reinterpret_cast<float4*>(out + row * C + c) = o;
    if (out_act) store4(out_act + row * C + c, o);
}
// dz = s * gamma * dout (ACT dtype, the LayerNorm backward's input); dgamma partials: part[block][c] = sum_rows s * dout * z
template <typename T>
__global__ __launch_bounds__(256) void scale_residual_bwd_kernel(const float* __restrict__ dout, const T* __restrict__ z, const float* __restrict__ gamma,
                                                                const float* __restrict__ sample_scale, int rows_per_sample, T* __restrict__ dz,
                                                                float* __restrict__ part, int C, int64_t rows, int64_t rows_per_block) {
    __shared__ float4 red[4][64];
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = 4 * ((int)blockIdx.x * 64 + cq);
    const bool cok = c < C;
    const int cc = cok ? c : 0;
    const float4 gv = *reinterpret_cast<const float4*>(gamma + cc);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    r1 = r1 < rows ? r1 : rows;
    for (int64_t row = r0 + rl; row < r1; row += 4) {
        const float s = sample_scale ? sample_scale[(uint32_t)row / (uint32_t)rows_per_sample] : 1.f;      // (rows < 2^31, checked by the host)
        const float4 d = *reinterpret_cast<const float4*>(dout + row * C + cc), zv = load4(z + row * C + cc);
        acc.x += s * d.x * zv.x; acc.y += s * d.y * zv.y; acc.z += s * d.z * zv.z; acc.w += s * d.w * zv.w;
        if (cok) store4(dz + row * C + c, make_float4(s * gv.x * d.x, s * gv.y * d.y, s * gv.z * d.z, s * gv.w * d.w));
    }
    red[rl][cq] = acc;
    __syncthreads();
    if (rl == 0 && cok) {
        const float4 a = red[0][cq], b = red[1][cq], e = red[2][cq], f = red[3][cq];
        *reinterpret_cast<float4*>(part + (int64_t)blockIdx.y * C + c) = make_float4((a.x + b.x) + (e.x + f.x), (a.y + b.y) + (e.y + f.y), (a.z + b.z) + (e.z + f.z), (a.w + b.w) + (e.w + f.w));
    }
}

// dst[r][c] = c < n ? src[r][c] : 0   (f32 rows -> ACT rows of pitch ld >= n: the padded contraction operand of a GEMM)
template <typename T>
__global__ __launch_bounds__(256) void cast_pad_rows_kernel(const float* __restrict__ src, int64_t n, T* __restrict__ dst, int64_t ld, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int64_t r = idx / ld, c = idx - r * ld;
    Elem<T>::store(dst + idx, c < n ? src[r * n + c] : 0.f);
}

// dst[r][0..n) = src[r][0..n), both with their own row pitch (compacting a padded GEMM output)
template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ src, int64_t src_ld, T* __restrict__ dst, int64_t dst_ld, int64_t n, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int64_t r = idx / n, c = idx - r * n;
    dst[r * dst_ld + c] = src[r * src_ld + c];
}

inline unsigned blocks_for(int64_t total) { return (unsigned)((total + 255) / 256); }


// ---------------------------------------------------------------------------------------------------------------------
// Depth-wise k x k convolution for any odd k (InternImage-H/G's dw_kernel_size, DCNM:124, 146-151), channels-last, stride 1, "same" padding: the plain forms --
// a lane = (pixel, 4 channels), a loop over the taps.  (The 3 x 3 kernels above are the fast path every MTP configuration runs.)
template <typename T, bool DX>
__global__ __launch_bounds__(256) void dwconvk_kernel(const T* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y, float* __restrict__ dx,
                                                     int accumulate, int H, int W, int C, int k, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c4n = C >> 2, c = (int)(idx % c4n) * 4;
    const int64_t pix = idx / c4n;
    const int wo = (int)(pix % W);
    const int64_t t = pix / W;
    const int ho = (int)(t % H);
    const int64_t n = t / H;
    const int p = (k - 1) >> 1, kk = k * k;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!DX && bias) acc = *reinterpret_cast<const float4*>(bias + c);
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) {
            // forward: y[h][w] += w[i][j] x[h + i - p][w + j - p];   data gradient: dx[h][w] += w[i][j] dy[h - i + p][w - j + p]
            const int hh = DX ? ho - i + p : ho + i - p, ww = DX ? wo - j + p : wo + j - p;
            if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
            const float4 v = load4(src + ((n * H + hh) * W + ww) * C + c);
            const int tp = i * k + j;
            acc.x += w[(c + 0) * kk + tp] * v.x; acc.y += w[(c + 1) * kk + tp] * v.y; acc.z += w[(c + 2) * kk + tp] * v.z; acc.w += w[(c + 3) * kk + tp] * v.w;
        }
    if (DX) {
        float* d = dx + pix * C + c;
        if (accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(d);
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        *reinterpret_cast<float4*>(d) = acc;
    } else {
        store4(y + pix * C + c, acc);
    }
}
// weight / bias gradient: grid (column blocks, row blocks, taps); block = 64 four-channel columns x 4 row lanes; f32 atomics into dw (C, 1, k, k) and db (C)
template <typename T>
__global__ __launch_bounds__(256) void dwconvk_dw_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dw, float* __restrict__ db, int H, int W, int C, int k,
                                                        int64_t pixels, int64_t rows_per_block) {
    __shared__ float4 red[4][64];
    __shared__ float4 redb[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + cx) * 4;
    const int tp = blockIdx.z, i = tp / k, j = tp - i * k, p = (k - 1) >> 1;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    r1 = r1 < pixels ? r1 : pixels;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), sb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) {
        for (int64_t r = r0 + ry; r < r1; r += 4) {
            const int wo = (int)(r % W);
            const int64_t t = r / W;
            const int ho = (int)(t % H);
            const int64_t n = t / H;
            const float4 g = load4(dy + r * C + c);
            if (tp == 0) { sb.x += g.x; sb.y += g.y; sb.z += g.z; sb.w += g.w; }
            const int hh = ho + i - p, ww = wo + j - p;
            if (hh < 0 || hh >= H || ww < 0 || ww >= W) continue;
            const float4 v = load4(x + ((n * H + hh) * W + ww) * C + c);
            s.x += g.x * v.x; s.y += g.y * v.y; s.z += g.z * v.z; s.w += g.w * v.w;
        }
    }
    red[ry][cx] = s;
    redb[ry][cx] = sb;
    __syncthreads();
    if (ry == 0 && c < C) {
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float4 a = red[q][cx], b = redb[q][cx];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            sb.x += b.x; sb.y += b.y; sb.z += b.z; sb.w += b.w;
        }
        const int kk = k * k;
        atomicAdd(dw + (c + 0) * kk + tp, s.x); atomicAdd(dw + (c + 1) * kk + tp, s.y); atomicAdd(dw + (c + 2) * kk + tp, s.z); atomicAdd(dw + (c + 3) * kk + tp, s.w);
        if (tp == 0 && db) { atomicAdd(db + c, sb.x); atomicAdd(db + c + 1, sb.y); atomicAdd(db + c + 2, sb.z); atomicAdd(db + c + 3, sb.w); }
    }
}

// center_feature_scale (InternImage-H/G, DCNM:209-215):  out = y (1 - s) + xp s,  s[row][group] = sigmoid(logits[row][group]) shared by the group's channels.
// A lane = (row, group).  Backward: dy = dout (1 - s);  dxp (f32) = dout s;  dlogits = s (1 - s) sum_c dout_c (xp_c - y_c), pad columns zeroed.
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void cfs_kernel(const T* __restrict__ a0, const T* __restrict__ y, const T* __restrict__ xp, const T* __restrict__ logits, int64_t ld,
                                                 T* __restrict__ out, float* __restrict__ dxp, T* __restrict__ dlogits, int G, int GC, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int g = (int)(idx % G);
    const int64_t row = idx / G;
    const int64_t e0 = (row * G + g) * GC;
    const float s = 1.0f / (1.0f + __expf(-Elem<T>::load(logits + row * ld + g)));
    if (!BWD) {
        for (int c = 0; c < GC; c += 4) {
            const float4 u = load4(y + e0 + c), v = load4(xp + e0 + c);
            store4(out + e0 + c, make_float4(u.x + s * (v.x - u.x), u.y + s * (v.y - u.y), u.z + s * (v.z - u.z), u.w + s * (v.w - u.w)));
        }
    } else {
        float acc = 0.f;
        for (int c = 0; c < GC; c += 4) {
            const float4 d = load4(a0 + e0 + c), u = load4(y + e0 + c), v = load4(xp + e0 + c);
            acc += d.x * (v.x - u.x) + d.y * (v.y - u.y) + d.z * (v.z - u.z) + d.w * (v.w - u.w);
            store4(out + e0 + c, make_float4(d.x * (1.f - s), d.y * (1.f - s), d.z * (1.f - s), d.w * (1.f - s)));
            *reinterpret_cast<float4*>(dxp + e0 + c) = make_float4(d.x * s, d.y * s, d.z * s, d.w * s);
        }
        Elem<T>::store(dlogits + row * ld + g, acc * s * (1.f - s));
        if (g == 0)
            for (int64_t q = G; q < ld; ++q) Elem<T>::store(dlogits + row * ld + q, 0.f);
    }
}

}  // namespace

extern "C" int mtp_im2col3x3(const void* x, int x_dtype, int64_t sN, int64_t sH, int64_t sW, int64_t sC, void* cols, int cols_dtype,
                             int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t stride, int64_t Kp, mtp_stream_t stream) {
    if (!x || !cols || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || stride < 1 || Kp < 9 * Cin) return MTP_ERR_ARG;
    const int Ho = (int)((H - 1) / stride + 1), Wo = (int)((W - 1) / stride + 1);      // (H + 2 - 3) / s + 1
    const int64_t total = N * Ho * Wo * Kp;
    if (total / 256 > 0x7fffffff) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MTP_BF16 && cols_dtype == MTP_BF16 && sC == 1 && !(Cin & 7) && !(Kp & 7) && !((sN | sH | sW) & 7) && !(((uintptr_t)x | (uintptr_t)cols) & 15)) {
        const int64_t total8 = total / 8;
        hipLaunchKernelGGL(im2col3x3_v8_kernel, dim3(blocks_for(total8)), dim3(256), 0, s, (const bf16_t*)x, sN, sH, sW, (bf16_t*)cols, (int)H, (int)W, (int)Cin, Ho, Wo, (int)stride, (int)Kp, total8);
        return mtp_launch_status();
    }
    const dim3 grid(blocks_for(total)), block(256);
#define MTP_LAUNCH_I2C(TX, TC) hipLaunchKernelGGL((im2col3x3_kernel<TX, TC>), grid, block, 0, s, (const TX*)x, sN, sH, sW, sC, (TC*)cols, (int)H, (int)W, (int)Cin, Ho, Wo, (int)stride, (int)Kp, total)
    if (x_dtype == MTP_F32 && cols_dtype == MTP_F32) MTP_LAUNCH_I2C(float, float);
    else if (x_dtype == MTP_F32 && cols_dtype == MTP_BF16) MTP_LAUNCH_I2C(float, bf16_t);
    else if (x_dtype == MTP_BF16 && cols_dtype == MTP_BF16) MTP_LAUNCH_I2C(bf16_t, bf16_t);
    else return MTP_ERR_UNSUPPORTED;
#undef MTP_LAUNCH_I2C
    return mtp_launch_status();
}

extern "C" int mtp_col2im3x3(const void* dcols, int cols_dtype, float* dx, int64_t sN, int64_t sH, int64_t sW, int64_t sC,
                             int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t stride, int64_t Kp, int accumulate, mtp_stream_t stream) {
    if (!dcols || !dx || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || stride < 1 || Kp < 9 * Cin) return MTP_ERR_ARG;
    const int Ho = (int)((H - 1) / stride + 1), Wo = (int)((W - 1) / stride + 1);
    const int64_t total = N * H * W * Cin;
    hipStream_t s = (hipStream_t)stream;
    if (cols_dtype == MTP_BF16 && sC == 1 && !(Cin & 7) && !(Kp & 7) && !((sN | sH | sW) & 7) && !(((uintptr_t)dcols | (uintptr_t)dx) & 15)) {
        const int64_t total8 = total / 8;
        hipLaunchKernelGGL(col2im3x3_v8_kernel, dim3(blocks_for(total8)), dim3(256), 0, s, (const bf16_t*)dcols, dx, sN, sH, sW, (int)H, (int)W, (int)Cin, Ho, Wo, (int)stride, (int)Kp, accumulate, total8);
        return mtp_launch_status();
    }
    const dim3 grid(blocks_for(total)), block(256);
    if (cols_dtype == MTP_BF16)
        hipLaunchKernelGGL((col2im3x3_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)dcols, dx, sN, sH, sW, sC, (int)H, (int)W, (int)Cin, Ho, Wo, (int)stride, (int)Kp, accumulate, total);
    else if (cols_dtype == MTP_F32)
        hipLaunchKernelGGL((col2im3x3_kernel<float>), grid, block, 0, s, (const float*)dcols, dx, sN, sH, sW, sC, (int)H, (int)W, (int)Cin, Ho, Wo, (int)stride, (int)Kp, accumulate, total);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_conv3x3_pack(const float* w, void* w2, void* w2t, int dtype, int64_t Cout, int64_t Cin, int64_t Kp, mtp_stream_t stream) {
    if (!w || (!w2 && !w2t) || Cout <= 0 || Cin <= 0 || Kp < 9 * Cin) return MTP_ERR_ARG;
    const dim3 grid(blocks_for(Cout * Kp)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16) hipLaunchKernelGGL((conv3x3_pack_kernel<bf16_t>), grid, block, 0, s, w, (bf16_t*)w2, (bf16_t*)w2t, (int)Cout, (int)Cin, (int)Kp);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((conv3x3_pack_kernel<float>), grid, block, 0, s, w, (float*)w2, (float*)w2t, (int)Cout, (int)Cin, (int)Kp);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_conv3x3_unpack_grad(const float* dw2, float* dw, int64_t Cout, int64_t Cin, int64_t Kp, mtp_stream_t stream) {
    if (!dw2 || !dw || Cout <= 0 || Cin <= 0 || Kp < 9 * Cin) return MTP_ERR_ARG;
    hipLaunchKernelGGL(conv3x3_unpack_kernel, dim3(blocks_for(Cout * Cin * 9)), dim3(256), 0, (hipStream_t)stream, dw2, dw, (int)Cout, (int)Cin, (int)Kp);
    return mtp_launch_status();
}

extern "C" int mtp_pack_rows_padded(const float* w, void* wp, void* wpt, int dtype, int64_t R, int64_t C, int64_t Rp, mtp_stream_t stream) {
    if (!w || (!wp && !wpt) || R <= 0 || C <= 0 || Rp < R) return MTP_ERR_ARG;
    const dim3 grid(blocks_for(Rp * C)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16) hipLaunchKernelGGL((pack_rows_padded_kernel<bf16_t>), grid, block, 0, s, w, (bf16_t*)wp, (bf16_t*)wpt, (int)R, (int)C, (int)Rp);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((pack_rows_padded_kernel<float>), grid, block, 0, s, w, (float*)wp, (float*)wpt, (int)R, (int)C, (int)Rp);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_dwconv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int dtype, int64_t N, int64_t H, int64_t W, int64_t C, mtp_stream_t stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4)) return MTP_ERR_ARG;
    const int64_t total = N * H * W * (C / 4);
    if (total >= ((int64_t)1 << 31)) return MTP_ERR_UNSUPPORTED;      // 32-bit index arithmetic in the kernel
    const dim3 grid(blocks_for(total)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16 && !(W & 7) && !(((uintptr_t)w | (uintptr_t)(bias ? bias : w)) & 15)) {
        hipLaunchKernelGGL((dwconv3x3_p8_kernel<bf16_t, false>), dim3(blocks_for(total / 8)), block, 0, s, (const bf16_t*)x, w, bias, (bf16_t*)y, (int)H, (int)W, (int)C, 0, total / 8);
        return mtp_launch_status();
    }
    if (dtype == MTP_BF16) hipLaunchKernelGGL((dwconv3x3_kernel<bf16_t, bf16_t, false>), grid, block, 0, s, (const bf16_t*)x, w, bias, (bf16_t*)y, (int)H, (int)W, (int)C, 0, total);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((dwconv3x3_kernel<float, float, false>), grid, block, 0, s, (const float*)x, w, bias, (float*)y, (int)H, (int)W, (int)C, 0, total);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_dwconv3x3_bwd_dx(const void* dy, int dtype, const float* w, float* dx, int accumulate, int64_t N, int64_t H, int64_t W, int64_t C, mtp_stream_t stream) {
    if (!dy || !w || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4)) return MTP_ERR_ARG;
    const int64_t total = N * H * W * (C / 4);
    if (total >= ((int64_t)1 << 31)) return MTP_ERR_UNSUPPORTED;      // 32-bit index arithmetic in the kernel
    const dim3 grid(blocks_for(total)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16 && !(W & 7) && !((uintptr_t)w & 15)) {
        hipLaunchKernelGGL((dwconv3x3_p8_kernel<float, true>), dim3(blocks_for(total / 8)), block, 0, s, (const bf16_t*)dy, w, (const float*)nullptr, dx, (int)H, (int)W, (int)C, accumulate, total / 8);
        return mtp_launch_status();
    }
    if (dtype == MTP_BF16) hipLaunchKernelGGL((dwconv3x3_kernel<bf16_t, float, true>), grid, block, 0, s, (const bf16_t*)dy, w, (const float*)nullptr, dx, (int)H, (int)W, (int)C, accumulate, total);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((dwconv3x3_kernel<float, float, true>), grid, block, 0, s, (const float*)dy, w, (const float*)nullptr, dx, (int)H, (int)W, (int)C, accumulate, total);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int64_t mtp_dwconv3x3_bwd_dw_partial_rows(int64_t N, int64_t H, int64_t W) {
    const int64_t pixels = N * H * W;
    const int64_t nb = (pixels + 127) / 128;      // (round 4: up to 1024 workgroups; with 256 a level-0 launch had one 4-wave workgroup per CU and ran
    return nb < 1024 ? nb : 1024;                 //  on the latency of its loads.  Round 5: 32 pixels per block measured -- no gain, four times the partial rows)
}

/* part: (mtp_dwconv3x3_bwd_dw_partial_rows, 10 C) f32 = per-block partials of [dweight (C, 9) | dbias (C)] */
extern "C" int mtp_dwconv3x3_bwd_dw(const void* dy, const void* x, int dtype, float* part, int64_t N, int64_t H, int64_t W, int64_t C, mtp_stream_t stream) {
    if (!dy || !x || !part || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4)) return MTP_ERR_ARG;
    const int64_t pixels = N * H * W, nb = mtp_dwconv3x3_bwd_dw_partial_rows(N, H, W);
    const int64_t ppb = (pixels + nb - 1) / nb;
    const dim3 grid((unsigned)((C / 4 + 63) / 64), (unsigned)nb), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16 && !(W & 3)) {      // blocks of whole 4-pixel groups (the last blocks may be empty: they write zero partials)
        const int64_t ppb4 = (ppb + 3) / 4 * 4;
        hipLaunchKernelGGL(dwconv3x3_dw_px_kernel<4>, grid, block, 0, s, (const bf16_t*)dy, (const bf16_t*)x, part, (int)H, (int)W, (int)C, pixels, ppb4);
        return mtp_launch_status();
    }
    if (dtype == MTP_BF16) hipLaunchKernelGGL((dwconv3x3_dw_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)dy, (const bf16_t*)x, part, (int)H, (int)W, (int)C, pixels, ppb);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((dwconv3x3_dw_kernel<float>), grid, block, 0, s, (const float*)dy, (const float*)x, part, (int)H, (int)W, (int)C, pixels, ppb);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_softmax_groups_fwd(const void* logits, int64_t ld, void* prob, int dtype, int64_t rows, int64_t G, int64_t P, mtp_stream_t stream) {
    if (!logits || !prob || rows <= 0 || G <= 0 || P <= 0 || P > 32 || ld < G * P) return MTP_ERR_ARG;
    const int64_t total = rows * G;
    const dim3 grid(blocks_for(total)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (total >= ((int64_t)1 << 31)) return MTP_ERR_UNSUPPORTED;
#define MTP_SMX_FWD(T_, PC_) hipLaunchKernelGGL((softmax_groups_fwd_kernel<T_, PC_>), grid, block, 0, s, (const T_*)logits, ld, (T_*)prob, (int)G, (int)P, total)
    if (dtype == MTP_BF16) { if (P == 9) MTP_SMX_FWD(bf16_t, 9); else if (P == 8) MTP_SMX_FWD(bf16_t, 8); else MTP_SMX_FWD(bf16_t, 0); }
    else if (dtype == MTP_F32) { if (P == 9) MTP_SMX_FWD(float, 9); else if (P == 8) MTP_SMX_FWD(float, 8); else MTP_SMX_FWD(float, 0); }
    else return MTP_ERR_UNSUPPORTED;
#undef MTP_SMX_FWD
    return mtp_launch_status();
}

extern "C" int mtp_softmax_groups_bwd(const void* prob, const float* dprob, void* dlogits, int64_t ld, int dtype, int64_t rows, int64_t G, int64_t P, mtp_stream_t stream) {
    if (!prob || !dprob || !dlogits || rows <= 0 || G <= 0 || P <= 0 || P > 32 || ld < G * P) return MTP_ERR_ARG;
    const int64_t total = rows * G;
    const dim3 grid(blocks_for(total)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (total >= ((int64_t)1 << 31)) return MTP_ERR_UNSUPPORTED;
#define MTP_SMX_BWD(T_, PC_) hipLaunchKernelGGL((softmax_groups_bwd_kernel<T_, PC_>), grid, block, 0, s, (const T_*)prob, dprob, (T_*)dlogits, ld, (int)G, (int)P, total)
    if (dtype == MTP_BF16) { if (P == 9) MTP_SMX_BWD(bf16_t, 9); else if (P == 8) MTP_SMX_BWD(bf16_t, 8); else MTP_SMX_BWD(bf16_t, 0); }
    else if (dtype == MTP_F32) { if (P == 9) MTP_SMX_BWD(float, 9); else if (P == 8) MTP_SMX_BWD(float, 8); else MTP_SMX_BWD(float, 0); }
    else return MTP_ERR_UNSUPPORTED;
#undef MTP_SMX_BWD
    return mtp_launch_status();
}

extern "C" int mtp_scale_residual_fwd(const float* x, const void* z, int dtype, const float* gamma, const float* sample_scale, int64_t rows_per_sample,
                                      float* out, void* out_act, int64_t rows, int64_t C, mtp_stream_t stream) {
    if (!x || !z || !gamma || !out || rows <= 0 || C <= 0 || (C % 4) || (sample_scale && rows_per_sample <= 0)) return MTP_ERR_ARG;
    const int64_t total = rows * (C / 4);
    if (total >= ((int64_t)1 << 31)) return MTP_ERR_UNSUPPORTED;      // 32-bit index arithmetic in the kernel
    const dim3 grid(blocks_for(total)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int rps = (int)(rows_per_sample > 0 ? rows_per_sample : 1);
    if (dtype == MTP_BF16) hipLaunchKernelGGL((scale_residual_fwd_kernel<bf16_t>), grid, block, 0, s, x, (const bf16_t*)z, gamma, sample_scale, rps, out, (bf16_t*)out_act, (int)C, total);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((scale_residual_fwd_kernel<float>), grid, block, 0, s, x, (const float*)z, gamma, sample_scale, rps, out, (float*)out_act, (int)C, total);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int64_t mtp_scale_residual_bwd_partial_rows(int64_t rows) {
    const int64_t nb = (rows + 63) / 64;
    return nb < 256 ? nb : 256;
}

/* part: (mtp_scale_residual_bwd_partial_rows(rows), C) f32 = per-block partials of dgamma */
extern "C" int mtp_scale_residual_bwd(const float* dout, const void* z, int dtype, const float* gamma, const float* sample_scale, int64_t rows_per_sample,
                                      void* dz, float* part, int64_t rows, int64_t C, mtp_stream_t stream) {
    if (!dout || !z || !gamma || !dz || !part || rows <= 0 || C <= 0 || (C % 4) || (sample_scale && rows_per_sample <= 0)) return MTP_ERR_ARG;
    if (rows >= ((int64_t)1 << 31)) return MTP_ERR_UNSUPPORTED;
    const int64_t nb = mtp_scale_residual_bwd_partial_rows(rows);
    const int64_t rpb = (rows + nb - 1) / nb;
    const dim3 grid((unsigned)((C / 4 + 63) / 64), (unsigned)nb), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int rps = (int)(rows_per_sample > 0 ? rows_per_sample : 1);
    if (dtype == MTP_BF16) hipLaunchKernelGGL((scale_residual_bwd_kernel<bf16_t>), grid, block, 0, s, dout, (const bf16_t*)z, gamma, sample_scale, rps, (bf16_t*)dz, part, (int)C, rows, rpb);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((scale_residual_bwd_kernel<float>), grid, block, 0, s, dout, (const float*)z, gamma, sample_scale, rps, (float*)dz, part, (int)C, rows, rpb);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_cast_pad_rows(const float* src, int64_t n, void* dst, int dst_dtype, int64_t ld, int64_t rows, mtp_stream_t stream) {
    if (!src || !dst || n <= 0 || ld < n || rows <= 0) return MTP_ERR_ARG;
    const int64_t total = rows * ld;
    const dim3 grid(blocks_for(total)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dst_dtype == MTP_BF16) hipLaunchKernelGGL((cast_pad_rows_kernel<bf16_t>), grid, block, 0, s, src, n, (bf16_t*)dst, ld, total);
    else if (dst_dtype == MTP_F32) hipLaunchKernelGGL((cast_pad_rows_kernel<float>), grid, block, 0, s, src, n, (float*)dst, ld, total);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_copy_rows(const void* src, int64_t src_ld, void* dst, int64_t dst_ld, int dtype, int64_t n, int64_t rows, mtp_stream_t stream) {
    if (!src || !dst || n <= 0 || src_ld < n || dst_ld < n || rows <= 0) return MTP_ERR_ARG;
    const int64_t total = rows * n;
    const dim3 grid(blocks_for(total)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16) hipLaunchKernelGGL((copy_rows_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)src, src_ld, (bf16_t*)dst, dst_ld, n, total);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((copy_rows_kernel<float>), grid, block, 0, s, (const float*)src, src_ld, (float*)dst, dst_ld, n, total);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

/* ---- depth-wise k x k convolution, any odd k (InternImage-H/G: dw_kernel_size, ops_dcnv3/modules/dcnv3.py:124,146-151) -- the plain forms; k = 3 callers use the
 * mtp_dwconv3x3_* entries.  w (C, 1, k, k) f32.  bwd_dw ACCUMULATES into dw / db (f32 atomics): clear them first. */
extern "C" int mtp_dwconv_fwd(const void* x, const float* w, const float* bias, void* y, int dtype, int64_t N, int64_t H, int64_t W, int64_t C, int k, mtp_stream_t stream) {
    if (!x || !w || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) || k < 1 || !(k & 1) || k > 15) return MTP_ERR_ARG;
    const int64_t total = N * H * W * (C / 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16) hipLaunchKernelGGL((dwconvk_kernel<bf16_t, false>), dim3(blocks_for(total)), dim3(256), 0, s, (const bf16_t*)x, w, bias, (bf16_t*)y, (float*)nullptr, 0, (int)H, (int)W, (int)C, k, total);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((dwconvk_kernel<float, false>), dim3(blocks_for(total)), dim3(256), 0, s, (const float*)x, w, bias, (float*)y, (float*)nullptr, 0, (int)H, (int)W, (int)C, k, total);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}
extern "C" int mtp_dwconv_bwd_dx(const void* dy, int dtype, const float* w, float* dx, int accumulate, int64_t N, int64_t H, int64_t W, int64_t C, int k, mtp_stream_t stream) {
    if (!dy || !w || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) || k < 1 || !(k & 1) || k > 15) return MTP_ERR_ARG;
    const int64_t total = N * H * W * (C / 4);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16) hipLaunchKernelGGL((dwconvk_kernel<bf16_t, true>), dim3(blocks_for(total)), dim3(256), 0, s, (const bf16_t*)dy, w, (const float*)nullptr, (bf16_t*)nullptr, dx, accumulate, (int)H, (int)W, (int)C, k, total);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((dwconvk_kernel<float, true>), dim3(blocks_for(total)), dim3(256), 0, s, (const float*)dy, w, (const float*)nullptr, (float*)nullptr, dx, accumulate, (int)H, (int)W, (int)C, k, total);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}
extern "C" int mtp_dwconv_bwd_dw(const void* dy, const void* x, int dtype, float* dw, float* db, int64_t N, int64_t H, int64_t W, int64_t C, int k, mtp_stream_t stream) {
    if (!dy || !x || !dw || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 4) || k < 1 || !(k & 1) || k > 15) return MTP_ERR_ARG;
    const int64_t pixels = N * H * W;
    int64_t rb = (pixels + 511) / 512;
    rb = rb < 256 ? rb : 256;
    const int64_t rpb = ((pixels + rb - 1) / rb + 3) / 4 * 4;
    const dim3 grid((unsigned)((C / 4 + 63) / 64), (unsigned)((pixels + rpb - 1) / rpb), (unsigned)(k * k)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16) hipLaunchKernelGGL((dwconvk_dw_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)dy, (const bf16_t*)x, dw, db, (int)H, (int)W, (int)C, k, pixels, rpb);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((dwconvk_dw_kernel<float>), grid, block, 0, s, (const float*)dy, (const float*)x, dw, db, (int)H, (int)W, (int)C, k, pixels, rpb);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

/* ---- center_feature_scale (InternImage-H/G; ops_dcnv3/modules/dcnv3.py:209-215): out (rows, G * GC) = y (1 - s) + xp s with s = sigmoid(logits[row][group]) (logits:
 * rows of ld >= G elements, the G-output Linear on the depth-wise branch).  bwd: dy (`dtype`) = dout (1 - s); dxp (f32) = dout s; dlogits (rows of ld, pad columns
 * zeroed) = s (1 - s) sum over the group's channels of dout (xp - y). */
extern "C" int mtp_center_feature_scale_fwd(const void* y, const void* xp, const void* logits, int64_t ld, void* out, int dtype, int64_t rows, int64_t G, int64_t GC, mtp_stream_t stream) {
    if (!y || !xp || !logits || !out || rows <= 0 || G <= 0 || GC <= 0 || (GC % 4) || ld < G) return MTP_ERR_ARG;
    const int64_t total = rows * G;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16) hipLaunchKernelGGL((cfs_kernel<bf16_t, false>), dim3(blocks_for(total)), dim3(256), 0, s, (const bf16_t*)nullptr, (const bf16_t*)y, (const bf16_t*)xp, (const bf16_t*)logits, ld, (bf16_t*)out, (float*)nullptr, (bf16_t*)nullptr, (int)G, (int)GC, total);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((cfs_kernel<float, false>), dim3(blocks_for(total)), dim3(256), 0, s, (const float*)nullptr, (const float*)y, (const float*)xp, (const float*)logits, ld, (float*)out, (float*)nullptr, (float*)nullptr, (int)G, (int)GC, total);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}
extern "C" int mtp_center_feature_scale_bwd(const void* dout, const void* y, const void* xp, const void* logits, int64_t ld, void* dy, float* dxp, void* dlogits, int dtype, int64_t rows,
                                            int64_t G, int64_t GC, mtp_stream_t stream) {
    if (!dout || !y || !xp || !logits || !dy || !dxp || !dlogits || rows <= 0 || G <= 0 || GC <= 0 || (GC % 4) || ld < G) return MTP_ERR_ARG;
    const int64_t total = rows * G;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16) hipLaunchKernelGGL((cfs_kernel<bf16_t, true>), dim3(blocks_for(total)), dim3(256), 0, s, (const bf16_t*)dout, (const bf16_t*)y, (const bf16_t*)xp, (const bf16_t*)logits, ld, (bf16_t*)dy, dxp, (bf16_t*)dlogits, (int)G, (int)GC, total);
    else if (dtype == MTP_F32) hipLaunchKernelGGL((cfs_kernel<float, true>), dim3(blocks_for(total)), dim3(256), 0, s, (const float*)dout, (const float*)y, (const float*)xp, (const float*)logits, ld, (float*)dy, dxp, (float*)dlogits, (int)G, (int)GC, total);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}
