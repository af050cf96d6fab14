// HBM-bound layout / elementwise kernels of the backbone path (gfx950): im2col for the patch embedding, dtype
// casts and weight (re)packing, token-major <-> NCHW feature-map transposes (FPN tail), MaxPool2d(2,2), small f32
// linear layers of the RVSA sampling heads, and the flat-buffer optimizer step.  All accesses are 8/16-byte vectors
// on the contiguous dimension; transposes go through a padded LDS tile so both sides stay coalesced.
#include "common.h"
#include <atomic>
#include <mutex>

namespace {

inline unsigned blocks_for(int64_t n, int per_block, int64_t cap = 1 << 20) {
    int64_t b = (n + per_block - 1) / per_block;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ------------------------------------------------------------------------------------------------ patchify
// cols[t][c*P*P + ky*P + kx] = img[b][c][py*P+ky][px*P+kx], t = (b*Hp + py)*Wp + px   (VIT:529,536-539)
template <typename T, bool INVERSE>
__global__ __launch_bounds__(256) void patchify_kernel(float* __restrict__ img, T* __restrict__ cols, int B, int Cin, int H, int W, int P, int Hp, int Wp) {
    const int K4 = Cin * P * P / 4, P4 = P / 4;
    const int64_t total = (int64_t)B * Hp * Wp * K4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int k4 = (int)(i % K4);
        const int64_t t = i / K4;
        const int kx4 = k4 % P4, ky = (k4 / P4) % P, c = k4 / (P4 * P);
        const int px = (int)(t % Wp), py = (int)((t / Wp) % Hp), b = (int)(t / ((int64_t)Wp * Hp));
        float* ip = img + (((int64_t)b * Cin + c) * H + py * P + ky) * W + px * P + kx4 * 4;
        T* cp = cols + t * (K4 * 4) + k4 * 4;
        if (INVERSE)
            *reinterpret_cast<float4*>(ip) = load4(cp);
        else
            store4(cp, *reinterpret_cast<const float4*>(ip));
    }
}

// ------------------------------------------------------------------------------------------------ cast
template <typename Ts, typename Td>
__global__ __launch_bounds__(256) void cast_kernel(const Ts* __restrict__ s, Td* __restrict__ d, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) store4(d + 4 * i, load4(s + 4 * i));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) Elem<Td>::store(d + (n4 << 2) + threadIdx.x, Elem<Ts>::load(s + (n4 << 2) + threadIdx.x));
}

// ------------------------------------------------------------------------------------------------ tiled transposes
// Generic 64x64 tile through LDS.  "row side": rows (length-C vectors, C contiguous);  "col side": (C, S) with S contiguous.
//   ROWS2COLS: out[bt][c][s] = in[bt][rowmap(s)][c]      (tokens -> NCHW; weight transpose with rowmap = identity)
//   else     : out[bt][rowmap(s)][c] = in[bt][c][s]      (NCHW -> tokens)
// rowmap(s): pixel s = (y, x) of the (Hp<<L, Wp<<L) map -> row ((py*Wp+px)*4 + q1)*4 + q2 ..., q_l = ky_l*2 + kx_l.
__device__ __forceinline__ int64_t pixel_to_row(int64_t s, int Wp, int L) {
    if (L == 0) return s;
    const int Wo = Wp << L;
    const int y = (int)(s / Wo), x = (int)(s % Wo);
    int64_t r = (int64_t)(y >> L) * Wp + (x >> L);
    for (int l = 1; l <= L; ++l) r = r * 4 + (((y >> (L - l)) & 1) << 1) + ((x >> (L - l)) & 1);
    return r;
}

template <typename Tin, typename Tout, bool ROWS2COLS>
__global__ __launch_bounds__(256) void transpose_kernel(const Tin* __restrict__ in, Tout* __restrict__ out, int64_t S, int64_t C, int Wp, int L) {
    __shared__ float tile[64][65];
    const int64_t bt = blockIdx.z;
    const int64_t s0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
    const Tin* ib = in + bt * S * C;
    Tout* ob = out + bt * S * C;
    const int t = threadIdx.x;
    const int a = t >> 2, g = (t & 3) * 16;   // a: index on the "slow" side of this phase, g: 16 contiguous elements
    if (ROWS2COLS) {
        // read rows: row s0+a, channels c0+g..g+15
        const int64_t s = s0 + a;
        if (s < S) {
            const Tin* rp = ib + pixel_to_row(s, Wp, L) * C + c0 + g;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (c0 + g + 4 * v < C) {
                    const float4 x = load4(rp + 4 * v);
                    tile[g + 4 * v + 0][a] = x.x; tile[g + 4 * v + 1][a] = x.y; tile[g + 4 * v + 2][a] = x.z; tile[g + 4 * v + 3][a] = x.w;
                }
            }
        }
        __syncthreads();
        // write cols: channel c0+a, positions s0+g..g+15
        const int64_t c = c0 + a;
        if (c < C) {
            Tout* wp = ob + c * S + s0 + g;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t s2 = s0 + g + 4 * v;
                if (s2 + 3 < S && (S & 3) == 0) {
                    store4(wp + 4 * v, make_float4(tile[a][g + 4 * v], tile[a][g + 4 * v + 1], tile[a][g + 4 * v + 2], tile[a][g + 4 * v + 3]));
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (s2 + e < S) Elem<Tout>::store(wp + 4 * v + e, tile[a][g + 4 * v + e]);
                }
            }
        }
    } else {
        const int64_t c = c0 + a;
        if (c < C) {
            const Tin* rp = ib + c * S + s0 + g;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int64_t s2 = s0 + g + 4 * v;
                if (s2 + 3 < S && (S & 3) == 0) {
                    const float4 x = load4(rp + 4 * v);
                    tile[a][g + 4 * v] = x.x; tile[a][g + 4 * v + 1] = x.y; tile[a][g + 4 * v + 2] = x.z; tile[a][g + 4 * v + 3] = x.w;
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (s2 + e < S) tile[a][g + 4 * v + e] = Elem<Tin>::load(rp + 4 * v + e);
                }
            }
        }
        __syncthreads();
        const int64_t s = s0 + a;
        if (s < S) {
            Tout* wp = ob + pixel_to_row(s, Wp, L) * C + c0 + g;
#pragma unroll
            for (int v = 0; v < 4; ++v)
                if (c0 + g + 4 * v < C)
                    store4(wp + 4 * v, make_float4(tile[g + 4 * v][a], tile[g + 4 * v + 1][a], tile[g + 4 * v + 2][a], tile[g + 4 * v + 3][a]));
        }
    }
}

// bf16 -> bf16 with S % 8 == 0 and C % 8 == 0 (round 5): a lane owns 8 consecutive elements (16 bytes) of two tile rows 32 apart, so every 128-byte line of the
// tile is read / written by 8 lanes of ONE instruction.  (The generic kernel above gives a lane 16 elements as four 8-byte pieces 32 bytes apart: each line is
// assembled from four partial accesses -- the same pattern cost weight_images_kernel a third of its time.)
template <bool ROWS2COLS>
__global__ __launch_bounds__(256) void transpose8_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int64_t S, int64_t C, int Wp, int L) {
    __shared__ float tile[64][65];      // [channel][position]
    const int64_t bt = blockIdx.z;
    const int64_t s0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
    const bf16_t* ib = in + bt * S * C;
    bf16_t* ob = out + bt * S * C;
    const int t = threadIdx.x, ra = t >> 3, cg = (t & 7) * 8;
    float x[8];
    if (ROWS2COLS) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t sp = s0 + ra + 32 * h;
            if (sp < S && c0 + cg < C) {
                load8(ib + pixel_to_row(sp, Wp, L) * C + c0 + cg, x);
#pragma unroll
                for (int e = 0; e < 8; ++e) tile[cg + e][ra + 32 * h] = x[e];      // bank = cg + e + ra (+ 32 h): distinct over a wave
            }
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t c = c0 + ra + 32 * h;
            if (c < C && s0 + cg < S) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = tile[ra + 32 * h][cg + e];
                store8(ob + c * S + s0 + cg, x);
            }
        }
    } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t c = c0 + ra + 32 * h;
            if (c < C && s0 + cg < S) {
                load8(ib + c * S + s0 + cg, x);
#pragma unroll
                for (int e = 0; e < 8; ++e) tile[ra + 32 * h][cg + e] = x[e];
            }
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t sp = s0 + ra + 32 * h;
            if (sp < S && c0 + cg < C) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = tile[cg + e][ra + 32 * h];
                store8(ob + pixel_to_row(sp, Wp, L) * C + c0 + cg, x);
            }
        }
    }
}

template <bool ROWS2COLS>
int launch_transpose(const void* in, int in_dt, void* out, int out_dt, int64_t batches, int64_t S, int64_t C, int Wp, int L, hipStream_t s) {
    if (C % 4) return MTP_ERR_ARG;
    dim3 grid((unsigned)((S + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)batches), block(256);
    if (in_dt == MTP_BF16 && out_dt == MTP_BF16 && !(S & 7) && !(C & 7) && !(((uintptr_t)in | (uintptr_t)out) & 15)) {
        hipLaunchKernelGGL((transpose8_bf16_kernel<ROWS2COLS>), grid, block, 0, s, (const bf16_t*)in, (bf16_t*)out, S, C, Wp, L);
        return mtp_launch_status();
    }
#define MTP_TR(TI, TO) hipLaunchKernelGGL((transpose_kernel<TI, TO, ROWS2COLS>), grid, block, 0, s, (const TI*)in, (TO*)out, S, C, Wp, L)
    if (in_dt == MTP_F32 && out_dt == MTP_F32) MTP_TR(float, float);
    else if (in_dt == MTP_F32 && out_dt == MTP_BF16) MTP_TR(float, bf16_t);
    else if (in_dt == MTP_BF16 && out_dt == MTP_F32) MTP_TR(bf16_t, float);
    else if (in_dt == MTP_BF16 && out_dt == MTP_BF16) MTP_TR(bf16_t, bf16_t);
    else return MTP_ERR_UNSUPPORTED;
#undef MTP_TR
    return mtp_launch_status();
}

// ------------------------------------------------------------------------------------------------ ConvTranspose2d weight packing
// w (Cin, Cout, 2, 2) -> wg[(q*Cout + co)][ci], wgT[ci][q*Cout + co], q = ky*2+kx
template <typename T>
__global__ __launch_bounds__(256) void convt_pack_kernel(const float* __restrict__ w, T* __restrict__ wg, T* __restrict__ wgT, int64_t Cin, int64_t Cout) {
    const int64_t total = Cin * Cout * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i & 3);
        const int64_t co = (i >> 2) % Cout, ci = (i >> 2) / Cout;
        const float v = w[i];
        if (wg) Elem<T>::store(wg + (q * Cout + co) * Cin + ci, v);
        if (wgT) Elem<T>::store(wgT + ci * (4 * Cout) + q * Cout + co, v);
    }
}
__global__ __launch_bounds__(256) void convt_unpack_kernel(const float* __restrict__ dwg, float* __restrict__ dw, int64_t Cin, int64_t Cout) {
    const int64_t total = Cin * Cout * 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i & 3);
        const int64_t co = (i >> 2) % Cout, ci = (i >> 2) / Cout;
        dw[i] = dwg[(q * Cout + co) * Cin + ci];
    }
}

// The same two packings through a (64 ci) x (64 co) tile in LDS, two taps at a time (round 5): every global access is a whole line -- the float4 of the four taps
// of (ci, co) on the parameter side (1 KiB per wave instruction), 8 lanes x 16 bytes per row of the images.  (The element-wise kernels above write 2-byte pieces
// scattered over the images: 31 us for a 1024 x 1024 weight.)  Cin, Cout multiples of 8.
template <typename T>
__global__ __launch_bounds__(256) void convt_pack_tiled_kernel(const float* __restrict__ w, T* __restrict__ wg, T* __restrict__ wgT, int Cin, int Cout) {
    __shared__ float tile[2][64][65];      // [tap of the pair][ci][co]
    const int ci0 = blockIdx.y * 64, co0 = blockIdx.x * 64;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, ra = t >> 3, cg = (t & 7) * 8;
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ci = ci0 + wave + 4 * k, co = co0 + lane;
        v[k] = (ci < Cin && co < Cout) ? load4(w + ((int64_t)ci * Cout + co) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            tile[0][wave + 4 * k][lane] = half ? v[k].z : v[k].x;
            tile[1][wave + 4 * k][lane] = half ? v[k].w : v[k].y;
        }
        __syncthreads();
        float o[8];
        if (wgT) {         // wgT[ci][q * Cout + co]: a line = (ci, q), 64 co
            for (int l = ra; l < 128; l += 32) {
                const int r = l >> 1, qq = l & 1, ci = ci0 + r;
                if (ci < Cin && co0 + cg < Cout) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = tile[qq][r][cg + e];
                    store8(wgT + (int64_t)ci * 4 * Cout + (int64_t)(2 * half + qq) * Cout + co0 + cg, o);
                }
            }
        }
        if (wg) {          // wg[q * Cout + co][ci]: a line = (q, co), 64 ci
            for (int l = ra; l < 128; l += 32) {
                const int qq = l >> 6, c = l & 63, co = co0 + c;
                if (co < Cout && ci0 + cg < Cin) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = tile[qq][cg + e][c];
                    store8(wg + ((int64_t)(2 * half + qq) * Cout + co) * Cin + ci0 + cg, o);
                }
            }
        }
    }
}
// dw[ci][co][q] = dwg[q * Cout + co][ci]
__global__ __launch_bounds__(256) void convt_unpack_tiled_kernel(const float* __restrict__ dwg, float* __restrict__ dw, int Cin, int Cout) {
    __shared__ float tile[2][64][65];      // [tap of the pair][ci][co]
    const int ci0 = blockIdx.y * 64, co0 = blockIdx.x * 64;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, rb = t >> 4, c4 = (t & 15) * 4;
    float g[16][4];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
        for (int l = rb; l < 128; l += 16) {       // a line = (q, co): 64 ci = 16 lanes x 16 bytes
            const int qq = l >> 6, c = l & 63, co = co0 + c;
            if (co < Cout && ci0 + c4 < Cin) {
                const float4 x = load4(dwg + ((int64_t)(2 * half + qq) * Cout + co) * Cin + ci0 + c4);
                tile[qq][c4 + 0][c] = x.x; tile[qq][c4 + 1][c] = x.y; tile[qq][c4 + 2][c] = x.z; tile[qq][c4 + 3][c] = x.w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            g[k][2 * half] = tile[0][wave + 4 * k][lane];
            g[k][2 * half + 1] = tile[1][wave + 4 * k][lane];
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ci = ci0 + wave + 4 * k, co = co0 + lane;
        if (ci < Cin && co < Cout) store4(dw + ((int64_t)ci * Cout + co) * 4, make_float4(g[k][0], g[k][1], g[k][2], g[k][3]));
    }
}

// ------------------------------------------------------------------------------------------------ MaxPool2d(2,2) on tokens
template <typename Tout>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, Tout* __restrict__ y, int B, int Hp, int Wp, int C) {
    const int Ho = Hp / 2, Wo = Wp / 2, C4 = C / 4;
    const int64_t total = (int64_t)B * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const int64_t o = i / C4;
        const int xo = (int)(o % Wo), yo = (int)((o / Wo) % Ho), b = (int)(o / ((int64_t)Wo * Ho));
        const float* p = x + (((int64_t)b * Hp + 2 * yo) * Wp + 2 * xo) * C + 4 * c4;
        const float4 a = load4(p), bb = load4(p + C), c = load4(p + (int64_t)Wp * C), d = load4(p + (int64_t)Wp * C + C);
        store4(y + o * C + 4 * c4, make_float4(fmaxf(fmaxf(a.x, bb.x), fmaxf(c.x, d.x)), fmaxf(fmaxf(a.y, bb.y), fmaxf(c.y, d.y)),
                                              fmaxf(fmaxf(a.z, bb.z), fmaxf(c.z, d.z)), fmaxf(fmaxf(a.w, bb.w), fmaxf(c.w, d.w))));
    }
}
// dx[token] (+)= dy[window] where token is the FIRST maximum of its 2x2 window in scan order (torch's tie rule)
template <typename Tdy>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const Tdy* __restrict__ dy, float* __restrict__ dx, int accumulate,
                                                         int B, int Hp, int Wp, int C) {
    const int Ho = Hp / 2, Wo = Wp / 2, C4 = C / 4;
    const int64_t total = (int64_t)B * Hp * Wp * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const int64_t t = i / C4;
        const int xx = (int)(t % Wp), yy = (int)((t / Wp) % Hp), b = (int)(t / ((int64_t)Wp * Hp));
        float4 g = make_float4(0, 0, 0, 0);
        if (yy < 2 * Ho && xx < 2 * Wo) {
            const int yo = yy >> 1, xo = xx >> 1, me = ((yy & 1) << 1) | (xx & 1);
            const float* p = x + (((int64_t)b * Hp + 2 * yo) * Wp + 2 * xo) * C + 4 * c4;
            float v[4][4];
            const float4 q0 = load4(p), q1 = load4(p + C), q2 = load4(p + (int64_t)Wp * C), q3 = load4(p + (int64_t)Wp * C + C);
            v[0][0] = q0.x; v[0][1] = q0.y; v[0][2] = q0.z; v[0][3] = q0.w;
            v[1][0] = q1.x; v[1][1] = q1.y; v[1][2] = q1.z; v[1][3] = q1.w;
            v[2][0] = q2.x; v[2][1] = q2.y; v[2][2] = q2.z; v[2][3] = q2.w;
            v[3][0] = q3.x; v[3][1] = q3.y; v[3][2] = q3.z; v[3][3] = q3.w;
            const float4 d = load4(dy + (((int64_t)b * Ho + yo) * Wo + xo) * C + 4 * c4);
            const float dd[4] = {d.x, d.y, d.z, d.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int arg = 0;
                float m = v[0][e];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (v[k][e] > m) { m = v[k][e]; arg = k; }
                o[e] = arg == me ? dd[e] : 0.f;
            }
            g = make_float4(o[0], o[1], o[2], o[3]);
        }
        float* dp = dx + t * C + 4 * c4;
        if (accumulate) {
            const float4 old = load4(dp);
            g.x += old.x; g.y += old.y; g.z += old.z; g.w += old.w;
        }
        store4(dp, g);
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] += alpha * x[i];
}

// ------------------------------------------------------------------------------------------------ RVSA sampling heads
// zero-pad to (He,We), AvgPool2d(7,7) (divide by 49 always), LeakyReLU(0.01)    (VIT:229-230, 347)
template <typename T>
__global__ __launch_bounds__(256) void rvsa_pool_fwd_kernel(const T* __restrict__ x, float* __restrict__ avg, float* __restrict__ pooled,
                                                           int Hp, int Wp, int C, int pad_t, int pad_l, int nh, int nw) {
    const int win = blockIdx.x;   // (b, i, j)
    const int j = win % nw, i = (win / nw) % nh, b = win / (nw * nh);
    for (int c4 = blockIdx.y * 256 + threadIdx.x; c4 < C / 4; c4 += gridDim.y * 256) {
        float4 s = make_float4(0, 0, 0, 0);
        for (int a = 0; a < 7; ++a) {
            const int y = i * 7 + a - pad_t;
            if (y < 0 || y >= Hp) continue;
            for (int bb = 0; bb < 7; ++bb) {
                const int xx = j * 7 + bb - pad_l;
                if (xx < 0 || xx >= Wp) continue;
                const float4 v = load4(x + (((int64_t)b * Hp + y) * Wp + xx) * C + 4 * c4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        const float inv = 1.0f / 49.0f;
        s = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
        store4(avg + (int64_t)win * C + 4 * c4, s);
        store4(pooled + (int64_t)win * C + 4 * c4, make_float4(s.x > 0 ? s.x : 0.01f * s.x, s.y > 0 ? s.y : 0.01f * s.y,
                                                              s.z > 0 ? s.z : 0.01f * s.z, s.w > 0 ? s.w : 0.01f * s.w));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void rvsa_pool_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ avg, T* __restrict__ dx, int accumulate,
                                                           int B, int Hp, int Wp, int C, int pad_t, int pad_l, int nh, int nw) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)B * Hp * Wp * C4;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(idx % C4);
        const int64_t t = idx / C4;
        const int xx = (int)(t % Wp), yy = (int)((t / Wp) % Hp), b = (int)(t / ((int64_t)Wp * Hp));
        const int win = (b * nh + (yy + pad_t) / 7) * nw + (xx + pad_l) / 7;
        const float4 d = load4(dpooled + (int64_t)win * C + 4 * c4), a = load4(avg + (int64_t)win * C + 4 * c4);
        const float k = 1.0f / 49.0f;
        float4 g = make_float4(d.x * (a.x > 0 ? k : 0.01f * k), d.y * (a.y > 0 ? k : 0.01f * k), d.z * (a.z > 0 ? k : 0.01f * k), d.w * (a.w > 0 ? k : 0.01f * k));
        T* p = dx + t * C + 4 * c4;
        if (accumulate) {
            const float4 o = load4(p);
            g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
        }
        store4(p, g);
    }
}

// ---- the three RVSA 1x1-conv heads as one small f32 linear layer (R = windows ~ 1e3, K = C, N = 5*heads = 80) -----------------
// Far too small for the MFMA GEMMs (8 output tiles); the kernels below are shaped so that the 320 KB weight is not re-read
// from L2 by every thread (the first versions moved ~335 MB of L2 traffic per call and took 35-39 us each).
// y (R,N) = x (R,K) W(N,K)^T + b : generic fallback, one block per row, one wave per output column group
__global__ __launch_bounds__(256) void small_linear_fwd_generic_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                      float* __restrict__ y, int N, int K) {
    const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xr = x + (int64_t)r * K;
    for (int n = wave; n < N; n += 4) {
        const float* wr = w + (int64_t)n * K;
        float s = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
            const float4 a = load4(xr + k), b = load4(wr + k);
            s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
        s = wave_sum(s);
        if (lane == 0) y[(int64_t)r * N + n] = s + (bias ? bias[n] : 0.f);
    }
}
// K <= 256*MAXJ: one workgroup per ROWS rows, wave = a quarter of the output columns, the row slices stay in registers and every
// weight vector loaded from L2 is used for ROWS rows (the one-row version re-read the whole 320 KB weight per row: 335 MB of L2
// traffic per call, 39 us); 4 output columns (4*MAXJ weight loads) in flight per pass
template <int ROWS, int MAXJ>
__global__ __launch_bounds__(256) void small_linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ y, int R, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * ROWS;
    const int K4 = K >> 2;
    float4 xs[ROWS][MAXJ];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const float* xr = x + (int64_t)(r0 + i < R ? r0 + i : R - 1) * K;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int k4 = lane + 64 * j;
            xs[i][j] = k4 < K4 ? load4(xr + 4 * k4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const int per = (N + 3) / 4, n_lo = wave * per, n_hi = (n_lo + per) < N ? (n_lo + per) : N;
    for (int n = n_lo; n < n_hi; n += 4) {
        float s[4][ROWS];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int nn = n + q < n_hi ? n + q : n_hi - 1;
            const float* wr = w + (int64_t)nn * K;
#pragma unroll
            for (int i = 0; i < ROWS; ++i) s[q][i] = 0.f;
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                // unconditional load from a clamped index (xs is zero beyond K): a branch around the load makes hipcc wait for
                // each one separately -- 4*MAXJ serialised L2 latencies per pass, measured 10 us per pass
                const int k4 = lane + 64 * j, k4c = k4 < K4 ? k4 : K4 - 1;
                const float4 b = load4(wr + 4 * k4c);
#pragma unroll
                for (int i = 0; i < ROWS; ++i) s[q][i] += xs[i][j].x * b.x + xs[i][j].y * b.y + xs[i][j].z * b.z + xs[i][j].w * b.w;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < ROWS; ++i) s[q][i] = wave_sum(s[q][i]);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < ROWS; ++i)
                    if (n + q < n_hi && r0 + i < R) y[(int64_t)(r0 + i) * N + n + q] = s[q][i] + (bias ? bias[n + q] : 0.f);
        }
    }
}
// dx (R,K) = dy (R,N) W (N,K): thread = 4 k-columns x 4 rows (the dy factors are workgroup-uniform: scalar loads), so each
// weight vector is loaded once per 4 rows
__global__ __launch_bounds__(256) void small_linear_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int R, int N, int K) {
    const int k = (blockIdx.x * 256 + threadIdx.x) * 4, r0 = blockIdx.y * 4;
    if (k >= K) return;
    const float* d[4];
    float4 s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = r0 + i < R ? r0 + i : R - 1;
        d[i] = dy + (int64_t)rr * N;
        s[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll 4
    for (int n = 0; n < N; ++n) {
        const float4 ww = load4(w + (int64_t)n * K + k);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dv = d[i][n];
            s[i].x += dv * ww.x; s[i].y += dv * ww.y; s[i].z += dv * ww.z; s[i].w += dv * ww.w;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (r0 + i < R) store4(dx + (int64_t)(r0 + i) * K + k, s[i]);
}
// dw (N,K) = dy^T x ; db[n] = sum_r dy[r][n].  Workgroup = 256 k-columns x 8 outputs n x 4*SL_DW_ROWS rows: lane = 4 k-columns,
// wave = a quarter of the rows (dy factors wave-uniform: scalar loads; each x vector feeds 8 outputs), waves combined through
// LDS, then ONE set of f32 atomics per workgroup into the zeroed outputs (the atomics were the cost of the first versions).
constexpr int SL_DW_ROWS = 32;
// SEG: the N output rows are slices of up to 4 separate parameters (the three stacked RVSA heads): row n of segment j goes to
// seg.dw[j] + (n - seg.row0[j]) * K -- accumulated straight into the parameter gradients (no stacked scratch, no clearing pass, no copy)
struct SlSegs {
    float* dw[4];
    float* db[4];
    int row0[5];
    int nseg;
};
// batched form (round 4): the same gradients of up to SL_BATCH independent problems of one shape (the stacked heads of a burst of RVSA
// blocks) in ONE launch -- blockIdx.z = problem * zsplit + row split
constexpr int SL_BATCH = 8;
struct SlBatch {
    const float* dy[SL_BATCH];
    const float* x[SL_BATCH];
    SlSegs seg[SL_BATCH];
    int zsplit;
};
template <bool SEG>
__device__ __forceinline__ void small_linear_dw_body(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ db, int R, int N, int K,
                                                     const SlSegs& seg, int zrow) {
    __shared__ float4 red[3][8][64];
    __shared__ float redb[3][8];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = (blockIdx.x * 64 + lane) * 4, n0 = blockIdx.y * 8;
    const int r0 = (zrow * 4 + wave) * SL_DW_ROWS, r1 = (r0 + SL_DW_ROWS) < R ? (r0 + SL_DW_ROWS) : R;
    const bool kok = k < K;
    const int kc = kok ? k : 0;
    int nn[8];
    float4 s[8];
    float sb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        nn[q] = n0 + q < N ? n0 + q : N - 1;
        s[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        sb[q] = 0.f;
    }
#pragma unroll 8      // (round 4: 8 rows in flight; with 2 the 32 rows of a wave were 16 serialised round trips -- the launch is latency, not bytes)
    for (int r = r0; r < r1; ++r) {
        const float4 xv = load4(x + (int64_t)r * K + kc);
        const float* dr = dy + (int64_t)r * N;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float dv = dr[nn[q]];
            s[q].x += dv * xv.x; s[q].y += dv * xv.y; s[q].z += dv * xv.z; s[q].w += dv * xv.w;
            sb[q] += dv;
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            red[wave - 1][q][lane] = s[q];
            if (lane == 0) redb[wave - 1][q] = sb[q];
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const float4 o = red[v][q][lane];
                s[q].x += o.x; s[q].y += o.y; s[q].z += o.z; s[q].w += o.w;
                sb[q] += redb[v][q];
            }
            float* dwrow = dw + (int64_t)(n0 + q) * K;
            float* dbp = db ? db + n0 + q : nullptr;
            if constexpr (SEG) {
                int j = 0;
                const int n = n0 + q < N ? n0 + q : N - 1;
#pragma unroll
                for (int t = 1; t < 4; ++t) j += (t < seg.nseg && n >= seg.row0[t]) ? 1 : 0;
                dwrow = seg.dw[j] + (int64_t)(n - seg.row0[j]) * K;
                dbp = seg.db[j] ? seg.db[j] + (n - seg.row0[j]) : nullptr;
            }
            if (kok && n0 + q < N) {
                float* o = dwrow + k;
                atomicAdd(o, s[q].x); atomicAdd(o + 1, s[q].y); atomicAdd(o + 2, s[q].z); atomicAdd(o + 3, s[q].w);
            }
            if (dbp && blockIdx.x == 0 && lane == 0 && n0 + q < N) atomicAdd(dbp, sb[q]);
        }
    }
}
template <bool SEG>
__global__ __launch_bounds__(256) void small_linear_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ db, int R, int N, int K, SlSegs seg) {
    small_linear_dw_body<SEG>(dy, x, dw, db, R, N, K, seg, (int)blockIdx.z);
}
__global__ __launch_bounds__(256) void small_linear_dw_batched_kernel(SlBatch t, int R, int N, int K) {
    const int pi = __builtin_amdgcn_readfirstlane((int)blockIdx.z / t.zsplit);
    small_linear_dw_body<true>(t.dy[pi], t.x[pi], nullptr, nullptr, R, N, K, t.seg[pi], (int)blockIdx.z - pi * t.zsplit);
}

// ------------------------------------------------------------------------------------------------ optimizer
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, float* __restrict__ out, int64_t n) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = load4(g + 4 * i);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// AdamW (torch.optim.AdamW semantics, MAIN:424-457) over a flat buffer; segments start at multiples of 4 elements.
// (round 5: nontemporal loads / stores on all seven streams: 1660 -> 1603 us alone, no difference in the step -- profiles/r05_ab_late_adamw_nontemporal.txt; not kept)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const int64_t* __restrict__ seg_start, const float* __restrict__ seg_wd, int nseg,
                                                   const float* __restrict__ hyper, const float* __restrict__ sqnorm, float max_norm, float grad_scale) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[4], bc2 = hyper[5];
    float gs = grad_scale;
    if (sqnorm) {
        const float total = sqrtf(*sqnorm) * grad_scale;
        const float coef = max_norm / (total + 1e-6f);
        gs *= coef < 1.0f ? coef : 1.0f;
    }
    const float rbc2 = rsqrtf(bc2), step = lr / bc1;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        int lo = 0, hi = nseg - 1;   // last segment with start <= 4*i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (seg_start[mid] <= 4 * i) lo = mid; else hi = mid - 1;
        }
        const float decay = 1.0f - lr * seg_wd[lo];
        float4 pv = load4(p + 4 * i), gv = load4(g + 4 * i), mv = load4(m + 4 * i), vv = load4(v + 4 * i);
        float P[4] = {pv.x, pv.y, pv.z, pv.w}, G[4] = {gv.x, gv.y, gv.z, gv.w}, M[4] = {mv.x, mv.y, mv.z, mv.w}, V[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = G[e] * gs;
            P[e] *= decay;
            M[e] = b1 * M[e] + (1.0f - b1) * ge;
            V[e] = b2 * V[e] + (1.0f - b2) * ge * ge;
            P[e] -= step * M[e] / (sqrtf(V[e]) * rbc2 + eps);
        }
        store4(p + 4 * i, make_float4(P[0], P[1], P[2], P[3]));
        store4(m + 4 * i, make_float4(M[0], M[1], M[2], M[3]));
        store4(v + 4 * i, make_float4(V[0], V[1], V[2], V[3]));
    }
}

}  // namespace

extern "C" int mtp_patchify(const float* img, void* cols, int dtype, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t P, mtp_stream_t stream) {
    if (!img || !cols || B <= 0 || (P % 4) || (W % 4) || H < P || W < P) return MTP_ERR_ARG;
    const int Hp = (int)(H / P), Wp = (int)(W / P);
    const int64_t total = B * Hp * Wp * Cin * P * P / 4;
    dim3 grid(blocks_for(total, 256, 8192)), block(256);
    if (dtype == MTP_BF16)
        hipLaunchKernelGGL((patchify_kernel<bf16_t, false>), grid, block, 0, (hipStream_t)stream, (float*)img, (bf16_t*)cols, (int)B, (int)Cin, (int)H, (int)W, (int)P, Hp, Wp);
    else
        hipLaunchKernelGGL((patchify_kernel<float, false>), grid, block, 0, (hipStream_t)stream, (float*)img, (float*)cols, (int)B, (int)Cin, (int)H, (int)W, (int)P, Hp, Wp);
    return mtp_launch_status();
}

// ---- uint8 HWC image -> normalised patch rows (data preprocessor + im2col in one pass over 1 byte per sample) ------------------
// thread = 4 consecutive pixels of one patch row: 12 input bytes (three aligned dwords when W % 4 == 0), three 4-element
// stores (one per output channel plane of the patch row).  IEEE division, so the f32 result is bit-equal to (x - mean) / std.
struct PreNorm {
    float mean[3], std[3];
};
template <typename T>
__global__ __launch_bounds__(256) void preprocess_patchify_kernel(const uint8_t* __restrict__ img, T* __restrict__ cols, int B, int H, int W, int P, int Hp, int Wp,
                                                                 PreNorm nm, int flip, float pad_value) {
    const int P4 = P / 4, K = 3 * P * P;
    const int64_t total = (int64_t)B * Hp * Wp * P * P4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x4 = (int)(i % P4), ky = (int)((i / P4) % P);
        const int64_t t = i / ((int64_t)P4 * P);
        const int px = (int)(t % Wp), py = (int)((t / Wp) % Hp), b = (int)(t / ((int64_t)Wp * Hp));
        const int y = py * P + ky, x0 = px * P + 4 * x4;
        float v[3][4];
        if (y < H && x0 + 3 < W && (W & 3) == 0) {
            const uint32_t* p = reinterpret_cast<const uint32_t*>(img + (((int64_t)b * H + y) * W + x0) * 3);
            const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
            const uint32_t byte[12] = {w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, w0 >> 24, w1 & 255u, (w1 >> 8) & 255u, (w1 >> 16) & 255u, w1 >> 24,
                                       w2 & 255u, (w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c][j] = ((float)byte[3 * j + (flip ? 2 - c : c)] - nm.mean[c]) / nm.std[c];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = y < H && x0 + j < W;
                const uint8_t* p = img + (((int64_t)b * H + (in ? y : 0)) * W + (in ? x0 + j : 0)) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[c][j] = in ? ((float)p[flip ? 2 - c : c] - nm.mean[c]) / nm.std[c] : pad_value;
            }
        }
        T* cp = cols + t * K + ky * P + 4 * x4;
#pragma unroll
        for (int c = 0; c < 3; ++c) store4(cp + c * P * P, make_float4(v[c][0], v[c][1], v[c][2], v[c][3]));
    }
}

extern "C" int mtp_preprocess_patchify(const uint8_t* img, void* cols, int dtype, int64_t B, int64_t H, int64_t W, int64_t P, int64_t pad_divisor,
                                       const float* mean, const float* std, int bgr_to_rgb, float pad_value, mtp_stream_t stream) {
    if (!img || !cols || !mean || !std || B <= 0 || H <= 0 || W <= 0 || P <= 0 || (P % 4) || pad_divisor <= 0) return MTP_ERR_ARG;
    const int64_t Hpad = (H + pad_divisor - 1) / pad_divisor * pad_divisor, Wpad = (W + pad_divisor - 1) / pad_divisor * pad_divisor;
    if ((Hpad % P) || (Wpad % P)) return MTP_ERR_ARG;
    PreNorm nm;
    for (int c = 0; c < 3; ++c) {
        if (std[c] == 0.f) return MTP_ERR_ARG;
        nm.mean[c] = mean[c];
        nm.std[c] = std[c];
    }
    const int Hp = (int)(Hpad / P), Wp = (int)(Wpad / P);
    const int64_t total = B * Hp * Wp * P * (P / 4);
    dim3 grid(blocks_for(total, 256, 8192)), block(256);
    if (dtype == MTP_BF16)
        hipLaunchKernelGGL((preprocess_patchify_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, img, (bf16_t*)cols, (int)B, (int)H, (int)W, (int)P, Hp, Wp, nm, bgr_to_rgb, pad_value);
    else if (dtype == MTP_F32)
        hipLaunchKernelGGL((preprocess_patchify_kernel<float>), grid, block, 0, (hipStream_t)stream, img, (float*)cols, (int)B, (int)H, (int)W, (int)P, Hp, Wp, nm, bgr_to_rgb, pad_value);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_unpatchify(const void* cols, int dtype, float* dimg, int64_t B, int64_t Cin, int64_t H, int64_t W, int64_t P, mtp_stream_t stream) {
    if (!dimg || !cols || B <= 0 || (P % 4) || (W % 4) || H < P || W < P) return MTP_ERR_ARG;
    const int Hp = (int)(H / P), Wp = (int)(W / P);
    hipStream_t s = (hipStream_t)stream;
    if ((H % P) || (W % P)) {
        hipError_t e = hipMemsetAsync(dimg, 0, sizeof(float) * (size_t)(B * Cin * H * W), s);
        if (e != hipSuccess) return (int)e;
    }
    const int64_t total = B * Hp * Wp * Cin * P * P / 4;
    dim3 grid(blocks_for(total, 256, 8192)), block(256);
    if (dtype == MTP_BF16)
        hipLaunchKernelGGL((patchify_kernel<bf16_t, true>), grid, block, 0, s, dimg, (bf16_t*)cols, (int)B, (int)Cin, (int)H, (int)W, (int)P, Hp, Wp);
    else
        hipLaunchKernelGGL((patchify_kernel<float, true>), grid, block, 0, s, dimg, (float*)cols, (int)B, (int)Cin, (int)H, (int)W, (int)P, Hp, Wp);
    return mtp_launch_status();
}

extern "C" int mtp_cast(const void* src, int sd, void* dst, int dd, int64_t n, mtp_stream_t stream) {
    if (!src || !dst || n <= 0) return MTP_ERR_ARG;
    dim3 grid(blocks_for(n / 4 + 1, 256, 8192)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (sd == MTP_F32 && dd == MTP_BF16) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, block, 0, s, (const float*)src, (bf16_t*)dst, n);
    else if (sd == MTP_BF16 && dd == MTP_F32) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)src, (float*)dst, n);
    else if (sd == MTP_F32 && dd == MTP_F32) hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, s, (const float*)src, (float*)dst, n);
    else if (sd == MTP_BF16 && dd == MTP_BF16) hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_transpose_cast(const float* src, void* dst, int dst_dtype, int64_t R, int64_t C, mtp_stream_t stream) {
    if (!src || !dst || R <= 0 || C <= 0) return MTP_ERR_ARG;
    // src (R rows of C) is the "row side": out[c][r] = in[r][c]
    return launch_transpose<true>(src, MTP_F32, dst, dst_dtype, 1, R, C, 1, 0, (hipStream_t)stream);
}

// ---- every weight image of the model in one launch: workgroup = one 64x64 tile of one matrix (descriptor table in HBM)
template <typename T>
__global__ __launch_bounds__(256) void weight_images_kernel(const mtp_wimg_desc* __restrict__ descs, int n) {
    __shared__ float tile[64][65];
    const int64_t tl = blockIdx.x;
    int lo = 0, hi = n - 1;   // last descriptor with tile0 <= tl (uniform over the workgroup: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile0 <= tl) lo = mid; else hi = mid - 1;
    }
    const mtp_wimg_desc d = descs[lo];
    const int64_t local = tl - d.tile0, tc = (d.C + 63) / 64;
    const int64_t r0 = (local / tc) * 64, c0 = (local % tc) * 64;
    const int t = threadIdx.x, a = t >> 2, g = (t & 3) * 16;
    const int64_t r = r0 + a;
    const bool f32o = d.f32_out != 0;
    const int am = (f32o || sizeof(T) == 4) ? 3 : 7;     // 16-byte stores: 4 floats or 8 bf16 per lane
    if ((d.C & am) || (d.wt && (d.R & am))) {   // odd-sized (tiny) matrices: element-wise
        for (int e = 0; e < 16; ++e) {
            const int64_t c = c0 + g + e;
            if (r < d.R && c < d.C) {
                const float v = d.src[r * d.C + c];
                if (d.w) { if (f32o) reinterpret_cast<float*>(d.w)[r * d.C + c] = v; else Elem<T>::store(reinterpret_cast<T*>(d.w) + r * d.C + c, v); }
                if (d.wt) { if (f32o) reinterpret_cast<float*>(d.wt)[c * d.R + r] = v; else Elem<T>::store(reinterpret_cast<T*>(d.wt) + c * d.R + r, v); }
            }
        }
        return;
    }
    // vector path (round 5): a lane owns 8 consecutive columns of two rows (32 apart), so a row of the tile is ONE 128-byte (bf16) line written by 8 lanes of one
    // instruction -- and likewise a row of the transposed tile.  (Rounds 1-4: 16 columns per lane as four 8-byte bf16 stores 32 bytes apart: every line of
    // both images was assembled from four partial writes -- 1.26 x the algorithmic bytes at the L2 boundary, 0.51 of the HBM peak.)
    const int ra = t >> 3, cg = (t & 7) * 8;
    float x[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t rr = r0 + ra + 32 * h, c = c0 + cg;
        const bool ok = rr < d.R && c < d.C;         // (C % 4 == 0: columns c .. c + 3 are in range; c + 4 .. c + 7 checked separately)
        const bool ok2 = ok && c + 4 < d.C;
        const float4 v0 = ok ? load4(d.src + rr * d.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v1 = ok2 ? load4(d.src + rr * d.C + c + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        x[h][0] = v0.x; x[h][1] = v0.y; x[h][2] = v0.z; x[h][3] = v0.w; x[h][4] = v1.x; x[h][5] = v1.y; x[h][6] = v1.z; x[h][7] = v1.w;
        if (ok && d.w) {
            if (ok2) {
                if (f32o) store8(reinterpret_cast<float*>(d.w) + rr * d.C + c, x[h]);
                else store8(reinterpret_cast<T*>(d.w) + rr * d.C + c, x[h]);
            } else {
                if (f32o) store4(reinterpret_cast<float*>(d.w) + rr * d.C + c, v0);
                else store4(reinterpret_cast<T*>(d.w) + rr * d.C + c, v0);
            }
        }
    }
    if (d.wt) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[cg + e][ra + 32 * h] = x[h][e];      // bank = cg + e + ra (+ 32 h): distinct over the 64 lanes
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cl = ra + 32 * h;                   // source column = image row
            const int64_t c = c0 + cl, rr = r0 + cg;
            if (c < d.C && rr < d.R) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = tile[cl][cg + e];
                if (rr + 4 < d.R) {
                    if (f32o) store8(reinterpret_cast<float*>(d.wt) + c * d.R + rr, o);
                    else store8(reinterpret_cast<T*>(d.wt) + c * d.R + rr, o);
                } else {                                  // (R % 4 == 0)
                    if (f32o) store4(reinterpret_cast<float*>(d.wt) + c * d.R + rr, make_float4(o[0], o[1], o[2], o[3]));
                    else store4(reinterpret_cast<T*>(d.wt) + c * d.R + rr, make_float4(o[0], o[1], o[2], o[3]));
                }
            }
        }
    }
}

// ---- AdamW of the whole flat buffer AND every GEMM-side weight image in one launch (round 6): the update of a 64 x 64 tile of a parameter matrix is followed, from
// the same registers, by the tile's bf16 row-major image and (through LDS) its transpose.  The separate pass (weight_images_kernel) read every f32 master once more:
// 4 of its 8 bytes per GEMM weight, 1.2 GB per ViT-L step.  Descriptors as for mtp_weight_images, one per parameter of the flat buffers (1-D parameters as rows of 64
// with no images), `src` = the parameter inside the flat data buffer, `wd` = its weight decay; g / m / v live at the same offset of their flat buffers.
__device__ __forceinline__ void adamw_elem(float& P, float G, float& M, float& V, float gs, float decay, float b1, float b2, float step, float rbc2, float eps) {
    const float ge = G * gs;
    P *= decay;
    M = b1 * M + (1.0f - b1) * ge;
    V = b2 * V + (1.0f - b2) * ge * ge;
    P -= step * M / (sqrtf(V) * rbc2 + eps);
}

template <typename T>
__global__ __launch_bounds__(256) void adamw_images_kernel(const mtp_wimg_desc* __restrict__ descs, int n, const float* __restrict__ p_base, const float* __restrict__ g_base,
                                                          float* __restrict__ m_base, float* __restrict__ v_base, const float* __restrict__ hyper,
                                                          const float* __restrict__ sqnorm, float max_norm, float grad_scale) {
    __shared__ float tile[64][65];
    const int64_t tl = blockIdx.x;
    int lo = 0, hi = n - 1;   // last descriptor with tile0 <= tl (uniform over the workgroup: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile0 <= tl) lo = mid; else hi = mid - 1;
    }
    const mtp_wimg_desc d = descs[lo];
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[4], bc2 = hyper[5];
    float gs = grad_scale;
    if (sqnorm) {
        const float total = sqrtf(*sqnorm) * grad_scale;
        const float coef = max_norm / (total + 1e-6f);
        gs *= coef < 1.0f ? coef : 1.0f;
    }
    const float rbc2 = rsqrtf(bc2), step = lr / bc1, decay = 1.0f - lr * d.wd;
    float* __restrict__ P = const_cast<float*>(d.src);
    const int64_t off = d.src - p_base;
    const float* __restrict__ G = g_base + off;
    float* __restrict__ M = m_base + off;
    float* __restrict__ V = v_base + off;
    const int64_t local = tl - d.tile0, tc = (d.C + 63) / 64;
    const int64_t r0 = (local / tc) * 64, c0 = (local % tc) * 64;
    const int t = threadIdx.x;
    const bool f32o = d.f32_out != 0;
    const int am = (f32o || sizeof(T) == 4) ? 3 : 7;     // 16-byte image stores: 4 floats or 8 bf16 per lane
    if ((d.C & 3) || ((d.w || d.wt) && (d.C & am)) || (d.wt && (d.R & am))) {   // odd-sized (tiny) matrices: element-wise
        const int a = t >> 2, g = (t & 3) * 16;
        const int64_t r = r0 + a;
        for (int e = 0; e < 16; ++e) {
            const int64_t c = c0 + g + e;
            if (r < d.R && c < d.C) {
                const int64_t i = r * d.C + c;
                float pv = P[i], mv = M[i], vv = V[i];
                adamw_elem(pv, G[i], mv, vv, gs, decay, b1, b2, step, rbc2, eps);
                P[i] = pv; M[i] = mv; V[i] = vv;
                if (d.w) { if (f32o) reinterpret_cast<float*>(d.w)[i] = pv; else Elem<T>::store(reinterpret_cast<T*>(d.w) + i, pv); }
                if (d.wt) { if (f32o) reinterpret_cast<float*>(d.wt)[c * d.R + r] = pv; else Elem<T>::store(reinterpret_cast<T*>(d.wt) + c * d.R + r, pv); }
            }
        }
        return;
    }
    // a lane owns 8 consecutive columns of two rows (32 apart): whole 128-byte lines of both bf16 images per store instruction (weight_images_kernel)
    const int ra = t >> 3, cg = (t & 7) * 8;
    float x[2][8];
    float4 pv[2][2], gv[2][2], mv[2][2], vv[2][2];
    bool okv[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {      // all 16 loads of the lane in flight before the first use
        const int64_t rr = r0 + ra + 32 * h, c = c0 + cg;
        okv[h][0] = rr < d.R && c < d.C;         // (C % 4 == 0: columns c .. c + 3 are in range; c + 4 .. c + 7 checked separately)
        okv[h][1] = okv[h][0] && c + 4 < d.C;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t i = okv[h][q] ? rr * d.C + c + 4 * q : 0;
            pv[h][q] = load4(P + i); gv[h][q] = load4(G + i); mv[h][q] = load4(M + i); vv[h][q] = load4(V + i);
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t rr = r0 + ra + 32 * h, c = c0 + cg;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float Pq[4] = {pv[h][q].x, pv[h][q].y, pv[h][q].z, pv[h][q].w}, Mq[4] = {mv[h][q].x, mv[h][q].y, mv[h][q].z, mv[h][q].w};
            float Vq[4] = {vv[h][q].x, vv[h][q].y, vv[h][q].z, vv[h][q].w};
            const float Gq[4] = {gv[h][q].x, gv[h][q].y, gv[h][q].z, gv[h][q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) adamw_elem(Pq[e], Gq[e], Mq[e], Vq[e], gs, decay, b1, b2, step, rbc2, eps);
            if (okv[h][q]) {
                const int64_t i = rr * d.C + c + 4 * q;
                store4(P + i, make_float4(Pq[0], Pq[1], Pq[2], Pq[3]));
                store4(M + i, make_float4(Mq[0], Mq[1], Mq[2], Mq[3]));
                store4(V + i, make_float4(Vq[0], Vq[1], Vq[2], Vq[3]));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) x[h][4 * q + e] = okv[h][q] ? Pq[e] : 0.f;
        }
        if (okv[h][0] && d.w) {
            if (okv[h][1]) {
                if (f32o) store8(reinterpret_cast<float*>(d.w) + rr * d.C + c, x[h]);
                else store8(reinterpret_cast<T*>(d.w) + rr * d.C + c, x[h]);
            } else {
                if (f32o) store4(reinterpret_cast<float*>(d.w) + rr * d.C + c, make_float4(x[h][0], x[h][1], x[h][2], x[h][3]));
                else store4(reinterpret_cast<T*>(d.w) + rr * d.C + c, make_float4(x[h][0], x[h][1], x[h][2], x[h][3]));
            }
        }
    }
    if (d.wt) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) tile[cg + e][ra + 32 * h] = x[h][e];
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int cl = ra + 32 * h;                   // source column = image row
            const int64_t c = c0 + cl, rr = r0 + cg;
            if (c < d.C && rr < d.R) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = tile[cl][cg + e];
                if (rr + 4 < d.R) {
                    if (f32o) store8(reinterpret_cast<float*>(d.wt) + c * d.R + rr, o);
                    else store8(reinterpret_cast<T*>(d.wt) + c * d.R + rr, o);
                } else {                                  // (R % 4 == 0)
                    if (f32o) store4(reinterpret_cast<float*>(d.wt) + c * d.R + rr, make_float4(o[0], o[1], o[2], o[3]));
                    else store4(reinterpret_cast<T*>(d.wt) + c * d.R + rr, make_float4(o[0], o[1], o[2], o[3]));
                }
            }
        }
    }
}

extern "C" int mtp_adamw_weight_images(const mtp_wimg_desc* descs_dev, int n, int64_t total_tiles, int act_dtype, float* p_base, const float* g_base, float* m_base,
                                       float* v_base, const float* hyper, const float* sqnorm, float max_norm, float grad_scale, mtp_stream_t stream) {
    if (!descs_dev || n <= 0 || total_tiles <= 0 || total_tiles > INT32_MAX || !p_base || !g_base || !m_base || !v_base || !hyper) return MTP_ERR_ARG;
    if (act_dtype == MTP_BF16)
        hipLaunchKernelGGL((adamw_images_kernel<bf16_t>), dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, descs_dev, n, p_base, g_base, m_base, v_base, hyper,
                           sqnorm, max_norm, grad_scale);
    else if (act_dtype == MTP_F32)
        hipLaunchKernelGGL((adamw_images_kernel<float>), dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, descs_dev, n, p_base, g_base, m_base, v_base, hyper,
                           sqnorm, max_norm, grad_scale);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_weight_images(const mtp_wimg_desc* descs_dev, int n, int64_t total_tiles, int act_dtype, mtp_stream_t stream) {
    if (!descs_dev || n <= 0 || total_tiles <= 0 || total_tiles > INT32_MAX) return MTP_ERR_ARG;
    if (act_dtype == MTP_BF16)
        hipLaunchKernelGGL((weight_images_kernel<bf16_t>), dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, descs_dev, n);
    else if (act_dtype == MTP_F32)
        hipLaunchKernelGGL((weight_images_kernel<float>), dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, descs_dev, n);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_convt_pack(const float* w, void* wg, void* wgT, int dtype, int64_t Cin, int64_t Cout, mtp_stream_t stream) {
    if (!w || Cin <= 0 || Cout <= 0) return MTP_ERR_ARG;
    if (!(Cin & 7) && !(Cout & 7) && Cin < (1 << 20) && Cout < (1 << 20) && (dtype == MTP_BF16 || dtype == MTP_F32)) {
        const dim3 tg((unsigned)((Cout + 63) / 64), (unsigned)((Cin + 63) / 64));
        if (dtype == MTP_BF16)
            hipLaunchKernelGGL((convt_pack_tiled_kernel<bf16_t>), tg, dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)wg, (bf16_t*)wgT, (int)Cin, (int)Cout);
        else
            hipLaunchKernelGGL((convt_pack_tiled_kernel<float>), tg, dim3(256), 0, (hipStream_t)stream, w, (float*)wg, (float*)wgT, (int)Cin, (int)Cout);
        return mtp_launch_status();
    }
    dim3 grid(blocks_for(Cin * Cout * 4, 256, 4096)), block(256);
    if (dtype == MTP_BF16)
        hipLaunchKernelGGL((convt_pack_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, w, (bf16_t*)wg, (bf16_t*)wgT, Cin, Cout);
    else
        hipLaunchKernelGGL((convt_pack_kernel<float>), grid, block, 0, (hipStream_t)stream, w, (float*)wg, (float*)wgT, Cin, Cout);
    return mtp_launch_status();
}

extern "C" int mtp_convt_unpack_grad(const float* dwg, float* dw, int64_t Cin, int64_t Cout, mtp_stream_t stream) {
    if (!dwg || !dw || Cin <= 0 || Cout <= 0) return MTP_ERR_ARG;
    if (!(Cin & 7) && !(Cout & 7) && Cin < (1 << 20) && Cout < (1 << 20)) {
        hipLaunchKernelGGL(convt_unpack_tiled_kernel, dim3((unsigned)((Cout + 63) / 64), (unsigned)((Cin + 63) / 64)), dim3(256), 0, (hipStream_t)stream, dwg, dw, (int)Cin, (int)Cout);
        return mtp_launch_status();
    }
    hipLaunchKernelGGL(convt_unpack_kernel, dim3(blocks_for(Cin * Cout * 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, dwg, dw, Cin, Cout);
    return mtp_launch_status();
}

extern "C" int mtp_tokens_to_nchw(const void* x, int x_dtype, void* out, int out_dtype, int64_t B, int64_t Hp, int64_t Wp, int64_t C, int levels, mtp_stream_t stream) {
    if (!x || !out || B <= 0 || Hp <= 0 || Wp <= 0 || levels < 0 || levels > 4) return MTP_ERR_ARG;
    return launch_transpose<true>(x, x_dtype, out, out_dtype, B, (Hp * Wp) << (2 * levels), C, (int)Wp, levels, (hipStream_t)stream);
}
extern "C" int mtp_nchw_to_tokens(const void* f, int f_dtype, void* out, int out_dtype, int64_t B, int64_t Hp, int64_t Wp, int64_t C, int levels, mtp_stream_t stream) {
    if (!f || !out || B <= 0 || Hp <= 0 || Wp <= 0 || levels < 0 || levels > 4) return MTP_ERR_ARG;
    return launch_transpose<false>(f, f_dtype, out, out_dtype, B, (Hp * Wp) << (2 * levels), C, (int)Wp, levels, (hipStream_t)stream);
}

extern "C" int mtp_maxpool2_tokens_fwd(const float* x, void* y, int y_dtype, int64_t B, int64_t Hp, int64_t Wp, int64_t C, mtp_stream_t stream) {
    if (!x || !y || B <= 0 || Hp < 2 || Wp < 2 || (C % 4)) return MTP_ERR_ARG;
    dim3 grid(blocks_for(B * (Hp / 2) * (Wp / 2) * C / 4, 256, 8192)), block(256);
    if (y_dtype == MTP_BF16)
        hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, x, (bf16_t*)y, (int)B, (int)Hp, (int)Wp, (int)C);
    else
        hipLaunchKernelGGL((maxpool_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, x, (float*)y, (int)B, (int)Hp, (int)Wp, (int)C);
    return mtp_launch_status();
}
extern "C" int mtp_maxpool2_tokens_bwd(const float* x, const void* dy, int dy_dtype, float* dx, int accumulate, int64_t B, int64_t Hp, int64_t Wp, int64_t C, mtp_stream_t stream) {
    if (!x || !dy || !dx || B <= 0 || Hp < 2 || Wp < 2 || (C % 4)) return MTP_ERR_ARG;
    dim3 grid(blocks_for(B * Hp * Wp * C / 4, 256, 8192)), block(256);
    if (dy_dtype == MTP_BF16)
        hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, x, (const bf16_t*)dy, dx, accumulate, (int)B, (int)Hp, (int)Wp, (int)C);
    else
        hipLaunchKernelGGL((maxpool_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, x, (const float*)dy, dx, accumulate, (int)B, (int)Hp, (int)Wp, (int)C);
    return mtp_launch_status();
}

extern "C" int mtp_axpy_f32(float* y, const float* x, float alpha, int64_t n, mtp_stream_t stream) {
    if (!y || !x || n <= 0) return MTP_ERR_ARG;
    hipLaunchKernelGGL(axpy_kernel, dim3(blocks_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, y, x, alpha, n);
    return mtp_launch_status();
}

// several small independent copies in one launch (blockIdx.y = segment); the table travels in the kernel arguments
struct CopySegs {
    const float* src[MTP_MAX_SEGMENTS];
    float* dst[MTP_MAX_SEGMENTS];
    int64_t count[MTP_MAX_SEGMENTS];
};
__global__ __launch_bounds__(256) void copy_segments_kernel(CopySegs t) {
    const int sgm = blockIdx.y;
    const float* __restrict__ s = t.src[sgm];
    float* __restrict__ d = t.dst[sgm];
    const int64_t n = t.count[sgm];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d[i] = s[i];
}
extern "C" int mtp_copy_segments_f32(const float* const* src, float* const* dst, const int64_t* count, int n, mtp_stream_t stream) {
    if (!src || !dst || !count || n <= 0 || n > MTP_MAX_SEGMENTS) return MTP_ERR_ARG;
    CopySegs t;
    int64_t mx = 0;
    for (int i = 0; i < n; ++i) {
        if (!src[i] || !dst[i] || count[i] <= 0) return MTP_ERR_ARG;
        t.src[i] = src[i]; t.dst[i] = dst[i]; t.count[i] = count[i];
        mx = count[i] > mx ? count[i] : mx;
    }
    hipLaunchKernelGGL(copy_segments_kernel, dim3(blocks_for(mx, 256, 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
    return mtp_launch_status();
}

static void rvsa_geom(int64_t Hp, int64_t Wp, int& pt, int& pl, int& nh, int& nw) {
    const int pad_h = (int)((7 - Hp % 7) % 7), pad_w = (int)((7 - Wp % 7) % 7);
    pt = pad_h / 2; pl = pad_w / 2;
    nh = (int)((Hp + pad_h) / 7); nw = (int)((Wp + pad_w) / 7);
}

// ---- the sampling heads of one RVSA block in ONE launch each way (VIT:344-358: zero pad, AvgPool2d(7, 7), LeakyReLU, three 1x1
// convolutions stacked as one (N = 5 * heads) x C linear layer).  One workgroup per window: the 49 token rows are averaged with all
// loads of a window row in flight (no branch around a load), the pooled vector stays in LDS, each wave produces a quarter of the N
// outputs.  Backward: dpooled = dsamp . W per window, times leaky'(avg) / 49, added to the 49 token rows of dx.
// (separately: pool 16.5 us + linear 26.8 us, linear-dx 12.6 us + pool-backward 19.4 us per block at ViT-L, B = 64)
template <typename T>
__global__ __launch_bounds__(256) void rvsa_sampling_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ avg, float* __restrict__ pooled, float* __restrict__ samp,
                                                               int Hp, int Wp, int C, int N, int pad_t, int pad_l, int nh, int nw) {
    extern __shared__ __attribute__((aligned(16))) float pl[];     // pooled row of this window
    const int win = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = win % nw, i = (win / nw) % nh, b = win / (nw * nh);
    for (int c4 = threadIdx.x; c4 < C / 4; c4 += 256) {
        float4 s = make_float4(0, 0, 0, 0);
#pragma unroll      // (round 4: all 49 row loads of the window in flight -- with `unroll 1` the seven rows were seven serialised round trips, most of the kernel's 17.6 us)
        for (int a = 0; a < 7; ++a) {
            const int y = i * 7 + a - pad_t;
            const bool yok = y >= 0 && y < Hp;
            const int yc = yok ? y : 0;
            float4 v[7];
#pragma unroll
            for (int bb = 0; bb < 7; ++bb) {
                const int xx = j * 7 + bb - pad_l;
                const int xc = (xx >= 0 && xx < Wp) ? xx : 0;
                v[bb] = load4(x + (((int64_t)b * Hp + yc) * Wp + xc) * C + 4 * c4);
            }
#pragma unroll
            for (int bb = 0; bb < 7; ++bb) {
                const int xx = j * 7 + bb - pad_l;
                const float m = (yok && xx >= 0 && xx < Wp) ? 1.f : 0.f;
                s.x += m * v[bb].x; s.y += m * v[bb].y; s.z += m * v[bb].z; s.w += m * v[bb].w;
            }
        }
        const float inv = 1.0f / 49.0f;
        s = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
        const float4 p = make_float4(s.x > 0 ? s.x : 0.01f * s.x, s.y > 0 ? s.y : 0.01f * s.y, s.z > 0 ? s.z : 0.01f * s.z, s.w > 0 ? s.w : 0.01f * s.w);
        if (blockIdx.y == 0) {
            store4(avg + (int64_t)win * C + 4 * c4, s);
            store4(pooled + (int64_t)win * C + 4 * c4, p);
        }
        *reinterpret_cast<float4*>(pl + 4 * c4) = p;
    }
    __syncthreads();
    // gridDim.y workgroups share a window: each pools it (the second reads come out of L2) and produces its share of the N outputs
    const int nper = ((N + (int)gridDim.y - 1) / (int)gridDim.y + 3) / 4 * 4, nlo = (int)blockIdx.y * nper, nhi = (nlo + nper) < N ? (nlo + nper) : N;
    for (int n0 = nlo + 4 * wave; n0 < nhi; n0 += 16) {      // 4 output columns per pass
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // 4 k-steps x 4 outputs = 16 weight loads of 16 B in flight per lane before the first use (round 6: written as one k-step per iteration the loop bound is a
        // run-time value, hipcc kept the iterations apart and a pass was C / 256 dependent L2 round trips -- most of the kernel's 21.6 us at C = 1024)
        for (int k0 = lane * 4; k0 < C; k0 += 1024) {
            float4 ww[4][4], a[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = k0 + 256 * kk, kc = k < C ? k : 0;
                a[kk] = k < C ? *reinterpret_cast<const float4*>(pl + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + q < N ? n0 + q : N - 1;
                    ww[kk][q] = load4(w + (int64_t)n * C + kc);
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] += a[kk].x * ww[kk][q].x + a[kk].y * ww[kk][q].y + a[kk].z * ww[kk][q].z + a[kk].w * ww[kk][q].w;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t = wave_sum(acc[q]);
            if (lane == 0 && n0 + q < nhi) samp[(int64_t)win * N + n0 + q] = t + (bias ? bias[n0 + q] : 0.f);
        }
    }
}

// workgroup = (window, 64 channel quads); its 4 waves split the N head outputs of the dsamp . w product (N / 4 dependent-free weight
// loads each instead of N: with one quad per thread and the whole product in it the launch had one latency-bound wave per SIMD),
// meet in LDS, then split the 7 window rows of the dx update
template <typename T>
__global__ __launch_bounds__(256) void rvsa_sampling_bwd_kernel(const float* __restrict__ dsamp, const float* __restrict__ w, const float* __restrict__ avg,
                                                               T* __restrict__ dx, int Hp, int Wp, int C, int N, int pad_t, int pad_l, int nh, int nw) {
    extern __shared__ __attribute__((aligned(16))) float ds[];     // dsamp row of this window
    __shared__ float4 part[4][64];
    const int win = blockIdx.x;
    const int j = win % nw, i = (win / nw) % nh, b = win / (nw * nh);
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c4 = blockIdx.y * 64 + lane;
    const bool live = c4 < C / 4;
    for (int n = threadIdx.x; n < N; n += 256) ds[n] = dsamp[(int64_t)win * N + n];
    __syncthreads();
    const int nq = (N + 3) / 4, n0 = q * nq, n1 = (n0 + nq) < N ? (n0 + nq) : N;
    float4 d = make_float4(0, 0, 0, 0);
    if (live) {
#pragma unroll 4
        for (int n = n0; n < n1; ++n) {
            const float4 ww = load4(w + (int64_t)n * C + 4 * c4);
            const float t = ds[n];
            d.x += t * ww.x; d.y += t * ww.y; d.z += t * ww.z; d.w += t * ww.w;
        }
    }
    part[q][lane] = d;
    __syncthreads();
    if (!live) return;
    const float4 p0 = part[0][lane], p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];   // fixed order: every wave gets the same bits
    d = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
    const float4 a = load4(avg + (int64_t)win * C + 4 * c4);
    const float k = 1.0f / 49.0f;
    const float4 g = make_float4(d.x * (a.x > 0 ? k : 0.01f * k), d.y * (a.y > 0 ? k : 0.01f * k), d.z * (a.z > 0 ? k : 0.01f * k), d.w * (a.w > 0 ? k : 0.01f * k));
#pragma unroll 1
    for (int aa = q; aa < 7; aa += 4) {
        const int y = i * 7 + aa - pad_t;
        if (y < 0 || y >= Hp) continue;       // (uniform over the wave)
        float4 o[7];
#pragma unroll
        for (int bb = 0; bb < 7; ++bb) {
            const int xx = j * 7 + bb - pad_l;
            const int xc = (xx >= 0 && xx < Wp) ? xx : 0;
            o[bb] = load4(dx + (((int64_t)b * Hp + y) * Wp + xc) * C + 4 * c4);
        }
#pragma unroll
        for (int bb = 0; bb < 7; ++bb) {
            const int xx = j * 7 + bb - pad_l;
            if (xx >= 0 && xx < Wp)
                store4(dx + (((int64_t)b * Hp + y) * Wp + xx) * C + 4 * c4, make_float4(o[bb].x + g.x, o[bb].y + g.y, o[bb].z + g.z, o[bb].w + g.w));
        }
    }
}

// The same product, left per window: g (windows, C) f32 = (dsamp . w) * leaky'(avg) / 49 -- the LayerNorm backward that consumes dx adds it to
// every token row of the window while it reads that row anyway (mtp_layernorm_bwd_win), instead of a read-modify-write pass over (T, C).
__global__ __launch_bounds__(256) void rvsa_sampling_bwd_win_kernel(const float* __restrict__ dsamp, const float* __restrict__ w, const float* __restrict__ avg,
                                                                   float* __restrict__ g, int C, int N) {
    extern __shared__ __attribute__((aligned(16))) float ds[];     // dsamp row of this window
    __shared__ float4 part[4][64];
    const int win = blockIdx.x;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c4 = blockIdx.y * 64 + lane;
    const bool live = c4 < C / 4;
    for (int n = threadIdx.x; n < N; n += 256) ds[n] = dsamp[(int64_t)win * N + n];
    __syncthreads();
    const int nq = (N + 3) / 4, n0 = q * nq, n1 = (n0 + nq) < N ? (n0 + nq) : N;
    float4 d = make_float4(0, 0, 0, 0);
    if (live) {
#pragma unroll 10
        for (int n = n0; n < n1; ++n) {
            const float4 ww = load4(w + (int64_t)n * C + 4 * c4);
            const float t = ds[n];
            d.x += t * ww.x; d.y += t * ww.y; d.z += t * ww.z; d.w += t * ww.w;
        }
    }
    part[q][lane] = d;
    __syncthreads();
    if (!live || q) return;
    const float4 p0 = part[0][lane], p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];   // the order of rvsa_sampling_bwd_kernel: same bits
    d = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
    const float4 a = load4(avg + (int64_t)win * C + 4 * c4);
    const float k = 1.0f / 49.0f;
    *reinterpret_cast<float4*>(g + (int64_t)win * C + 4 * c4) =
        make_float4(d.x * (a.x > 0 ? k : 0.01f * k), d.y * (a.y > 0 ? k : 0.01f * k), d.z * (a.z > 0 ? k : 0.01f * k), d.w * (a.w > 0 ? k : 0.01f * k));
}

extern "C" int mtp_rvsa_sampling_bwd_win(const float* dsamp, const float* w, const float* avg, float* g, int64_t windows, int64_t C, int64_t N, mtp_stream_t stream) {
    if (!dsamp || !w || !avg || !g || windows <= 0 || C <= 0 || (C % 4) || N <= 0 || N > 4096) return MTP_ERR_ARG;
    const dim3 grid((unsigned)windows, (unsigned)((C / 4 + 63) / 64)), block(256);
    hipLaunchKernelGGL(rvsa_sampling_bwd_win_kernel, grid, block, sizeof(float) * (size_t)N, (hipStream_t)stream, dsamp, w, avg, g, (int)C, (int)N);
    return mtp_launch_status();
}

extern "C" int mtp_rvsa_sampling_fwd(const void* x, int dtype, const float* w, const float* bias, float* avg, float* pooled, float* samp,
                                     int64_t B, int64_t Hp, int64_t Wp, int64_t C, int64_t N, mtp_stream_t stream) {
    if (!x || !w || !avg || !pooled || !samp || B <= 0 || Hp <= 0 || Wp <= 0 || C <= 0 || (C % 4) || C > 8192 || N <= 0) return MTP_ERR_ARG;
    int pt, pl, nh, nw;
    rvsa_geom(Hp, Wp, pt, pl, nh, nw);
    // workgroups per window (each pools the window again -- L2 reads -- and makes its share of the N outputs): round 2 measured 24.1 / 18.1 / 19.3 us at 1 / 2 / 4;
    // round 6, with the weight loads of a pass in flight together: 16.8 / 16.3 / 21.1 / 26.2 us at 1 / 2 / 3 / 5 (17.5 before) -- the repeated pooling, not the
    // product, is what more workgroups per window cost
    constexpr int ysplit = 2;
    const dim3 grid((unsigned)(B * nh * nw), (unsigned)(N >= 16 * ysplit ? ysplit : 1)), block(256);
    const size_t lds = sizeof(float) * (size_t)C;
    if (dtype == MTP_BF16)
        hipLaunchKernelGGL((rvsa_sampling_fwd_kernel<bf16_t>), grid, block, lds, (hipStream_t)stream, (const bf16_t*)x, w, bias, avg, pooled, samp, (int)Hp, (int)Wp, (int)C, (int)N, pt, pl, nh, nw);
    else if (dtype == MTP_F32)
        hipLaunchKernelGGL((rvsa_sampling_fwd_kernel<float>), grid, block, lds, (hipStream_t)stream, (const float*)x, w, bias, avg, pooled, samp, (int)Hp, (int)Wp, (int)C, (int)N, pt, pl, nh, nw);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}
/* dx (T, C) ACT += (dsamp (R, N) . w (N, C)) * leaky'(avg) / 49, broadcast over each window's tokens */
extern "C" int mtp_rvsa_sampling_bwd(const float* dsamp, const float* w, const float* avg, void* dx, int dtype,
                                     int64_t B, int64_t Hp, int64_t Wp, int64_t C, int64_t N, mtp_stream_t stream) {
    if (!dsamp || !w || !avg || !dx || B <= 0 || Hp <= 0 || Wp <= 0 || C <= 0 || (C % 4) || N <= 0 || N > 4096) return MTP_ERR_ARG;
    int pt, pl, nh, nw;
    rvsa_geom(Hp, Wp, pt, pl, nh, nw);
    const dim3 grid((unsigned)(B * nh * nw), (unsigned)((C / 4 + 63) / 64)), block(256);
    const size_t lds = sizeof(float) * (size_t)N;
    if (dtype == MTP_BF16)
        hipLaunchKernelGGL((rvsa_sampling_bwd_kernel<bf16_t>), grid, block, lds, (hipStream_t)stream, dsamp, w, avg, (bf16_t*)dx, (int)Hp, (int)Wp, (int)C, (int)N, pt, pl, nh, nw);
    else if (dtype == MTP_F32)
        hipLaunchKernelGGL((rvsa_sampling_bwd_kernel<float>), grid, block, lds, (hipStream_t)stream, dsamp, w, avg, (float*)dx, (int)Hp, (int)Wp, (int)C, (int)N, pt, pl, nh, nw);
    else return MTP_ERR_UNSUPPORTED;
    return mtp_launch_status();
}

extern "C" int mtp_rvsa_pool_fwd(const void* x, int dtype, float* avg, float* pooled, int64_t B, int64_t Hp, int64_t Wp, int64_t C, mtp_stream_t stream) {
    if (!x || !avg || !pooled || B <= 0 || (C % 4)) return MTP_ERR_ARG;
    int pt, pl, nh, nw;
    rvsa_geom(Hp, Wp, pt, pl, nh, nw);
    dim3 grid((unsigned)(B * nh * nw), (unsigned)((C / 4 + 255) / 256)), block(256);
    if (dtype == MTP_BF16)
        hipLaunchKernelGGL((rvsa_pool_fwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)x, avg, pooled, (int)Hp, (int)Wp, (int)C, pt, pl, nh, nw);
    else
        hipLaunchKernelGGL((rvsa_pool_fwd_kernel<float>), grid, block, 0, (hipStream_t)stream, (const float*)x, avg, pooled, (int)Hp, (int)Wp, (int)C, pt, pl, nh, nw);
    return mtp_launch_status();
}
extern "C" int mtp_rvsa_pool_bwd(const float* dpooled, const float* avg, void* dx, int dtype, int accumulate, int64_t B, int64_t Hp, int64_t Wp, int64_t C, mtp_stream_t stream) {
    if (!dpooled || !avg || !dx || B <= 0 || (C % 4)) return MTP_ERR_ARG;
    int pt, pl, nh, nw;
    rvsa_geom(Hp, Wp, pt, pl, nh, nw);
    dim3 grid(blocks_for(B * Hp * Wp * C / 4, 256, 8192)), block(256);
    if (dtype == MTP_BF16)
        hipLaunchKernelGGL((rvsa_pool_bwd_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, dpooled, avg, (bf16_t*)dx, accumulate, (int)B, (int)Hp, (int)Wp, (int)C, pt, pl, nh, nw);
    else
        hipLaunchKernelGGL((rvsa_pool_bwd_kernel<float>), grid, block, 0, (hipStream_t)stream, dpooled, avg, (float*)dx, accumulate, (int)B, (int)Hp, (int)Wp, (int)C, pt, pl, nh, nw);
    return mtp_launch_status();
}

extern "C" int mtp_small_linear_fwd(const float* x, const float* w, const float* b, float* y, int64_t R, int64_t N, int64_t K, mtp_stream_t stream) {
    if (!x || !w || !y || R <= 0 || N <= 0 || (K % 4)) return MTP_ERR_ARG;
    if (K <= 1024)
        hipLaunchKernelGGL((small_linear_fwd_kernel<4, 4>), dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, (int)R, (int)N, (int)K);
    else if (K <= 2048)
        hipLaunchKernelGGL((small_linear_fwd_kernel<2, 8>), dim3((unsigned)((R + 1) / 2)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, (int)R, (int)N, (int)K);
    else
        hipLaunchKernelGGL(small_linear_fwd_generic_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, x, w, b, y, (int)N, (int)K);
    return mtp_launch_status();
}
extern "C" int mtp_small_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int64_t R, int64_t N, int64_t K, mtp_stream_t stream) {
    if (!x || !w || !dy || R <= 0 || N <= 0 || (K % 4)) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dx) hipLaunchKernelGGL(small_linear_dx_kernel, dim3((unsigned)((K + 1023) / 1024), (unsigned)((R + 3) / 4)), dim3(256), 0, s, dy, w, dx, (int)R, (int)N, (int)K);
    if (dw) {
        if (db == dw + N * K) {   // one buffer [dw | db]: one clearing pass
            (void)hipMemsetAsync(dw, 0, sizeof(float) * (size_t)(N * K + N), s);
        } else {
            (void)hipMemsetAsync(dw, 0, sizeof(float) * (size_t)(N * K), s);
            if (db) (void)hipMemsetAsync(db, 0, sizeof(float) * (size_t)N, s);
        }
        hipLaunchKernelGGL(small_linear_dw_kernel<false>, dim3((unsigned)((K + 255) / 256), (unsigned)((N + 7) / 8), (unsigned)((R + 4 * SL_DW_ROWS - 1) / (4 * SL_DW_ROWS))), dim3(256), 0, s,
                           dy, x, dw, db, (int)R, (int)N, (int)K, SlSegs{});
    }
    return mtp_launch_status();
}
/* the weight / bias gradients of nseg <= 4 layers stacked along N, ACCUMULATED into their own (rows_j, K) / (rows_j) f32 buffers:
 * dw[j] += dy[:, r0_j : r0_j + rows_j]^T x.  Host arrays; db[j] may be NULL. */
extern "C" int mtp_small_linear_dw_segments(const float* x, const float* dy, int64_t R, int64_t N, int64_t K, int nseg, const int64_t* seg_rows,
                                            float* const* dw, float* const* db, mtp_stream_t stream) {
    if (!x || !dy || !seg_rows || !dw || R <= 0 || N <= 0 || K <= 0 || (K % 4) || nseg < 1 || nseg > 4) return MTP_ERR_ARG;
    SlSegs seg{};
    int64_t r = 0;
    for (int j = 0; j < nseg; ++j) {
        if (!dw[j] || seg_rows[j] <= 0) return MTP_ERR_ARG;
        seg.dw[j] = dw[j];
        seg.db[j] = db ? db[j] : nullptr;
        seg.row0[j] = (int)r;
        r += seg_rows[j];
    }
    if (r != N) return MTP_ERR_ARG;
    seg.row0[nseg] = (int)N;
    seg.nseg = nseg;
    hipLaunchKernelGGL(small_linear_dw_kernel<true>, dim3((unsigned)((K + 255) / 256), (unsigned)((N + 7) / 8), (unsigned)((R + 4 * SL_DW_ROWS - 1) / (4 * SL_DW_ROWS))), dim3(256), 0,
                       (hipStream_t)stream, dy, x, (float*)nullptr, (float*)nullptr, (int)R, (int)N, (int)K, seg);
    return mtp_launch_status();
}

/* the same for `count` <= 8 problems of one shape in one launch: xs / dys host arrays of device pointers, dw / db host arrays of
 * count * nseg device pointers (problem-major) */
extern "C" int mtp_small_linear_dw_segments_batched(const float* const* xs, const float* const* dys, int count, int64_t R, int64_t N, int64_t K, int nseg,
                                                    const int64_t* seg_rows, float* const* dw, float* const* db, mtp_stream_t stream) {
    if (!xs || !dys || !seg_rows || !dw || count <= 0 || count > SL_BATCH || R <= 0 || N <= 0 || K <= 0 || (K % 4) || nseg < 1 || nseg > 4) return MTP_ERR_ARG;
    SlBatch t{};
    for (int i = 0; i < count; ++i) {
        if (!xs[i] || !dys[i]) return MTP_ERR_ARG;
        t.x[i] = xs[i];
        t.dy[i] = dys[i];
        int64_t r = 0;
        for (int j = 0; j < nseg; ++j) {
            if (!dw[i * nseg + j] || seg_rows[j] <= 0) return MTP_ERR_ARG;
            t.seg[i].dw[j] = dw[i * nseg + j];
            t.seg[i].db[j] = db ? db[i * nseg + j] : nullptr;
            t.seg[i].row0[j] = (int)r;
            r += seg_rows[j];
        }
        if (r != N) return MTP_ERR_ARG;
        t.seg[i].row0[nseg] = (int)N;
        t.seg[i].nseg = nseg;
    }
    t.zsplit = (int)((R + 4 * SL_DW_ROWS - 1) / (4 * SL_DW_ROWS));
    hipLaunchKernelGGL(small_linear_dw_batched_kernel, dim3((unsigned)((K + 255) / 256), (unsigned)((N + 7) / 8), (unsigned)(t.zsplit * count)), dim3(256), 0,
                       (hipStream_t)stream, t, (int)R, (int)N, (int)K);
    return mtp_launch_status();
}

// base[start[i] .. start[i] + count[i]) = 0 for n segments (device tables; the host splits long runs so that one workgroup clears at
// most 64 K floats): the gradients that ACCUMULATE (biases, LayerNorm, rel-pos tables, sampling heads, FPN) inside the flat gradient
// buffer, without touching the 99 % of it that the weight-gradient GEMMs overwrite
__global__ __launch_bounds__(256) void zero_segments_kernel(float* __restrict__ base, const int64_t* __restrict__ start, const int64_t* __restrict__ count, int n) {
    for (int sgm = blockIdx.x; sgm < n; sgm += gridDim.x) {
        float* p = base + start[sgm];
        const int64_t c = count[sgm];
        for (int64_t i = threadIdx.x; i < c; i += 256) p[i] = 0.f;
    }
}
extern "C" int mtp_zero_segments_f32(float* base, const int64_t* start, const int64_t* count, int n, mtp_stream_t stream) {
    if (!base || !start || !count || n <= 0) return MTP_ERR_ARG;
    hipLaunchKernelGGL(zero_segments_kernel, dim3((unsigned)(n < 4096 ? n : 4096)), dim3(256), 0, (hipStream_t)stream, base, start, count, n);
    return mtp_launch_status();
}

// out += sum of squares over n runs base[start[i] .. start[i] + count[i]) (runs of at most 64 K floats, as mtp_zero_segments_f32): the part of the gradient norm that
// is not a by-product of the weight-gradient launches (biases, LayerNorm, tables, sampling heads, split problems) -- ~3 % of the buffer
__global__ __launch_bounds__(256) void sqnorm_segments_kernel(const float* __restrict__ base, const int64_t* __restrict__ start, const int64_t* __restrict__ count, int n,
                                                              float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int sgm = blockIdx.x; sgm < n; sgm += gridDim.x) {
        const float* p = base + start[sgm];
        const int64_t c = count[sgm];
        if (((start[sgm] | c) & 3) == 0) {      // 16-byte loads (the flat buffers pad every parameter to 64 elements: always, for their tables)
            for (int64_t i = 4 * threadIdx.x; i < c; i += 1024) {
                const float4 v = load4(p + i);
                s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        } else {
            for (int64_t i = threadIdx.x; i < c; i += 256) {
                const float v = p[i];
                s += v * v;
            }
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
extern "C" int mtp_sqnorm_segments_f32(const float* base, const int64_t* start, const int64_t* count, int n, float* out, mtp_stream_t stream) {
    if (!base || !start || !count || n <= 0 || !out) return MTP_ERR_ARG;
    hipLaunchKernelGGL(sqnorm_segments_kernel, dim3((unsigned)(n < 4096 ? n : 4096)), dim3(256), 0, (hipStream_t)stream, base, start, count, n, out);
    return mtp_launch_status();
}

extern "C" int mtp_sqnorm_f32(const float* g, float* out, int64_t n, mtp_stream_t stream) {
    if (!g || !out || n <= 0) return MTP_ERR_ARG;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks_for(n / 4 + 1, 256, 2048)), dim3(256), 0, (hipStream_t)stream, g, out, n);
    return mtp_launch_status();
}

extern "C" int mtp_adamw_flat(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_start, const float* seg_wd, int nseg,
                              const float* hyper, const float* sqnorm, float max_norm, float grad_scale, mtp_stream_t stream) {
    if (!p || !g || !m || !v || n <= 0 || (n % 4) || !seg_start || !seg_wd || nseg <= 0 || !hyper) return MTP_ERR_ARG;
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks_for(n / 4, 256, 8192)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, seg_start, seg_wd, nseg, hyper, sqnorm, max_norm, grad_scale);
    return mtp_launch_status();
}

namespace {
// dst[r][c] = scale[r / rows_per_sample] * src[r][c]  (f32 -> ACT), the drop-path-scaled operand copy of a residual gradient
template <typename T>
__global__ __launch_bounds__(256) void scale_rows_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, const float* __restrict__ scale,
                                                             int64_t rows_per_sample, int64_t rows, int64_t C) {
    const int64_t C4 = C / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * C4; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C4;
        const float s = scale ? scale[r / rows_per_sample] : 1.0f;
        const float4 v = load4(src + 4 * i);
        store4(dst + 4 * i, make_float4(v.x * s, v.y * s, v.z * s, v.w * s));
    }
}
}  // namespace

extern "C" int mtp_scale_rows_cast(const float* src, void* dst, int dst_dtype, const float* scale, int64_t rows_per_sample, int64_t rows, int64_t C, mtp_stream_t stream) {
    if (!src || !dst || rows <= 0 || (C % 4) || (scale && rows_per_sample <= 0)) return MTP_ERR_ARG;
    dim3 grid(blocks_for(rows * C / 4, 256, 8192)), block(256);
    if (dst_dtype == MTP_BF16)
        hipLaunchKernelGGL((scale_rows_cast_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, src, (bf16_t*)dst, scale, rows_per_sample, rows, C);
    else
        hipLaunchKernelGGL((scale_rows_cast_kernel<float>), grid, block, 0, (hipStream_t)stream, src, (float*)dst, scale, rows_per_sample, rows, C);
    return mtp_launch_status();
}

// 0.4: round 4 -- mtp_gemm_args as of round 3 (workspace / workspace_bytes trailing fields; now ignored: the stream-K form is gone),
// mtp_gemm_tn_grouped honours split_k / aux.  Bump whenever a struct in include/mtp_hip.h changes size or a field changes meaning.
// 0.5: round 5 -- no struct changed; mtp_gemm_args.variant gained bits 17 / 18 (strip kernel) and 19 (grouped TN: plain phases), mtp_gemm_nt_tile may answer 64.
// 0.6: round 6 -- mtp_wimg_desc.pad_ became `float wd` (same size; read only by mtp_adamw_weight_images); new entry points mtp_adamw_weight_images, mtp_stream_create_cu_mask,
// mtp_probe_placement, mtp_comm_info.
extern "C" const char* mtp_version(void) { return "mtp_hip 0.6 (gfx950)"; }

// A stream of the LOWEST priority the device offers (non-blocking), for work that is off the critical path and should only take the CUs
// the main stream leaves idle: the grouped weight-gradient launches next to under-filled data-gradient GEMMs (engine_intern.py).
// The caller owns the handle (mtp_stream_destroy); it can be wrapped as a torch.cuda.ExternalStream.
extern "C" int mtp_stream_create_low_priority(void** stream) {
    if (!stream) return MTP_ERR_ARG;
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return (int)e;
    hipStream_t s = nullptr;
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least);
    if (e != hipSuccess) return (int)e;
    *stream = (void*)s;
    return 0;
}
// ---- per-stream CU budget: written when a masked stream is created / destroyed (under a mutex), read by every GEMM dispatch (lock-free scan of 16 slots)
namespace {
constexpr int kMaskSlots = 16;
std::atomic<void*> g_mask_stream[kMaskSlots];
std::atomic<int> g_mask_cus[kMaskSlots];
std::mutex g_mask_mu;
int device_cus() {
    static std::atomic<int> ncu{0};
    int n = ncu.load(std::memory_order_relaxed);
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        n = prop.multiProcessorCount;
        ncu.store(n, std::memory_order_relaxed);
    }
    return n;
}
}  // namespace

int mtp_stream_cus(hipStream_t stream) {
    if (stream)
        for (int i = 0; i < kMaskSlots; ++i)
            if (g_mask_stream[i].load(std::memory_order_acquire) == (void*)stream) return g_mask_cus[i].load(std::memory_order_relaxed);
    return device_cus();
}

// A stream whose kernels may only use the CUs named by a bit mask (hipExtStreamCreateWithCUMask): the two half-batch schedule (engine.py,
// DESIGN section 5b) gives each half its own CUs, so that one half's HBM-bound kernels and epilogue bursts run beside the other half's K loops
// instead of queueing behind them.  gfx950 in SPX mode: bit i of the mask = XCC (i % 8), CU (i / 8) of that XCC in the driver's
// enumeration (checked on the hardware by mtp_probe_placement, profiles/r06_cu_mask_probe.txt).  A mask that leaves an XCC without any CU is rejected:
// the dispatcher still hands that XCC every eighth workgroup.
extern "C" int mtp_stream_create_cu_mask(const uint32_t* mask, int words, void** stream) {
    if (!mask || !stream || words <= 0 || words > 32) return MTP_ERR_ARG;
    for (int x = 0; x < 8; ++x) {
        bool any = false;
        for (int i = x; i < words * 32 && i < 256; i += 8) any |= (mask[i / 32] >> (i % 32)) & 1u;
        if (!any) return MTP_ERR_ARG;
    }
    int cus = 0;
    for (int i = 0; i < words * 32 && i < 256; ++i) cus += (mask[i / 32] >> (i % 32)) & 1u;
    std::lock_guard<std::mutex> lock(g_mask_mu);
    int slot = -1;
    for (int i = 0; i < kMaskSlots && slot < 0; ++i)
        if (!g_mask_stream[i].load(std::memory_order_relaxed)) slot = i;
    if (slot < 0) return MTP_ERR_UNSUPPORTED;      // more masked streams alive than the table holds
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) return (int)e;
    g_mask_cus[slot].store(cus, std::memory_order_relaxed);
    g_mask_stream[slot].store((void*)s, std::memory_order_release);
    *stream = (void*)s;
    return 0;
}

namespace {
// one record per workgroup: {XCC_ID, HW_ID}; every workgroup stays resident for `spin` clocks so that a launch of >= 2 workgroups per CU touches every
// CU the stream may use
__global__ void __launch_bounds__(256) probe_placement_kernel(int32_t* __restrict__ out, long long spin) {
    const long long t0 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x + 0] = (int32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        out[2 * blockIdx.x + 1] = (int32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    }
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

// Where do the workgroups of a launch on `stream` run?  out: (blocks, 2) int32 = {XCC id, HW_ID register} per workgroup.
extern "C" int mtp_probe_placement(int32_t* out, int blocks, int64_t spin_clocks, mtp_stream_t stream) {
    if (!out || blocks <= 0 || spin_clocks < 0 || spin_clocks > (1ll << 28)) return MTP_ERR_ARG;
    hipLaunchKernelGGL(probe_placement_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, (long long)spin_clocks);
    return mtp_launch_status();
}

extern "C" int mtp_stream_destroy(void* stream) {
    if (!stream) return MTP_ERR_ARG;
    {
        std::lock_guard<std::mutex> lock(g_mask_mu);
        for (int i = 0; i < kMaskSlots; ++i)
            if (g_mask_stream[i].load(std::memory_order_relaxed) == stream) g_mask_stream[i].store(nullptr, std::memory_order_release);
    }
    return (int)hipStreamDestroy((hipStream_t)stream);
}
