// LayerNorm forward/backward (nn.LayerNorm(C, eps=1e-6), VIT:484,496,579,596) for gfx950.
// HBM-bound: one 64-lane wavefront per row, 16-byte accesses, row kept in registers (C <= 1024*... see MAXV),
// statistics in f32 with a centred second pass, wave reductions via cross-lane shuffles (no LDS on the row path).
// Optional fused exact-erf GELU on the output (fpn1: Norm2d -> GELU, VIT:643-644).
// Backward fuses: residual-gradient add, an extra addend (FPN tap gradient), the ACT-dtype copy of the result
// (operand of the next dgrad/wgrad GEMMs, pre-multiplied by the drop-path factor), and per-block dgamma/dbeta partials.
#include "common.h"

namespace {

constexpr int LN_THREADS = 256;   // 4 rows per block pass
constexpr int MAXV = 8;           // float4 per lane kept in registers: kernels are instantiated for 1, 2, 3, 4, 6 and 8 (C <= 2048), see MTP_LN_MV

// All row loads are UNCONDITIONAL on a clamped column (lanes past the row end re-read the last group and are masked in the
// arithmetic / at the store): with `if (c4 < nv)` around them hipcc emitted an exec-masked block + s_waitcnt vmcnt(0) per load,
// i.e. one HBM latency after the other (the backward ran at 3.4 TB/s with 8-10 serialised waits per row).
template <typename Tx, typename Ty, bool GELU, int MV>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const Tx* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           Ty* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                           int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2;   // float4 groups per row
    int col[MV];
    bool ok[MV];
    float4 g[MV], bb[MV];
#pragma unroll
    for (int i = 0; i < MV; ++i) {
        const int c4 = lane + 64 * i;
        ok[i] = c4 < nv;
        col[i] = 4 * (ok[i] ? c4 : nv - 1);
        g[i] = *reinterpret_cast<const float4*>(gamma + col[i]);
        bb[i] = *reinterpret_cast<const float4*>(beta + col[i]);
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const Tx* xr = x + row * C;
        float4 v[MV];
#pragma unroll
        for (int i = 0; i < MV; ++i) v[i] = load4(xr + col[i]);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MV; ++i) s += ok[i] ? (v[i].x + v[i].y + v[i].z + v[i].w) : 0.f;
        const float mu = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
            q += ok[i] ? (a * a + b * b + c * c + d * d) : 0.f;
        }
        const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
        Ty* yr = y + row * C;
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            float4 o = make_float4((v[i].x - mu) * rs * g[i].x + bb[i].x, (v[i].y - mu) * rs * g[i].y + bb[i].y,
                                   (v[i].z - mu) * rs * g[i].z + bb[i].z, (v[i].w - mu) * rs * g[i].w + bb[i].w);
            if (GELU) o = make_float4(gelu_f(o.x), gelu_f(o.y), gelu_f(o.z), gelu_f(o.w));
            if (ok[i]) store4(yr + col[i], o);
        }
    }
}

// per-window addend of the incoming gradient (mtp_layernorm_bwd_win): token row r = (b, y, x) of a (B, Hp, Wp) grid belongs to the 7 x 7 window
// ((y + pad_t) / 7, (x + pad_l) / 7) of image b (VIT:298-310)
struct LnWin { int Hp, Wp, pad_t, pad_l, nh, nw; };

template <typename Tact, typename Tx, typename Tdx, bool GELU, int MV>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const Tact* __restrict__ dy, const Tx* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ dres, const float* __restrict__ extra, Tdx* __restrict__ dx,
                                                           Tact* __restrict__ dx_copy, const float* __restrict__ copy_scale, int rows_per_sample,
                                                           float* __restrict__ dgamma_part, float* __restrict__ dbeta_part, int64_t part_ld, int64_t rows, int C,
                                                           const float* __restrict__ win_add, LnWin wg) {
    __shared__ float4 red[2][3][64 * MV];   // waves 1..3 -> wave 0
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2;
    float4 gacc[MV], bacc[MV], g[MV], bt[MV];
    int col[MV];
    bool ok[MV];
#pragma unroll
    for (int i = 0; i < MV; ++i) {
        gacc[i] = make_float4(0, 0, 0, 0);
        bacc[i] = make_float4(0, 0, 0, 0);
        const int c4 = lane + 64 * i;
        ok[i] = c4 < nv;
        col[i] = 4 * (ok[i] ? c4 : nv - 1);
        g[i] = *reinterpret_cast<const float4*>(gamma + col[i]);
        bt[i] = GELU ? *reinterpret_cast<const float4*>(beta + col[i]) : make_float4(0, 0, 0, 0);
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        // ---- every load of the row first (the uniform `if (dres)` blocks contain loads only, no uses)
        float4 xv[MV], dv[MV], rr[MV], ee[MV];
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            xv[i] = load4(x + row * C + col[i]);
            dv[i] = load4(dy + row * C + col[i]);
            rr[i] = zero4;
            ee[i] = zero4;
        }
        if (dres) {
#pragma unroll
            for (int i = 0; i < MV; ++i) rr[i] = *reinterpret_cast<const float4*>(dres + row * C + col[i]);
        }
        if (extra) {
#pragma unroll
            for (int i = 0; i < MV; ++i) ee[i] = *reinterpret_cast<const float4*>(extra + row * C + col[i]);
        }
        float4 wa[MV];
#pragma unroll
        for (int i = 0; i < MV; ++i) wa[i] = zero4;
        if (win_add) {      // (uniform; the row -> window arithmetic is wave-uniform 32-bit integer work)
            const uint32_t r32 = (uint32_t)row, npi = (uint32_t)(wg.Hp * wg.Wp);
            const uint32_t b = r32 / npi, t = r32 - b * npi, y = t / (uint32_t)wg.Wp, xq = t - y * (uint32_t)wg.Wp;
            const uint32_t win = (b * (uint32_t)wg.nh + (y + (uint32_t)wg.pad_t) / 7u) * (uint32_t)wg.nw + (xq + (uint32_t)wg.pad_l) / 7u;
            const float* wrow = win_add + (int64_t)win * C;
#pragma unroll
            for (int i = 0; i < MV; ++i) wa[i] = *reinterpret_cast<const float4*>(wrow + col[i]);
        }
        const float mu = mean[row], rs = rstd[row];
        const float cs = (dx_copy && copy_scale) ? copy_scale[(uint32_t)row / (uint32_t)rows_per_sample] : 1.0f;   // (64-bit division is ~150 instructions)
        float4 xh[MV], d[MV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            dv[i].x += wa[i].x; dv[i].y += wa[i].y; dv[i].z += wa[i].z; dv[i].w += wa[i].w;
            if (!ok[i]) dv[i] = zero4;   // select on the loaded VALUE: masked lanes add nothing below
            xh[i] = make_float4((xv[i].x - mu) * rs, (xv[i].y - mu) * rs, (xv[i].z - mu) * rs, (xv[i].w - mu) * rs);
            if (GELU) {   // y = gelu(z), z = xhat*gamma+beta
                dv[i].x *= dgelu_f(xh[i].x * g[i].x + bt[i].x);
                dv[i].y *= dgelu_f(xh[i].y * g[i].y + bt[i].y);
                dv[i].z *= dgelu_f(xh[i].z * g[i].z + bt[i].z);
                dv[i].w *= dgelu_f(xh[i].w * g[i].w + bt[i].w);
            }
            gacc[i].x += dv[i].x * xh[i].x; gacc[i].y += dv[i].y * xh[i].y; gacc[i].z += dv[i].z * xh[i].z; gacc[i].w += dv[i].w * xh[i].w;
            bacc[i].x += dv[i].x; bacc[i].y += dv[i].y; bacc[i].z += dv[i].z; bacc[i].w += dv[i].w;
            d[i] = make_float4(dv[i].x * g[i].x, dv[i].y * g[i].y, dv[i].z * g[i].z, dv[i].w * g[i].w);
            s1 += d[i].x * xh[i].x + d[i].y * xh[i].y + d[i].z * xh[i].z + d[i].w * xh[i].w;
            s2 += d[i].x + d[i].y + d[i].z + d[i].w;
        }
        const float c1 = wave_sum(s1) / (float)C, c2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            const float4 o = make_float4((d[i].x - xh[i].x * c1 - c2) * rs + rr[i].x + ee[i].x, (d[i].y - xh[i].y * c1 - c2) * rs + rr[i].y + ee[i].y,
                                         (d[i].z - xh[i].z * c1 - c2) * rs + rr[i].z + ee[i].z, (d[i].w - xh[i].w * c1 - c2) * rs + rr[i].w + ee[i].w);
            if (ok[i]) {
                store4(dx + row * C + col[i], o);
                if (dx_copy) store4(dx_copy + row * C + col[i], make_float4(o.x * cs, o.y * cs, o.z * cs, o.w * cs));
            }
        }
    }
    // block-level reduction of the parameter-gradient partials (waves 1..3 -> LDS -> wave 0)
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            red[0][wave - 1][lane + 64 * i] = gacc[i];
            red[1][wave - 1][lane + 64 * i] = bacc[i];
        }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            if (ok[i]) {
                float4 a = gacc[i], b = bacc[i];
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const float4 ra = red[0][w][lane + 64 * i], rb = red[1][w][lane + 64 * i];
                    a.x += ra.x; a.y += ra.y; a.z += ra.z; a.w += ra.w;
                    b.x += rb.x; b.y += rb.y; b.z += rb.z; b.w += rb.w;
                }
                *reinterpret_cast<float4*>(dgamma_part + (int64_t)blockIdx.x * part_ld + col[i]) = a;
                *reinterpret_cast<float4*>(dbeta_part + (int64_t)blockIdx.x * part_ld + col[i]) = b;
            }
        }
    }
}

// ---- InternImage's post-norm residual (intern_image.py:424-426) in ONE pass each way (round 4; was LayerNorm + scale_residual, two launches
// and a bf16 round trip of the normalised rows each way):   out = x + s[sample] * ls * LayerNorm(h)
//   forward: reads h (ACT) and x (f32), writes out (f32), its ACT copy, mean / rstd.
//   backward: dout (f32) -> dh (ACT) = LN'(s * ls * dout); partials per workgroup [dgamma | dbeta | dls] with dls = sum_rows s * dout * z,
//   z = xhat * gamma + beta recomputed from the saved statistics (never stored).
template <typename Th, int MV>
__global__ __launch_bounds__(LN_THREADS) void ln_res_fwd_kernel(const Th* __restrict__ h, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ x, const float* __restrict__ ls, const float* __restrict__ sample_scale,
                                                               int rows_per_sample, float* __restrict__ out, Th* __restrict__ out_act,
                                                               float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2;
    int col[MV];
    bool ok[MV];
    float4 g[MV], bb[MV], lv[MV];
#pragma unroll
    for (int i = 0; i < MV; ++i) {
        const int c4 = lane + 64 * i;
        ok[i] = c4 < nv;
        col[i] = 4 * (ok[i] ? c4 : nv - 1);
        g[i] = *reinterpret_cast<const float4*>(gamma + col[i]);
        bb[i] = *reinterpret_cast<const float4*>(beta + col[i]);
        lv[i] = *reinterpret_cast<const float4*>(ls + col[i]);
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        float4 v[MV], xv[MV];
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            v[i] = load4(h + row * C + col[i]);
            xv[i] = *reinterpret_cast<const float4*>(x + row * C + col[i]);
        }
        const float sc = sample_scale ? sample_scale[(uint32_t)row / (uint32_t)rows_per_sample] : 1.0f;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MV; ++i) s += ok[i] ? (v[i].x + v[i].y + v[i].z + v[i].w) : 0.f;
        const float mu = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, d = v[i].w - mu;
            q += ok[i] ? (a * a + b * b + c * c + d * d) : 0.f;
        }
        const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            const float4 z = make_float4((v[i].x - mu) * rs * g[i].x + bb[i].x, (v[i].y - mu) * rs * g[i].y + bb[i].y,
                                         (v[i].z - mu) * rs * g[i].z + bb[i].z, (v[i].w - mu) * rs * g[i].w + bb[i].w);
            const float4 o = make_float4(xv[i].x + sc * lv[i].x * z.x, xv[i].y + sc * lv[i].y * z.y, xv[i].z + sc * lv[i].z * z.z, xv[i].w + sc * lv[i].w * z.w);
            if (ok[i]) {
                *reinterpret_cast<float4*>(out + row * C + col[i]) = o;
                if (out_act) store4(out_act + row * C + col[i], o);
            }
        }
    }
}

template <typename Th, int MV>
__global__ __launch_bounds__(LN_THREADS) void ln_res_bwd_kernel(const float* __restrict__ dout, const Th* __restrict__ h, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ ls, const float* __restrict__ sample_scale, int rows_per_sample,
                                                               Th* __restrict__ dh, float* __restrict__ part, int64_t rows, int C) {
    __shared__ float4 red[3][3][64 * MV];   // [dgamma | dbeta | dls] of waves 1..3 -> wave 0
    // MV = 8 (C up to 2048: InternImage-XL's 1536-channel level) makes this 72 KiB of STATIC LDS -- above the 64 KiB most targets allow, inside gfx950's
    // 160 KiB (the only target of this library), and two such workgroups still share a CU
    static_assert(sizeof(float4) * 3 * 3 * 64 * MV <= 80 * 1024, "ln_res_bwd_kernel: two workgroups per CU need <= 80 KiB of LDS each (gfx950: 160 KiB per CU)");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = C >> 2;
    float4 gacc[MV], bacc[MV], lacc[MV], g[MV], bt[MV], lv[MV];
    int col[MV];
    bool ok[MV];
#pragma unroll
    for (int i = 0; i < MV; ++i) {
        gacc[i] = make_float4(0, 0, 0, 0);
        bacc[i] = make_float4(0, 0, 0, 0);
        lacc[i] = make_float4(0, 0, 0, 0);
        const int c4 = lane + 64 * i;
        ok[i] = c4 < nv;
        col[i] = 4 * (ok[i] ? c4 : nv - 1);
        g[i] = *reinterpret_cast<const float4*>(gamma + col[i]);
        bt[i] = *reinterpret_cast<const float4*>(beta + col[i]);
        lv[i] = *reinterpret_cast<const float4*>(ls + col[i]);
    }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        float4 xv[MV], dv[MV];
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            xv[i] = load4(h + row * C + col[i]);
            dv[i] = *reinterpret_cast<const float4*>(dout + row * C + col[i]);
        }
        const float mu = mean[row], rs = rstd[row];
        const float sc = sample_scale ? sample_scale[(uint32_t)row / (uint32_t)rows_per_sample] : 1.0f;
        float4 xh[MV], d[MV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            if (!ok[i]) dv[i] = zero4;
            xh[i] = make_float4((xv[i].x - mu) * rs, (xv[i].y - mu) * rs, (xv[i].z - mu) * rs, (xv[i].w - mu) * rs);
            const float4 sd = make_float4(sc * dv[i].x, sc * dv[i].y, sc * dv[i].z, sc * dv[i].w);
            lacc[i].x += sd.x * (xh[i].x * g[i].x + bt[i].x); lacc[i].y += sd.y * (xh[i].y * g[i].y + bt[i].y);
            lacc[i].z += sd.z * (xh[i].z * g[i].z + bt[i].z); lacc[i].w += sd.w * (xh[i].w * g[i].w + bt[i].w);
            const float4 dy = make_float4(sd.x * lv[i].x, sd.y * lv[i].y, sd.z * lv[i].z, sd.w * lv[i].w);     // gradient of the LayerNorm output
            gacc[i].x += dy.x * xh[i].x; gacc[i].y += dy.y * xh[i].y; gacc[i].z += dy.z * xh[i].z; gacc[i].w += dy.w * xh[i].w;
            bacc[i].x += dy.x; bacc[i].y += dy.y; bacc[i].z += dy.z; bacc[i].w += dy.w;
            d[i] = make_float4(dy.x * g[i].x, dy.y * g[i].y, dy.z * g[i].z, dy.w * g[i].w);
            s1 += d[i].x * xh[i].x + d[i].y * xh[i].y + d[i].z * xh[i].z + d[i].w * xh[i].w;
            s2 += d[i].x + d[i].y + d[i].z + d[i].w;
        }
        const float c1 = wave_sum(s1) / (float)C, c2 = wave_sum(s2) / (float)C;
#pragma unroll
        for (int i = 0; i < MV; ++i)
            if (ok[i])
                store4(dh + row * C + col[i], make_float4((d[i].x - xh[i].x * c1 - c2) * rs, (d[i].y - xh[i].y * c1 - c2) * rs,
                                                          (d[i].z - xh[i].z * c1 - c2) * rs, (d[i].w - xh[i].w * c1 - c2) * rs));
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            red[0][wave - 1][lane + 64 * i] = gacc[i];
            red[1][wave - 1][lane + 64 * i] = bacc[i];
            red[2][wave - 1][lane + 64 * i] = lacc[i];
        }
    }
    __syncthreads();
    if (wave == 0) {
        float* prow = part + (int64_t)blockIdx.x * 3 * C;
#pragma unroll
        for (int i = 0; i < MV; ++i) {
            if (ok[i]) {
                float4 a = gacc[i], b = bacc[i], l = lacc[i];
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const float4 ra = red[0][w][lane + 64 * i], rb = red[1][w][lane + 64 * i], rl = red[2][w][lane + 64 * i];
                    a.x += ra.x; a.y += ra.y; a.z += ra.z; a.w += ra.w;
                    b.x += rb.x; b.y += rb.y; b.z += rb.z; b.w += rb.w;
                    l.x += rl.x; l.y += rl.y; l.z += rl.z; l.w += rl.w;
                }
                *reinterpret_cast<float4*>(prow + col[i]) = a;
                *reinterpret_cast<float4*>(prow + C + col[i]) = b;
                *reinterpret_cast<float4*>(prow + 2 * C + col[i]) = l;
            }
        }
    }
}

// sum of rows r0..r1 of one column: four independent loads in flight per thread (one 4-byte load at a time ran at ~1.8 TB/s out of L2)
__device__ __forceinline__ float column_sum(const float* __restrict__ col, int64_t ld, int64_t r0, int64_t r1) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int64_t r = r0;
    for (; r + 8 <= r1; r += 8) {     // (round 4: eight in flight -- the 16384-row partial buffers of the RVSA backward are ~113 rows per workgroup)
        const float a = col[r * ld], b = col[(r + 1) * ld], c = col[(r + 2) * ld], d = col[(r + 3) * ld];
        const float e = col[(r + 4) * ld], f = col[(r + 5) * ld], g = col[(r + 6) * ld], h = col[(r + 7) * ld];
        s0 += a; s1 += b; s2 += c; s3 += d;
        s0 += e; s1 += f; s2 += g; s3 += h;
    }
    for (; r + 4 <= r1; r += 4) {
        const float a = col[r * ld], b = col[(r + 1) * ld], c = col[(r + 2) * ld], d = col[(r + 3) * ld];
        s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; r < r1; ++r) s0 += col[r * ld];
    return (s0 + s1) + (s2 + s3);
}

// out[c] (+)= sum_r part[r][c]; thread per column, rows split over blockIdx.y, f32 atomics into (zeroed) out
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, int64_t ld, float* __restrict__ out, int64_t rows, int64_t C, int64_t rows_per_block) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    r1 = r1 < rows ? r1 : rows;
    atomicAdd(out + c, column_sum(part + c, ld, r0, r1));
}

// n equally-shaped partial buffers in one launch (blockIdx.z = buffer): the LayerNorm parameter gradients of a burst of blocks
struct RrBatch {
    const float* part[MTP_REDUCE_BATCH_MAX];
    float* out[MTP_REDUCE_BATCH_MAX];
};
__global__ __launch_bounds__(256) void reduce_rows_batched_kernel(RrBatch t, int64_t ld, int64_t rows, int64_t C, int64_t rows_per_block) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* __restrict__ part = t.part[blockIdx.z];
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    r1 = r1 < rows ? r1 : rows;
    atomicAdd(t.out[blockIdx.z] + c, column_sum(part + c, ld, r0, r1));
}

// the same with a transposed result: column j = a * C + b of part (rows, R * C) goes to out[b * R + a] (per-(window, head) partials of
// the (169, heads) bias-table gradient are written head-major by the RVSA backward: contiguous per workgroup)
__global__ __launch_bounds__(256) void reduce_rows_t_kernel(const float* __restrict__ part, int64_t ld, float* __restrict__ out, int64_t rows, int R, int C, int64_t rows_per_block) {
    // thread = OUTPUT element q = b * R + a (a fastest): the atomics of a wave are 64-byte runs (with thread = input column they were 64
    // separate segments per instruction: 35 us per call); the reads are strided by C floats but come out of L2 / L1
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= (int64_t)R * C) return;
    const int a = (int)(q % R), b = (int)(q / R);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    r1 = r1 < rows ? r1 : rows;
    const float* src = part + (int64_t)a * C + b;
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += src[r * ld];
    atomicAdd(out + q, s);
}

// ... and n of those in one launch (blockIdx.z = buffer): the bias-table gradients of a burst of RVSA blocks
__global__ __launch_bounds__(256) void reduce_rows_t_batched_kernel(RrBatch t, int64_t ld, int64_t rows, int R, int C, int64_t rows_per_block) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= (int64_t)R * C) return;
    const int a = (int)(q % R), b = (int)(q / R);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    r1 = r1 < rows ? r1 : rows;
    const float* src = t.part[blockIdx.z] + (int64_t)a * C + b;
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += src[r * ld];
    atomicAdd(t.out[blockIdx.z] + q, s);
}

// bias gradient: column sums of dY (M, N).  Block = 64 columns-of-4 x 4 row-lanes; grid.y splits the rows; f32 atomics.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dY, int64_t ld, float* __restrict__ out, int64_t M, int64_t N, int64_t rows_per_block) {
    __shared__ float4 red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int64_t n = ((int64_t)blockIdx.x * 64 + cx) * 4;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    r1 = r1 < M ? r1 : M;
    float4 s = make_float4(0, 0, 0, 0);
    if (n < N) {
        int64_t r = r0 + ry;
        for (; r + 12 < r1; r += 16) {   // 4 independent row loads in flight per thread
            const float4 v0 = load4(dY + r * ld + n), v1 = load4(dY + (r + 4) * ld + n), v2 = load4(dY + (r + 8) * ld + n), v3 = load4(dY + (r + 12) * ld + n);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; r < r1; r += 4) {
            const float4 v = load4(dY + r * ld + n);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[ry][cx] = s;
    __syncthreads();
    if (ry == 0 && n < N) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 v = red[w][cx];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        atomicAdd(out + n, s.x); atomicAdd(out + n + 1, s.y); atomicAdd(out + n + 2, s.z); atomicAdd(out + n + 3, s.w);
    }
}

int ln_grid(int64_t rows) {
    int64_t nb = (rows + 3) / 4;
    return (int)(nb < 2048 ? nb : 2048);
}

// float4 groups per lane the row kernels keep in registers: instantiated for 1, 2, 3, 4, 6, 8 (C <= 256 ... 2048).  Every load, store and arithmetic step of a
// row runs MV times per lane, masked or not (the loads are unconditional on clamped columns), so a 192-channel row on the MV = 4 kernel did four times its work --
// round 6: InternImage's 96- / 192-channel stem and first level ran at a quarter of their bandwidth (the stem's LayerNorm backward: 617 us for 300 MB).
#define MTP_LN_MV(C_, CALL)                                      \
    do {                                                         \
        const int mv_ = (int)(((C_) / 4 + 63) / 64);             \
        if (mv_ <= 1) { CALL(1); }                               \
        else if (mv_ == 2) { CALL(2); }                          \
        else if (mv_ == 3) { CALL(3); }                          \
        else if (mv_ == 4) { CALL(4); }                          \
        else if (mv_ <= 6) { CALL(6); }                          \
        else { CALL(8); }                                        \
    } while (0)

template <typename Tx, typename Ty>
int launch_ln_fwd(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, int64_t rows, int64_t C, float eps, int gelu, hipStream_t s) {
    dim3 grid(ln_grid(rows)), block(LN_THREADS);
#define MTP_CALL(MV_)                                                                                                                                        \
    if (gelu)                                                                                                                                                \
        hipLaunchKernelGGL((ln_fwd_kernel<Tx, Ty, true, MV_>), grid, block, 0, s, (const Tx*)x, g, b, (Ty*)y, mean, rstd, rows, (int)C, eps);                \
    else                                                                                                                                                     \
        hipLaunchKernelGGL((ln_fwd_kernel<Tx, Ty, false, MV_>), grid, block, 0, s, (const Tx*)x, g, b, (Ty*)y, mean, rstd, rows, (int)C, eps)
    MTP_LN_MV(C, MTP_CALL);
#undef MTP_CALL
    return mtp_launch_status();
}

template <typename Tact, typename Tx, typename Tdx>
int launch_ln_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int gelu,
                  const float* dres, const float* extra, void* dx, void* dx_copy, const float* copy_scale, int64_t rps,
                  float* dgp, float* dbp, int64_t part_ld, int64_t rows, int64_t C, hipStream_t s, const float* win_add = nullptr, LnWin wg = LnWin{1, 1, 0, 0, 1, 1}) {
    dim3 grid((unsigned)mtp_layernorm_bwd_partial_rows(rows)), block(LN_THREADS);
#define MTP_CALL(MV_)                                                                                                                                        \
    if (gelu)                                                                                                                                                \
        hipLaunchKernelGGL((ln_bwd_kernel<Tact, Tx, Tdx, true, MV_>), grid, block, 0, s, (const Tact*)dy, (const Tx*)x, mean, rstd, gamma, beta, dres, extra, \
                           (Tdx*)dx, (Tact*)dx_copy, copy_scale, (int)(rps > 0 ? rps : 1), dgp, dbp, part_ld, rows, (int)C, win_add, wg);                    \
    else                                                                                                                                                     \
        hipLaunchKernelGGL((ln_bwd_kernel<Tact, Tx, Tdx, false, MV_>), grid, block, 0, s, (const Tact*)dy, (const Tx*)x, mean, rstd, gamma, beta, dres, extra, \
                           (Tdx*)dx, (Tact*)dx_copy, copy_scale, (int)(rps > 0 ? rps : 1), dgp, dbp, part_ld, rows, (int)C, win_add, wg)
    MTP_LN_MV(C, MTP_CALL);
#undef MTP_CALL
    return mtp_launch_status();
}

}  // namespace

extern "C" int mtp_layernorm_fwd(const void* x, int x_dtype, const float* gamma, const float* beta, void* y, int y_dtype,
                                 float* mean, float* rstd, int64_t rows, int64_t C, float eps, int fuse_gelu, mtp_stream_t stream) {
    if (!x || !gamma || !beta || !y || rows <= 0 || C <= 0 || (C % 4) || C > 256 * MAXV) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MTP_F32 && y_dtype == MTP_F32) return launch_ln_fwd<float, float>(x, gamma, beta, y, mean, rstd, rows, C, eps, fuse_gelu, s);
    if (x_dtype == MTP_F32 && y_dtype == MTP_BF16) return launch_ln_fwd<float, bf16_t>(x, gamma, beta, y, mean, rstd, rows, C, eps, fuse_gelu, s);
    if (x_dtype == MTP_BF16 && y_dtype == MTP_BF16) return launch_ln_fwd<bf16_t, bf16_t>(x, gamma, beta, y, mean, rstd, rows, C, eps, fuse_gelu, s);
    if (x_dtype == MTP_BF16 && y_dtype == MTP_F32) return launch_ln_fwd<bf16_t, float>(x, gamma, beta, y, mean, rstd, rows, C, eps, fuse_gelu, s);
    return MTP_ERR_UNSUPPORTED;
}

template <typename Th>
static int launch_ln_res(bool fwd, const void* a0, const void* h, const float* mean_c, float* mean, const float* rstd_c, float* rstd, const float* gamma, const float* beta,
                         const float* x, const float* ls, const float* ss, int rps, float* out, void* oact, void* dh, float* part, int64_t rows, int64_t C, float eps,
                         hipStream_t s) {
    if (fwd) {
        const dim3 grid(ln_grid(rows)), block(LN_THREADS);
#define MTP_CALL(MV_) hipLaunchKernelGGL((ln_res_fwd_kernel<Th, MV_>), grid, block, 0, s, (const Th*)h, gamma, beta, x, ls, ss, rps, out, (Th*)oact, mean, rstd, rows, (int)C, eps)
        MTP_LN_MV(C, MTP_CALL);
#undef MTP_CALL
    } else {
        const dim3 grid((unsigned)mtp_layernorm_bwd_partial_rows(rows)), block(LN_THREADS);
#define MTP_CALL(MV_) hipLaunchKernelGGL((ln_res_bwd_kernel<Th, MV_>), grid, block, 0, s, (const float*)a0, (const Th*)h, mean_c, rstd_c, gamma, beta, ls, ss, rps, (Th*)dh, part, rows, (int)C)
        MTP_LN_MV(C, MTP_CALL);
#undef MTP_CALL
    }
    return mtp_launch_status();
}

extern "C" int mtp_layernorm_residual_fwd(const void* h, int dtype, const float* gamma, const float* beta, const float* x, const float* layer_scale,
                                          const float* sample_scale, int64_t rows_per_sample, float* out, void* out_act, float* mean, float* rstd,
                                          int64_t rows, int64_t C, float eps, mtp_stream_t stream) {
    if (!h || !gamma || !beta || !x || !layer_scale || !out || !mean || !rstd || rows <= 0 || rows > INT32_MAX || C <= 0 || (C % 4) || C > 256 * MAXV) return MTP_ERR_ARG;
    if (sample_scale && rows_per_sample <= 0) return MTP_ERR_ARG;
    const int rps = (int)(rows_per_sample > 0 ? rows_per_sample : 1);
    if (dtype == MTP_BF16) return launch_ln_res<bf16_t>(true, nullptr, h, nullptr, mean, nullptr, rstd, gamma, beta, x, layer_scale, sample_scale, rps, out, out_act, nullptr, nullptr, rows, C, eps, (hipStream_t)stream);
    if (dtype == MTP_F32) return launch_ln_res<float>(true, nullptr, h, nullptr, mean, nullptr, rstd, gamma, beta, x, layer_scale, sample_scale, rps, out, out_act, nullptr, nullptr, rows, C, eps, (hipStream_t)stream);
    return MTP_ERR_UNSUPPORTED;
}

extern "C" int mtp_layernorm_residual_bwd(const float* dout, const void* h, int dtype, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                          const float* layer_scale, const float* sample_scale, int64_t rows_per_sample, void* dh, float* part,
                                          int64_t rows, int64_t C, mtp_stream_t stream) {
    if (!dout || !h || !mean || !rstd || !gamma || !beta || !layer_scale || !dh || !part || rows <= 0 || rows > INT32_MAX || C <= 0 || (C % 4) || C > 256 * MAXV) return MTP_ERR_ARG;
    if (sample_scale && rows_per_sample <= 0) return MTP_ERR_ARG;
    const int rps = (int)(rows_per_sample > 0 ? rows_per_sample : 1);
    if (dtype == MTP_BF16) return launch_ln_res<bf16_t>(false, dout, h, mean, nullptr, rstd, nullptr, gamma, beta, nullptr, layer_scale, sample_scale, rps, nullptr, nullptr, dh, part, rows, C, 0.f, (hipStream_t)stream);
    if (dtype == MTP_F32) return launch_ln_res<float>(false, dout, h, mean, nullptr, rstd, nullptr, gamma, beta, nullptr, layer_scale, sample_scale, rps, nullptr, nullptr, dh, part, rows, C, 0.f, (hipStream_t)stream);
    return MTP_ERR_UNSUPPORTED;
}

extern "C" int64_t mtp_layernorm_bwd_partial_rows(int64_t rows) {
    // workgroups of the backward kernels = rows of their parameter-gradient partials.  512 (two per CU) up to 64 K rows -- every ViT shape; maps with more rows
    // (InternImage's 128 x 128 level and its stem at 512^2: 131 K / 524 K rows of 192 / 96 channels) get 1024 / 2048: their rows are short, the wave-serial row loop
    // is latency-bound and the partial buffer small (round 6)
    const int64_t nb = (rows + 3) / 4;
    const int64_t cap = rows > 262144 ? 2048 : rows > 65536 ? 1024 : 512;
    return nb < cap ? nb : cap;
}

extern "C" int mtp_layernorm_bwd(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                                 const float* gamma, const float* beta, int fuse_gelu, const float* dres, const float* extra, void* dx, int dx_dtype,
                                 void* dx_copy, int copy_dtype, const float* copy_scale, int64_t rows_per_sample,
                                 float* dgamma_part, float* dbeta_part, int64_t part_ld, int64_t rows, int64_t C, mtp_stream_t stream) {
    if (part_ld == 0) part_ld = C;
    if (part_ld < C || (part_ld % 4)) return MTP_ERR_ARG;
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma_part || !dbeta_part || rows <= 0 || rows > INT32_MAX || (C % 4) || C > 256 * MAXV) return MTP_ERR_ARG;
    if (fuse_gelu && !beta) return MTP_ERR_ARG;
    if (dx_copy && copy_dtype != dy_dtype) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dy_dtype == MTP_BF16 && x_dtype == MTP_F32 && dx_dtype == MTP_F32)
        return launch_ln_bwd<bf16_t, float, float>(dy, x, mean, rstd, gamma, beta, fuse_gelu, dres, extra, dx, dx_copy, copy_scale, rows_per_sample, dgamma_part, dbeta_part, part_ld, rows, C, s);
    if (dy_dtype == MTP_F32 && x_dtype == MTP_F32 && dx_dtype == MTP_F32)
        return launch_ln_bwd<float, float, float>(dy, x, mean, rstd, gamma, beta, fuse_gelu, dres, extra, dx, dx_copy, copy_scale, rows_per_sample, dgamma_part, dbeta_part, part_ld, rows, C, s);
    if (dy_dtype == MTP_BF16 && x_dtype == MTP_BF16 && dx_dtype == MTP_BF16)
        return launch_ln_bwd<bf16_t, bf16_t, bf16_t>(dy, x, mean, rstd, gamma, beta, fuse_gelu, dres, extra, dx, dx_copy, copy_scale, rows_per_sample, dgamma_part, dbeta_part, part_ld, rows, C, s);
    if (dy_dtype == MTP_F32 && x_dtype == MTP_BF16 && dx_dtype == MTP_BF16)      // f32 gradient straight out of a GEMM's f32 epilogue (InternImage's dw-conv branch): no cast pass
        return launch_ln_bwd<float, bf16_t, bf16_t>(dy, x, mean, rstd, gamma, beta, fuse_gelu, dres, extra, dx, dx_copy, copy_scale, rows_per_sample, dgamma_part, dbeta_part, part_ld, rows, C, s);
    return MTP_ERR_UNSUPPORTED;
}

extern "C" int mtp_layernorm_bwd_win(const void* dy, int dy_dtype, const void* x, int x_dtype, const float* mean, const float* rstd,
                                     const float* gamma, const float* dres, const float* extra, void* dx, int dx_dtype,
                                     void* dx_copy, int copy_dtype, const float* copy_scale, int64_t rows_per_sample,
                                     float* dgamma_part, float* dbeta_part, int64_t part_ld, int64_t rows, int64_t C,
                                     const float* win_add, int64_t B, int64_t Hp, int64_t Wp, mtp_stream_t stream) {
    if (part_ld == 0) part_ld = C;
    if (part_ld < C || (part_ld % 4)) return MTP_ERR_ARG;
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !dgamma_part || !dbeta_part || rows <= 0 || rows > INT32_MAX || (C % 4) || C > 256 * MAXV) return MTP_ERR_ARG;
    if (!win_add || B <= 0 || Hp <= 0 || Wp <= 0 || B * Hp * Wp != rows) return MTP_ERR_ARG;
    if (dx_copy && copy_dtype != dy_dtype) return MTP_ERR_ARG;
    const int pad_h = (int)((7 - Hp % 7) % 7), pad_w = (int)((7 - Wp % 7) % 7);
    const LnWin wg = {(int)Hp, (int)Wp, pad_h / 2, pad_w / 2, (int)((Hp + pad_h) / 7), (int)((Wp + pad_w) / 7)};
    hipStream_t s = (hipStream_t)stream;
    if (dy_dtype == MTP_BF16 && x_dtype == MTP_F32 && dx_dtype == MTP_F32)
        return launch_ln_bwd<bf16_t, float, float>(dy, x, mean, rstd, gamma, nullptr, 0, dres, extra, dx, dx_copy, copy_scale, rows_per_sample, dgamma_part, dbeta_part, part_ld, rows, C, s, win_add, wg);
    if (dy_dtype == MTP_F32 && x_dtype == MTP_F32 && dx_dtype == MTP_F32)
        return launch_ln_bwd<float, float, float>(dy, x, mean, rstd, gamma, nullptr, 0, dres, extra, dx, dx_copy, copy_scale, rows_per_sample, dgamma_part, dbeta_part, part_ld, rows, C, s, win_add, wg);
    return MTP_ERR_UNSUPPORTED;
}

extern "C" int mtp_reduce_rows_f32(const float* part, int64_t ld, float* out, int64_t rows, int64_t C, int accumulate, mtp_stream_t stream) {
    if (!part || !out || rows <= 0 || C <= 0 || ld < C) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)C, s);
        if (e != hipSuccess) return (int)e;
    }
    const int64_t col_blocks = (C + 255) / 256;
    int64_t splits = 1024 / col_blocks;
    if (splits < 1) splits = 1;
    if (splits > rows) splits = rows;
    const int64_t rpb = (rows + splits - 1) / splits;
    splits = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)col_blocks, (unsigned)splits), dim3(256), 0, s, part, ld, out, rows, C, rpb);
    return mtp_launch_status();
}

extern "C" int mtp_reduce_rows_batched_f32(const float* const* parts, float* const* outs, int n, int64_t ld, int64_t rows, int64_t C, int accumulate,
                                           mtp_stream_t stream) {
    if (!parts || !outs || n <= 0 || n > MTP_REDUCE_BATCH_MAX || rows <= 0 || C <= 0 || ld < C) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    RrBatch t;
    for (int i = 0; i < n; ++i) {
        if (!parts[i] || !outs[i]) return MTP_ERR_ARG;
        t.part[i] = parts[i];
        t.out[i] = outs[i];
    }
    if (!accumulate)
        for (int i = 0; i < n; ++i) {
            hipError_t e = hipMemsetAsync(outs[i], 0, sizeof(float) * (size_t)C, s);
            if (e != hipSuccess) return (int)e;
        }
    const int64_t col_blocks = (C + 255) / 256;
    // ~2048 workgroups for the WHOLE launch and at least 16 rows per thread (round 5).  Rounds 2-4 split every buffer as if it were alone (1024 / col_blocks
    // pieces each): ViT-L's bursts (n = 8 buffers of 512 x 2048 partial rows) became 8192 workgroups of four rows and one atomic each.  (The 77 us per launch
    // the InternImage-XL traces show for this kernel are as-run durations next to the weight-gradient burst it is queued behind -- 10 us of work)
    int64_t splits = 2048 / (col_blocks * n);
    if (splits > rows / 16) splits = rows / 16;
    if (splits < 1) splits = 1;
    const int64_t rpb = (rows + splits - 1) / splits;
    splits = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(reduce_rows_batched_kernel, dim3((unsigned)col_blocks, (unsigned)splits, (unsigned)n), dim3(256), 0, s, t, ld, rows, C, rpb);
    return mtp_launch_status();
}

extern "C" int mtp_reduce_rows_t_batched_f32(const float* const* parts, float* const* outs, int n, int64_t ld, int64_t rows, int64_t R, int64_t C, int accumulate,
                                             mtp_stream_t stream) {
    if (!parts || !outs || n <= 0 || n > MTP_REDUCE_BATCH_MAX || rows <= 0 || R <= 0 || C <= 0 || ld < R * C) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    RrBatch t;
    for (int i = 0; i < n; ++i) {
        if (!parts[i] || !outs[i]) return MTP_ERR_ARG;
        t.part[i] = parts[i];
        t.out[i] = outs[i];
        if (!accumulate) {
            hipError_t e = hipMemsetAsync(outs[i], 0, sizeof(float) * (size_t)(R * C), s);
            if (e != hipSuccess) return (int)e;
        }
    }
    const int64_t col_blocks = (R * C + 255) / 256;
    int64_t splits = 1024 / col_blocks;
    if (splits < 1) splits = 1;
    if (splits > rows) splits = rows;
    const int64_t rpb = (rows + splits - 1) / splits;
    splits = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(reduce_rows_t_batched_kernel, dim3((unsigned)col_blocks, (unsigned)splits, (unsigned)n), dim3(256), 0, s, t, ld, rows, (int)R, (int)C, rpb);
    return mtp_launch_status();
}

extern "C" int mtp_reduce_rows_t_f32(const float* part, int64_t ld, float* out, int64_t rows, int64_t R, int64_t C, int accumulate, mtp_stream_t stream) {
    if (!part || !out || rows <= 0 || R <= 0 || C <= 0 || ld < R * C) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)(R * C), s);
        if (e != hipSuccess) return (int)e;
    }
    const int64_t col_blocks = (R * C + 255) / 256;
    int64_t splits = 512 / col_blocks;      // >= 8 rows per thread: few atomics per output, enough loads in flight
    if (splits < 1) splits = 1;
    if (splits > (rows + 7) / 8) splits = (rows + 7) / 8;
    const int64_t rpb = (rows + splits - 1) / splits;
    splits = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(reduce_rows_t_kernel, dim3((unsigned)col_blocks, (unsigned)splits), dim3(256), 0, s, part, ld, out, rows, (int)R, (int)C, rpb);
    return mtp_launch_status();
}

static int colsum_impl(const void* dY, int dtype, int64_t ld, float* out, int64_t M, int64_t N, bool zero, mtp_stream_t stream);

extern "C" int mtp_colsum(const void* dY, int dtype, int64_t ld, float* out, int64_t M, int64_t N, mtp_stream_t stream) {
    return colsum_impl(dY, dtype, ld, out, M, N, true, stream);
}
// out[n] += sum_m dY[m][n]: no clearing pass (the training engine accumulates into a gradient buffer zeroed once per step)
extern "C" int mtp_colsum_acc(const void* dY, int dtype, int64_t ld, float* out, int64_t M, int64_t N, mtp_stream_t stream) {
    return colsum_impl(dY, dtype, ld, out, M, N, false, stream);
}

static int colsum_impl(const void* dY, int dtype, int64_t ld, float* out, int64_t M, int64_t N, bool zero, mtp_stream_t stream) {
    if (!dY || !out || M <= 0 || N <= 0 || (N % 4) || (ld % 4)) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (zero) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, s);
        if (e != hipSuccess) return (int)e;
    }
    const int64_t col_blocks = (N / 4 + 63) / 64;
    int64_t row_blocks = 1024 / col_blocks;   // >= 32 rows per thread: enough loads in flight to stream at HBM rate
    if (row_blocks < 1) row_blocks = 1;
    int64_t rpb = (M + row_blocks - 1) / row_blocks;
    rpb = (rpb + 3) / 4 * 4;
    row_blocks = (M + rpb - 1) / rpb;
    dim3 grid((unsigned)col_blocks, (unsigned)row_blocks), block(256);
    if (dtype == MTP_BF16)
        hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)dY, ld, out, M, N, rpb);
    else
        hipLaunchKernelGGL((colsum_kernel<float>), grid, block, 0, s, (const float*)dY, ld, out, M, N, rpb);
    return mtp_launch_status();
}
