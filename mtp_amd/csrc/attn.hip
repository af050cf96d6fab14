// Attention cores of the MTP backbone for gfx950 (head_dim = 64):
//   * full MHSA with decomposed relative position (Attention.forward, VIT:90-111 + calc_rel_pos_spatial VIT:142-193)
//   * rotated varied-size window attention (RotatedVariedSizeWindowAttention.forward, VIT:287-433): the sampling
//     grid is built in registers from the closed form (5 scalars per (image, window, head)), K/V rows are bilinearly
//     gathered straight from the token-major qkv buffer (replaces 2 grid_sample + ~14 permute/copy passes).
// Round-1 implementation: f32 VALU math with LDS-broadcast operands, I/O in bf16 or f32 (attention is ~1 % of the
// path's FLOPs; MFMA tiles are the next step).  lane = query in the score/softmax/PV phases, lane = key in the dK/dV
// phases, so no cross-lane reductions are needed on the hot loops.
#include <stdlib.h>

#include "attn_mfma.h"
#include "common.h"

namespace {

constexpr int HD = 64;
constexpr int KT = 32;  // keys (or queries) staged per LDS chunk in the full-attention kernels

template <typename T>
__device__ __forceinline__ void load_row(const T* p, float (&r)[HD], float mul = 1.0f) {
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) {
        float t[8];
        load8(p + 8 * i, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) r[8 * i + e] = t[e] * mul;
    }
}
template <typename T>
__device__ __forceinline__ void store_row(T* p, const float (&r)[HD], float mul = 1.0f) {
#pragma unroll
    for (int i = 0; i < HD / 8; ++i) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = r[8 * i + e] * mul;
        store8(p + 8 * i, t);
    }
}
__device__ __forceinline__ float dot_lds(const float (&q)[HD], const float* row) {   // row: LDS, wave-uniform address
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 k = *reinterpret_cast<const float4*>(row + 4 * i);
        s += q[4 * i] * k.x + q[4 * i + 1] * k.y + q[4 * i + 2] * k.z + q[4 * i + 3] * k.w;
    }
    return s;
}
__device__ __forceinline__ void axpy_lds(float (&acc)[HD], float a, const float* row) {
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 k = *reinterpret_cast<const float4*>(row + 4 * i);
        acc[4 * i] += a * k.x; acc[4 * i + 1] += a * k.y; acc[4 * i + 2] += a * k.z; acc[4 * i + 3] += a * k.w;
    }
}
__device__ __forceinline__ float dot_glb(const float (&q)[HD], const float* __restrict__ row) {   // f32 table row in global
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 k = *reinterpret_cast<const float4*>(row + 4 * i);
        s += q[4 * i] * k.x + q[4 * i + 1] * k.y + q[4 * i + 2] * k.z + q[4 * i + 3] * k.w;
    }
    return s;
}
__device__ __forceinline__ void axpy_glb(float (&acc)[HD], float a, const float* __restrict__ row) {
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 k = *reinterpret_cast<const float4*>(row + 4 * i);
        acc[4 * i] += a * k.x; acc[4 * i + 1] += a * k.y; acc[4 * i + 2] += a * k.z; acc[4 * i + 3] += a * k.w;
    }
}

// stage `cnt` rows (row r -> token base+r) of 64 channels at channel offset `coff` into LDS as f32 [KT][64]; 256 threads
template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ src, int64_t row_stride, int64_t first_row, int cnt, float* dst, float mul, int tid) {
    const int r = tid >> 3, d8 = (tid & 7) * 8;
    float t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < cnt) load8(src + (first_row + r) * row_stride + d8, t);
    *reinterpret_cast<float4*>(dst + r * HD + d8) = make_float4(t[0] * mul, t[1] * mul, t[2] * mul, t[3] * mul);
    *reinterpret_cast<float4*>(dst + r * HD + d8 + 4) = make_float4(t[4] * mul, t[5] * mul, t[6] * mul, t[7] * mul);
}

// =====================================================================================================================
// Full attention forward.  grid (B*heads, ceil(N/256)), 256 threads, thread = query.
// LDS floats: Ks[KT*64] | Vs[KT*64] | qr[(Hp+Wp)*256] | sbuf[KT*256]
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void full_attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ lse,
                                                           const float* __restrict__ rel_h, const float* __restrict__ rel_w,
                                                           int N, int Hp, int Wp, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ks = sm;
    float* Vs = Ks + KT * HD;
    float* qr = Vs + KT * HD;
    float* sbuf = qr + (Hp + Wp) * 256;
    const int tid = threadIdx.x;
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int C = heads * HD;
    const int64_t ld = 3 * (int64_t)C;
    const int n = blockIdx.y * 256 + tid;
    const bool valid = n < N;
    const T* base = qkv + (int64_t)b * N * ld + h * HD;

    float qs[HD];
    if (valid) load_row(base + (int64_t)n * ld, qs, scale);
    else {
#pragma unroll
        for (int d = 0; d < HD; ++d) qs[d] = 0.f;
    }
    const int hq = valid ? n / Wp : 0, wq = valid ? n % Wp : 0;
    for (int kh = 0; kh < Hp; ++kh) qr[kh * 256 + tid] = valid ? dot_glb(qs, rel_h + (hq - kh + Hp - 1) * HD) : 0.f;
    for (int kw = 0; kw < Wp; ++kw) qr[(Hp + kw) * 256 + tid] = valid ? dot_glb(qs, rel_w + (wq - kw + Wp - 1) * HD) : 0.f;

    float m = -INFINITY, l = 0.f, acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;

    for (int c0 = 0; c0 < N; c0 += KT) {
        const int cnt = (N - c0) < KT ? (N - c0) : KT;
        __syncthreads();
        stage_rows(base + C, ld, c0, cnt, Ks, 1.0f, tid);
        stage_rows(base + 2 * C, ld, c0, cnt, Vs, 1.0f, tid);
        __syncthreads();
        float cmax = -INFINITY;
        int hj = c0 / Wp, wj = c0 % Wp;
        for (int jj = 0; jj < cnt; ++jj) {
            const float s = dot_lds(qs, Ks + jj * HD) + qr[hj * 256 + tid] + qr[(Hp + wj) * 256 + tid];
            sbuf[jj * 256 + tid] = s;
            cmax = fmaxf(cmax, s);
            if (++wj == Wp) { wj = 0; ++hj; }
        }
        const float mnew = fmaxf(m, cmax);
        const float alpha = __expf(m - mnew);
        l *= alpha;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] *= alpha;
        m = mnew;
        for (int jj = 0; jj < cnt; ++jj) {
            const float p = __expf(sbuf[jj * 256 + tid] - m);
            l += p;
            axpy_lds(acc, p, Vs + jj * HD);
        }
    }
    if (valid) {
        store_row(o + ((int64_t)b * N + n) * C + h * HD, acc, 1.0f / l);
        lse[(int64_t)bh * N + n] = m + __logf(l);
    }
}

// =====================================================================================================================
// Full attention backward.  grid (B*heads), 256 threads, N <= 256.
// LDS floats: qr[(Hp+Wp)*256] | dqr[(Hp+Wp)*256] | delta[256] | lses[256] | cA[KT*64] | cB[KT*64]
// =====================================================================================================================
constexpr int MAXR = 16;   // table rows per thread in the table-gradient phase: (2Hp-1)+(2Wp-1) <= 64

template <typename T>
__global__ __launch_bounds__(256) void full_attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ dout,
                                                           const float* __restrict__ lse, T* __restrict__ dqkv,
                                                           const float* __restrict__ rel_h, const float* __restrict__ rel_w, float* __restrict__ drel_part,
                                                           int N, int Hp, int Wp, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HW = Hp + Wp;
    float* qr = sm;
    float* dqr = qr + HW * 256;
    float* delta = dqr + HW * 256;
    float* lses = delta + 256;
    float* cA = lses + 256;
    float* cB = cA + KT * HD;
    const int tid = threadIdx.x;
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int C = heads * HD;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)b * N * ld + h * HD;
    const T* dob = dout + (int64_t)b * N * C + h * HD;
    const bool valid = tid < N;
    const int hq = valid ? tid / Wp : 0, wq = valid ? tid % Wp : 0;

    // ---------------- phase Q: thread = query --------------------------------------------------------------------
    {
        float qs[HD], dO[HD], dq[HD];
        float dl = 0.f, ls = 0.f;
        if (valid) {
            load_row(base + (int64_t)tid * ld, qs, scale);
            load_row(dob + (int64_t)tid * C, dO);
            float ov[HD];
            load_row(o + ((int64_t)b * N + tid) * C + h * HD, ov);
#pragma unroll
            for (int d = 0; d < HD; ++d) dl += dO[d] * ov[d];
            ls = lse[(int64_t)bh * N + tid];
        } else {
#pragma unroll
            for (int d = 0; d < HD; ++d) { qs[d] = 0.f; dO[d] = 0.f; }
        }
        delta[tid] = dl;
        lses[tid] = ls;
        for (int kh = 0; kh < Hp; ++kh) {
            qr[kh * 256 + tid] = valid ? dot_glb(qs, rel_h + (hq - kh + Hp - 1) * HD) : 0.f;
            dqr[kh * 256 + tid] = 0.f;
        }
        for (int kw = 0; kw < Wp; ++kw) {
            qr[(Hp + kw) * 256 + tid] = valid ? dot_glb(qs, rel_w + (wq - kw + Wp - 1) * HD) : 0.f;
            dqr[(Hp + kw) * 256 + tid] = 0.f;
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) dq[d] = 0.f;
        for (int c0 = 0; c0 < N; c0 += KT) {
            const int cnt = (N - c0) < KT ? (N - c0) : KT;
            __syncthreads();
            stage_rows(base + C, ld, c0, cnt, cA, 1.0f, tid);
            stage_rows(base + 2 * C, ld, c0, cnt, cB, 1.0f, tid);
            __syncthreads();
            int hj = c0 / Wp, wj = c0 % Wp;
            for (int jj = 0; jj < cnt; ++jj) {
                const float s = dot_lds(qs, cA + jj * HD) + qr[hj * 256 + tid] + qr[(Hp + wj) * 256 + tid];
                const float p = valid ? __expf(s - ls) : 0.f;
                const float ds = p * (dot_lds(dO, cB + jj * HD) - dl);
                axpy_lds(dq, ds, cA + jj * HD);
                dqr[hj * 256 + tid] += ds;
                dqr[(Hp + wj) * 256 + tid] += ds;
                if (++wj == Wp) { wj = 0; ++hj; }
            }
        }
        if (valid) {
            for (int kh = 0; kh < Hp; ++kh) axpy_glb(dq, dqr[kh * 256 + tid], rel_h + (hq - kh + Hp - 1) * HD);
            for (int kw = 0; kw < Wp; ++kw) axpy_glb(dq, dqr[(Hp + kw) * 256 + tid], rel_w + (wq - kw + Wp - 1) * HD);
            store_row(dqkv + ((int64_t)b * N + tid) * ld + h * HD, dq, scale);
        }
    }
    __syncthreads();

    // ---------------- phase T: rel-pos table gradients, thread = (row group, channel) -------------------------------
    {
        const int d = tid & 63, rg = tid >> 6;
        const int RH = 2 * Hp - 1, RT = RH + 2 * Wp - 1;
        float acc[MAXR];
#pragma unroll
        for (int i = 0; i < MAXR; ++i) acc[i] = 0.f;
        int hn = 0, wn = 0;
        for (int n = 0; n < N; ++n) {
            const float qv = scale * Elem<T>::load(base + (int64_t)n * ld + d);
#pragma unroll
            for (int i = 0; i < MAXR; ++i) {
                const int r = rg + 4 * i;
                if (r < RH) {
                    const int kh = hn - r + Hp - 1;
                    if (kh >= 0 && kh < Hp) acc[i] += dqr[kh * 256 + n] * qv;
                } else if (r < RT) {
                    const int kw = wn - (r - RH) + Wp - 1;
                    if (kw >= 0 && kw < Wp) acc[i] += dqr[(Hp + kw) * 256 + n] * qv;
                }
            }
            if (++wn == Wp) { wn = 0; ++hn; }
        }
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int r = rg + 4 * i;
            if (r < RT) drel_part[((int64_t)bh * RT + r) * HD + d] = acc[i];
        }
    }

    // ---------------- phase K: thread = key -----------------------------------------------------------------------------
    {
        float kj[HD], vj[HD], dk[HD], dv[HD];
        if (valid) {
            load_row(base + C + (int64_t)tid * ld, kj);
            load_row(base + 2 * C + (int64_t)tid * ld, vj);
        } else {
#pragma unroll
            for (int d = 0; d < HD; ++d) { kj[d] = 0.f; vj[d] = 0.f; }
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        for (int c0 = 0; c0 < N; c0 += KT) {
            const int cnt = (N - c0) < KT ? (N - c0) : KT;
            __syncthreads();
            stage_rows(base, ld, c0, cnt, cA, scale, tid);          // scaled queries
            stage_rows(dob, (int64_t)C, c0, cnt, cB, 1.0f, tid);    // dO rows
            __syncthreads();
            for (int nn = 0; nn < cnt; ++nn) {
                const int n = c0 + nn;
                const float s = dot_lds(kj, cA + nn * HD) + qr[hq * 256 + n] + qr[(Hp + wq) * 256 + n];   // hq/wq = this KEY's row/col
                const float p = valid ? __expf(s - lses[n]) : 0.f;
                const float ds = p * (dot_lds(vj, cB + nn * HD) - delta[n]);
                axpy_lds(dk, ds, cA + nn * HD);
                axpy_lds(dv, p, cB + nn * HD);
            }
        }
        if (valid) {
            store_row(dqkv + ((int64_t)b * N + tid) * ld + C + h * HD, dk);
            store_row(dqkv + ((int64_t)b * N + tid) * ld + 2 * C + h * HD, dv);
        }
    }
}

// =====================================================================================================================
// Full attention backward for token grids beyond one workgroup (N > 256: 448^2 pretraining inputs -> 28 x 28, 512^2 -> 32 x 32).
// Same f32 math as the kernel above, split into three launches with the per-query quantities in a caller workspace:
//   ws floats per (image, head): qr[Hp+Wp][N] | dqr[Hp+Wp][N] | delta[N]
// Functional path (VALU); the MFMA kernels cover N <= 256, a flash-style MFMA version of this is the next step.
// =====================================================================================================================
// pass Q: thread = query (blockIdx.y * 256 + tid), streams all keys.  LDS: qr[(Hp+Wp)*256] | dqr[(Hp+Wp)*256] | cA | cB
template <typename T>
__global__ __launch_bounds__(256) void full_attn_bwdN_q_kernel(const T* __restrict__ qkv, const T* __restrict__ o, const T* __restrict__ dout,
                                                              const float* __restrict__ lse, T* __restrict__ dqkv,
                                                              const float* __restrict__ rel_h, const float* __restrict__ rel_w, float* __restrict__ ws,
                                                              int N, int Hp, int Wp, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HW = Hp + Wp;
    float* qr = sm;
    float* dqr = qr + HW * 256;
    float* cA = dqr + HW * 256;
    float* cB = cA + KT * HD;
    const int tid = threadIdx.x;
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int C = heads * HD;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)b * N * ld + h * HD;
    const T* dob = dout + (int64_t)b * N * C + h * HD;
    const int n = blockIdx.y * 256 + tid;
    const bool valid = n < N;
    const int nc = valid ? n : N - 1;
    const int hq = nc / Wp, wq = nc % Wp;
    float* wq_r = ws + (int64_t)bh * (2 * HW + 1) * N;   // qr rows, then dqr rows, then delta
    float qs[HD], dO[HD], dq[HD];
    load_row(base + (int64_t)nc * ld, qs, scale);
    load_row(dob + (int64_t)nc * C, dO);
    float dl = 0.f;
    {
        float ov[HD];
        load_row(o + ((int64_t)b * N + nc) * C + h * HD, ov);
#pragma unroll
        for (int d = 0; d < HD; ++d) dl += dO[d] * ov[d];
    }
    const float ls = lse[(int64_t)bh * N + nc];
    for (int kh = 0; kh < Hp; ++kh) {
        qr[kh * 256 + tid] = dot_glb(qs, rel_h + (hq - kh + Hp - 1) * HD);
        dqr[kh * 256 + tid] = 0.f;
    }
    for (int kw = 0; kw < Wp; ++kw) {
        qr[(Hp + kw) * 256 + tid] = dot_glb(qs, rel_w + (wq - kw + Wp - 1) * HD);
        dqr[(Hp + kw) * 256 + tid] = 0.f;
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = 0.f;
    for (int c0 = 0; c0 < N; c0 += KT) {
        const int cnt = (N - c0) < KT ? (N - c0) : KT;
        __syncthreads();
        stage_rows(base + C, ld, c0, cnt, cA, 1.0f, tid);
        stage_rows(base + 2 * C, ld, c0, cnt, cB, 1.0f, tid);
        __syncthreads();
        int hj = c0 / Wp, wj = c0 % Wp;
        for (int jj = 0; jj < cnt; ++jj) {
            const float s = dot_lds(qs, cA + jj * HD) + qr[hj * 256 + tid] + qr[(Hp + wj) * 256 + tid];
            const float p = valid ? __expf(s - ls) : 0.f;
            const float ds = p * (dot_lds(dO, cB + jj * HD) - dl);
            axpy_lds(dq, ds, cA + jj * HD);
            dqr[hj * 256 + tid] += ds;
            dqr[(Hp + wj) * 256 + tid] += ds;
            if (++wj == Wp) { wj = 0; ++hj; }
        }
    }
    if (valid) {
        for (int kh = 0; kh < Hp; ++kh) axpy_glb(dq, dqr[kh * 256 + tid], rel_h + (hq - kh + Hp - 1) * HD);
        for (int kw = 0; kw < Wp; ++kw) axpy_glb(dq, dqr[(Hp + kw) * 256 + tid], rel_w + (wq - kw + Wp - 1) * HD);
        store_row(dqkv + ((int64_t)b * N + n) * ld + h * HD, dq, scale);
        for (int r = 0; r < HW; ++r) {
            wq_r[(int64_t)r * N + n] = qr[r * 256 + tid];
            wq_r[(int64_t)(HW + r) * N + n] = dqr[r * 256 + tid];
        }
        wq_r[(int64_t)2 * HW * N + n] = dl;
    }
}

// pass T: rel-pos table gradients.  grid (B*heads, ceil(RT/4)); thread = (table row blockIdx.y*4 + tid>>6, channel tid&63)
template <typename T>
__global__ __launch_bounds__(256) void full_attn_bwdN_t_kernel(const T* __restrict__ qkv, const float* __restrict__ ws, float* __restrict__ drel_part,
                                                              int N, int Hp, int Wp, int heads, float scale) {
    const int tid = threadIdx.x, d = tid & 63, r = blockIdx.y * 4 + (tid >> 6);
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int C = heads * HD, HW = Hp + Wp;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)b * N * ld + h * HD;
    const float* dqr = ws + (int64_t)bh * (2 * HW + 1) * N + (int64_t)HW * N;
    const int RH = 2 * Hp - 1, RT = RH + 2 * Wp - 1;
    if (r >= RT) return;
    float acc = 0.f;
    int hn = 0, wn = 0;
    for (int n = 0; n < N; ++n) {
        const int kk = r < RH ? hn - r + Hp - 1 : wn - (r - RH) + Wp - 1;
        const int lim = r < RH ? Hp : Wp;
        if (kk >= 0 && kk < lim) acc += dqr[(int64_t)((r < RH ? 0 : Hp) + kk) * N + n] * (scale * Elem<T>::load(base + (int64_t)n * ld + d));
        if (++wn == Wp) { wn = 0; ++hn; }
    }
    drel_part[((int64_t)bh * RT + r) * HD + d] = acc;
}

// pass K: thread = key (blockIdx.y * 256 + tid), streams all queries.  LDS: cA | cB | qrt[(Hp+Wp)][KT] | ls[KT] | dl[KT]
template <typename T>
__global__ __launch_bounds__(256) void full_attn_bwdN_k_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, const float* __restrict__ lse,
                                                              T* __restrict__ dqkv, const float* __restrict__ ws,
                                                              int N, int Hp, int Wp, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HW = Hp + Wp;
    float* cA = sm;
    float* cB = cA + KT * HD;
    float* qrt = cB + KT * HD;
    float* lst = qrt + HW * KT;
    float* dlt = lst + KT;
    const int tid = threadIdx.x;
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int C = heads * HD;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)b * N * ld + h * HD;
    const T* dob = dout + (int64_t)b * N * C + h * HD;
    const float* wq_r = ws + (int64_t)bh * (2 * HW + 1) * N;
    const int key = blockIdx.y * 256 + tid;
    const bool valid = key < N;
    const int kc = valid ? key : N - 1;
    const int hk = kc / Wp, wk = kc % Wp;
    float kj[HD], vj[HD], dk[HD], dv[HD];
    load_row(base + C + (int64_t)kc * ld, kj);
    load_row(base + 2 * C + (int64_t)kc * ld, vj);
#pragma unroll
    for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int c0 = 0; c0 < N; c0 += KT) {
        const int cnt = (N - c0) < KT ? (N - c0) : KT;
        __syncthreads();
        stage_rows(base, ld, c0, cnt, cA, scale, tid);          // scaled queries
        stage_rows(dob, (int64_t)C, c0, cnt, cB, 1.0f, tid);    // dO rows
        for (int i = tid; i < HW * KT; i += 256) {
            const int r = i / KT, nn = i % KT;
            qrt[i] = nn < cnt ? wq_r[(int64_t)r * N + c0 + nn] : 0.f;
        }
        if (tid < KT) {
            lst[tid] = tid < cnt ? lse[(int64_t)bh * N + c0 + tid] : 0.f;
            dlt[tid] = tid < cnt ? wq_r[(int64_t)2 * HW * N + c0 + tid] : 0.f;
        }
        __syncthreads();
        for (int nn = 0; nn < cnt; ++nn) {
            const float s = dot_lds(kj, cA + nn * HD) + qrt[hk * KT + nn] + qrt[(Hp + wk) * KT + nn];
            const float p = valid ? __expf(s - lst[nn]) : 0.f;
            const float ds = p * (dot_lds(vj, cB + nn * HD) - dlt[nn]);
            axpy_lds(dk, ds, cA + nn * HD);
            axpy_lds(dv, p, cB + nn * HD);
        }
    }
    if (valid) {
        store_row(dqkv + ((int64_t)b * N + key) * ld + C + h * HD, dk);
        store_row(dqkv + ((int64_t)b * N + key) * ld + 2 * C + h * HD, dv);
    }
}

// =====================================================================================================================
// RVSA geometry shared by forward and backward
// =====================================================================================================================
struct RvsaGeom {
    int Hp, Wp, He, We, pad_t, pad_l, nh, nw, heads;
    float inv_div_x, inv_div_y;
};

struct Sample {         // one key position's sampling footprint
    float ix, iy, fx, fy;
    int x0, y0;
    float rx, ry, cs, sn, relx, rely;
};

__device__ __forceinline__ Sample make_sample(const RvsaGeom& g, const float* __restrict__ sp, int h, int wi, int wj, int a, int bb) {
    Sample s;
    const int H = g.heads;
    const float offx = sp[2 * h] * g.inv_div_x, offy = sp[2 * h + 1] * g.inv_div_y;
    const float sx = sp[2 * H + 2 * h] + 1.0f, sy = sp[2 * H + 2 * h + 1] + 1.0f;
    const float ang = sp[4 * H + h];
    const float stepx = 2.0f / (float)(g.We - 1), stepy = 2.0f / (float)(g.He - 1);
    const float cenx = -1.0f + stepx * (float)(7 * wj + 3), ceny = -1.0f + stepy * (float)(7 * wi + 3);   // mean of 7 linspace points
    s.relx = (float)(bb - 3) * stepx;
    s.rely = (float)(a - 3) * stepy;
    s.rx = s.relx * sx;
    s.ry = s.rely * sy;
    s.cs = cosf(ang);
    s.sn = sinf(ang);
    const float gx = cenx + (s.rx * s.cs - s.ry * s.sn) + offx;
    const float gy = ceny + (s.ry * s.cs + s.rx * s.sn) + offy;
    float ix = (gx + 1.0f) * 0.5f * (float)(g.We - 1), iy = (gy + 1.0f) * 0.5f * (float)(g.He - 1);
    ix = fminf(fmaxf(ix, -4.0f), (float)g.We + 4.0f);   // far-out samples contribute 0 anyway; keeps floor() in int range
    iy = fminf(fmaxf(iy, -4.0f), (float)g.He + 4.0f);
    s.ix = ix; s.iy = iy;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    s.x0 = (int)fx0; s.y0 = (int)fy0;
    s.fx = ix - fx0; s.fy = iy - fy0;
    return s;
}
// neighbour k in {0:(x0,y0), 1:(x1,y0), 2:(x0,y1), 3:(x1,y1)}: bilinear weight, and token index (or -1 when the neighbour
// is outside the padded map [zeros padding of grid_sample] or inside the zero padding ring of the map itself)
__device__ __forceinline__ int neighbour(const RvsaGeom& g, const Sample& s, int k, float& w, bool& in_map) {
    const int dx = k & 1, dy = k >> 1;
    const int xi = s.x0 + dx, yi = s.y0 + dy;
    w = (dx ? s.fx : 1.0f - s.fx) * (dy ? s.fy : 1.0f - s.fy);
    in_map = xi >= 0 && xi <= g.We - 1 && yi >= 0 && yi <= g.He - 1;
    const int tx = xi - g.pad_l, ty = yi - g.pad_t;
    if (!in_map || tx < 0 || tx >= g.Wp || ty < 0 || ty >= g.Hp) return -1;
    return ty * g.Wp + tx;
}

// =====================================================================================================================
// RVSA forward.  grid (B*nW*heads), 64 threads (one wave): lane = key for the gather, lane = query afterwards.
// LDS floats: Ksel[49*64] | Vsel[49*64] | relq[14*64] | sbuf[49*64] | tab[176]
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(64) void rvsa_attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ samp, T* __restrict__ o, float* __restrict__ lse,
                                                          const float* __restrict__ rel_h, const float* __restrict__ rel_w, const float* __restrict__ bias_table,
                                                          RvsaGeom g, float scale) {
    __shared__ __attribute__((aligned(16))) float Ksel[49 * HD];
    __shared__ __attribute__((aligned(16))) float Vsel[49 * HD];
    __shared__ float relq[14 * 64];
    __shared__ float sbuf[49 * 64];
    __shared__ float tab[176];
    const int lane = threadIdx.x;
    const int H = g.heads, nW = g.nh * g.nw;
    const int h = blockIdx.x % H, bw = blockIdx.x / H, b = bw / nW, win = bw % nW, wi = win / g.nw, wj = win % g.nw;
    const int C = H * HD, N = g.Hp * g.Wp;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)b * N * ld + h * HD;
    const bool active = lane < 49;
    const int a = active ? lane / 7 : 0, bb = active ? lane % 7 : 0;

    for (int i = lane; i < 169; i += 64) tab[i] = bias_table[i * H + h];

    float q[HD];
    int qtok = -1;
    {
        float ks[HD], vs[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { ks[d] = 0.f; vs[d] = 0.f; q[d] = 0.f; }
        if (active) {
            const Sample s = make_sample(g, samp + (int64_t)bw * 5 * H, h, wi, wj, a, bb);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float w; bool in_map;
                const int tok = neighbour(g, s, k, w, in_map);
                if (tok >= 0) {
                    float t[HD];
                    load_row(base + C + (int64_t)tok * ld, t);
#pragma unroll
                    for (int d = 0; d < HD; ++d) ks[d] += w * t[d];
                    load_row(base + 2 * C + (int64_t)tok * ld, t);
#pragma unroll
                    for (int d = 0; d < HD; ++d) vs[d] += w * t[d];
                }
            }
            const int ty = 7 * wi + a - g.pad_t, tx = 7 * wj + bb - g.pad_l;
            if (ty >= 0 && ty < g.Hp && tx >= 0 && tx < g.Wp) {
                qtok = ty * g.Wp + tx;
                load_row(base + (int64_t)qtok * ld, q);
            }
#pragma unroll
            for (int i = 0; i < HD / 4; ++i) {
                *reinterpret_cast<float4*>(Ksel + lane * HD + 4 * i) = make_float4(ks[4 * i], ks[4 * i + 1], ks[4 * i + 2], ks[4 * i + 3]);
                *reinterpret_cast<float4*>(Vsel + lane * HD + 4 * i) = make_float4(vs[4 * i], vs[4 * i + 1], vs[4 * i + 2], vs[4 * i + 3]);
            }
        }
    }
    for (int kk = 0; kk < 7; ++kk) {
        relq[kk * 64 + lane] = dot_glb(q, rel_h + (a - kk + 6) * HD);
        relq[(7 + kk) * 64 + lane] = dot_glb(q, rel_w + (bb - kk + 6) * HD);
    }
    __syncthreads();

    float m = -INFINITY;
    {
        int aj = 0, bj = 0;
        for (int j = 0; j < 49; ++j) {
            const float s = scale * dot_lds(q, Ksel + j * HD) + relq[aj * 64 + lane] + relq[(7 + bj) * 64 + lane] + tab[(a - aj + 6) * 13 + (bb - bj + 6)];
            sbuf[j * 64 + lane] = s;
            m = fmaxf(m, s);
            if (++bj == 7) { bj = 0; ++aj; }
        }
    }
    float l = 0.f, acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    for (int j = 0; j < 49; ++j) {
        const float p = __expf(sbuf[j * 64 + lane] - m);
        l += p;
        axpy_lds(acc, p, Vsel + j * HD);
    }
    if (active) {
        lse[(int64_t)blockIdx.x * 49 + lane] = m + __logf(l);
        if (qtok >= 0) store_row(o + ((int64_t)b * N + qtok) * C + h * HD, acc, 1.0f / l);
    }
}

// =====================================================================================================================
// RVSA backward.  grid (B*nW*heads), 64 threads.
// LDS floats: R1[2*49*64] (Ksel|Vsel, later Q|dO) | relq[14*64] | dqr[14*64] | tab[176] | dtab[176] | lses[64] | delta[64]
// =====================================================================================================================
template <typename T>
__global__ __launch_bounds__(64) void rvsa_attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ samp, const T* __restrict__ o, const T* __restrict__ dout,
                                                          const float* __restrict__ lse, T* __restrict__ dqkv, float* __restrict__ dkv, float* __restrict__ dsamp,
                                                          float* __restrict__ rel_part, float* __restrict__ tab_part,
                                                          const float* __restrict__ rel_h, const float* __restrict__ rel_w, const float* __restrict__ bias_table,
                                                          RvsaGeom g, float scale) {
    __shared__ __attribute__((aligned(16))) float R1[2 * 49 * HD];
    __shared__ float relq[14 * 64];
    __shared__ float dqr[14 * 64];
    __shared__ float tab[176];
    __shared__ float dtab[176];
    __shared__ float lses[64];
    __shared__ float delta[64];
    float* Ksel = R1;
    float* Vsel = R1 + 49 * HD;
    const int lane = threadIdx.x;
    const int H = g.heads, nW = g.nh * g.nw;
    const int h = blockIdx.x % H, bw = blockIdx.x / H, b = bw / nW, win = bw % nW, wi = win / g.nw, wj = win % g.nw;
    const int C = H * HD, N = g.Hp * g.Wp;
    const int64_t ld = 3 * (int64_t)C;
    const T* base = qkv + (int64_t)b * N * ld + h * HD;
    const bool active = lane < 49;
    const int a = active ? lane / 7 : 0, bb = active ? lane % 7 : 0;

    for (int i = lane; i < 176; i += 64) {
        tab[i] = i < 169 ? bias_table[i * H + h] : 0.f;
        dtab[i] = 0.f;
    }

    // ---- gather (lane = key): publish this key's sampled K/V for phase Q (reloaded into registers for phase K)
    float q[HD], dO[HD];
    Sample smp;
    int qtok = -1;
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = 0.f; dO[d] = 0.f; }
    float dl = 0.f, ls = 0.f;
    if (active) {
        smp = make_sample(g, samp + (int64_t)bw * 5 * H, h, wi, wj, a, bb);
        {
            float ks[HD], vs[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) { ks[d] = 0.f; vs[d] = 0.f; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float w; bool in_map;
                const int tok = neighbour(g, smp, k, w, in_map);
                if (tok >= 0) {
                    float t[HD];
                    load_row(base + C + (int64_t)tok * ld, t);
#pragma unroll
                    for (int d = 0; d < HD; ++d) ks[d] += w * t[d];
                    load_row(base + 2 * C + (int64_t)tok * ld, t);
#pragma unroll
                    for (int d = 0; d < HD; ++d) vs[d] += w * t[d];
                }
            }
#pragma unroll
            for (int i = 0; i < HD / 4; ++i) {
                *reinterpret_cast<float4*>(Ksel + lane * HD + 4 * i) = make_float4(ks[4 * i], ks[4 * i + 1], ks[4 * i + 2], ks[4 * i + 3]);
                *reinterpret_cast<float4*>(Vsel + lane * HD + 4 * i) = make_float4(vs[4 * i], vs[4 * i + 1], vs[4 * i + 2], vs[4 * i + 3]);
            }
        }
        const int ty = 7 * wi + a - g.pad_t, tx = 7 * wj + bb - g.pad_l;
        if (ty >= 0 && ty < g.Hp && tx >= 0 && tx < g.Wp) {
            qtok = ty * g.Wp + tx;
            load_row(base + (int64_t)qtok * ld, q);
            load_row(dout + ((int64_t)b * N + qtok) * C + h * HD, dO);
            float ov[HD];
            load_row(o + ((int64_t)b * N + qtok) * C + h * HD, ov);
#pragma unroll
            for (int d = 0; d < HD; ++d) dl += dO[d] * ov[d];
        }
        ls = lse[(int64_t)blockIdx.x * 49 + lane];
    }
    lses[lane] = ls;
    delta[lane] = dl;
    for (int kk = 0; kk < 7; ++kk) {
        relq[kk * 64 + lane] = dot_glb(q, rel_h + (a - kk + 6) * HD);
        relq[(7 + kk) * 64 + lane] = dot_glb(q, rel_w + (bb - kk + 6) * HD);
        dqr[kk * 64 + lane] = 0.f;
        dqr[(7 + kk) * 64 + lane] = 0.f;
    }
    __syncthreads();

    // ---- phase Q (lane = query): dq, rel-pos accumulators, bias-table gradient
    {
        float dq[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) dq[d] = 0.f;
        int aj = 0, bj = 0;
        for (int j = 0; j < 49; ++j) {
            const int ti = (a - aj + 6) * 13 + (bb - bj + 6);
            const float s = scale * dot_lds(q, Ksel + j * HD) + relq[aj * 64 + lane] + relq[(7 + bj) * 64 + lane] + tab[ti];
            const float p = active ? __expf(s - ls) : 0.f;
            const float ds = p * (dot_lds(dO, Vsel + j * HD) - dl);
            axpy_lds(dq, ds * scale, Ksel + j * HD);
            dqr[aj * 64 + lane] += ds;
            dqr[(7 + bj) * 64 + lane] += ds;
            if (active) dtab[ti] += ds;    // distinct ti across the wave's lanes for a fixed j -> no intra-instruction collision
            if (++bj == 7) { bj = 0; ++aj; }
        }
        if (qtok >= 0) {
            for (int kk = 0; kk < 7; ++kk) {
                axpy_glb(dq, dqr[kk * 64 + lane], rel_h + (a - kk + 6) * HD);
                axpy_glb(dq, dqr[(7 + kk) * 64 + lane], rel_w + (bb - kk + 6) * HD);
            }
            store_row(dqkv + ((int64_t)b * N + qtok) * ld + h * HD, dq);
        }
    }
    __syncthreads();
    // this key's sampled K/V back into registers (own LDS row), then republish: R1 <- Q | dO (padded queries = zeros)
    float ks[HD], vs[HD];
#pragma unroll
    for (int i = 0; i < HD / 4; ++i) {
        const float4 kk = active ? *reinterpret_cast<const float4*>(Ksel + lane * HD + 4 * i) : make_float4(0, 0, 0, 0);
        const float4 vv = active ? *reinterpret_cast<const float4*>(Vsel + lane * HD + 4 * i) : make_float4(0, 0, 0, 0);
        ks[4 * i] = kk.x; ks[4 * i + 1] = kk.y; ks[4 * i + 2] = kk.z; ks[4 * i + 3] = kk.w;
        vs[4 * i] = vv.x; vs[4 * i + 1] = vv.y; vs[4 * i + 2] = vv.z; vs[4 * i + 3] = vv.w;
    }
    __syncthreads();
    float* Qs = R1;
    float* dOs = R1 + 49 * HD;
    if (active) {
#pragma unroll
        for (int i = 0; i < HD / 4; ++i) {
            *reinterpret_cast<float4*>(Qs + lane * HD + 4 * i) = make_float4(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]);
            *reinterpret_cast<float4*>(dOs + lane * HD + 4 * i) = make_float4(dO[4 * i], dO[4 * i + 1], dO[4 * i + 2], dO[4 * i + 3]);
        }
    }
    __syncthreads();

    // ---- phase T (lane = channel): partial gradients of rel_pos_h / rel_pos_w and the bias table
    {
        float* rp = rel_part + (int64_t)blockIdx.x * 26 * HD;
        for (int r = 0; r < 13; ++r) {
            float ah = 0.f, aw = 0.f;
            int an = 0, bn = 0;
            for (int n = 0; n < 49; ++n) {
                const float qv = Qs[n * HD + lane];
                const int kh = an - r + 6, kw = bn - r + 6;
                if (kh >= 0 && kh < 7) ah += dqr[kh * 64 + n] * qv;
                if (kw >= 0 && kw < 7) aw += dqr[(7 + kw) * 64 + n] * qv;
                if (++bn == 7) { bn = 0; ++an; }
            }
            rp[r * HD + lane] = ah;
            rp[(13 + r) * HD + lane] = aw;
        }
        for (int i = lane; i < 169; i += 64) tab_part[((int64_t)bw * H + h) * 169 + i] = dtab[i];   // (window, head, 169): contiguous per workgroup
    }

    // ---- phase K (lane = key): d(K_sel), d(V_sel), then scatter through the bilinear footprint + coordinate gradients
    float dks[HD], dvs[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { dks[d] = 0.f; dvs[d] = 0.f; }
    {
        int an = 0, bn = 0;
        for (int n = 0; n < 49; ++n) {
            const float s = scale * dot_lds(ks, Qs + n * HD) + relq[a * 64 + n] + relq[(7 + bb) * 64 + n] + tab[(an - a + 6) * 13 + (bn - bb + 6)];
            const float p = active ? __expf(s - lses[n]) : 0.f;
            const float ds = p * (dot_lds(vs, dOs + n * HD) - delta[n]);
            axpy_lds(dks, ds * scale, Qs + n * HD);
            axpy_lds(dvs, p, dOs + n * HD);
            if (++bn == 7) { bn = 0; ++an; }
        }
    }
    float dix = 0.f, diy = 0.f;
    if (active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float w; bool in_map;
            const int tok = neighbour(g, smp, k, w, in_map);
            if (tok >= 0) {
                float* dkrow = dkv + ((int64_t)b * N + tok) * (2 * C) + h * HD;
                float t[HD];
                float dot = 0.f;
                load_row(base + C + (int64_t)tok * ld, t);
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    dot += dks[d] * t[d];
                    atomicAdd(dkrow + d, w * dks[d]);
                }
                load_row(base + 2 * C + (int64_t)tok * ld, t);
#pragma unroll
                for (int d = 0; d < HD; ++d) {
                    dot += dvs[d] * t[d];
                    atomicAdd(dkrow + C + d, w * dvs[d]);
                }
                const int dx = k & 1, dy = k >> 1;
                dix += dot * (dy ? smp.fy : 1.0f - smp.fy) * (dx ? 1.0f : -1.0f);
                diy += dot * (dx ? smp.fx : 1.0f - smp.fx) * (dy ? 1.0f : -1.0f);
            }
        }
    }
    // pixel -> normalised coords (align_corners=True), then the 5 sampling scalars of this (window, head)
    const float dgx = dix * 0.5f * (float)(g.We - 1), dgy = diy * 0.5f * (float)(g.He - 1);
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    if (active) {
        v0 = dgx * g.inv_div_x;
        v1 = dgy * g.inv_div_y;
        v2 = (dgx * smp.cs + dgy * smp.sn) * smp.relx;
        v3 = (-dgx * smp.sn + dgy * smp.cs) * smp.rely;
        v4 = dgx * (-smp.rx * smp.sn - smp.ry * smp.cs) + dgy * (-smp.ry * smp.sn + smp.rx * smp.cs);
    }
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2); v3 = wave_sum(v3); v4 = wave_sum(v4);
    if (lane == 0) {
        float* dp = dsamp + (int64_t)bw * 5 * H;
        dp[2 * h] = v0; dp[2 * h + 1] = v1; dp[2 * H + 2 * h] = v2; dp[2 * H + 2 * h + 1] = v3; dp[4 * H + h] = v4;
    }
}

// dqkv[t][C + c] = dkv[t][c]  (c < 2C)
template <typename T>
__global__ __launch_bounds__(256) void dkv_convert_kernel(const float* __restrict__ dkv, T* __restrict__ dqkv, int64_t Ttok, int C) {
    const int C2_4 = 2 * C / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < Ttok * C2_4; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C2_4);
        const int64_t t = i / C2_4;
        store4(dqkv + t * 3 * C + C + 4 * c4, load4(dkv + t * 2 * C + 4 * c4));
    }
}

RvsaGeom make_geom(int64_t Hp, int64_t Wp, int64_t heads) {
    RvsaGeom g;
    const int pad_h = (int)((7 - Hp % 7) % 7), pad_w = (int)((7 - Wp % 7) % 7);
    g.Hp = (int)Hp; g.Wp = (int)Wp;
    g.pad_t = pad_h / 2; g.pad_l = pad_w / 2;
    g.He = (int)Hp + pad_h; g.We = (int)Wp + pad_w;
    g.nh = g.He / 7; g.nw = g.We / 7;
    g.heads = (int)heads;
    g.inv_div_x = 1.0f / (float)(Hp / 7);   // VIT:359: x offset / (h // ws)
    g.inv_div_y = 1.0f / (float)(Wp / 7);   // VIT:360: y offset / (w // ws)
    return g;
}

}  // namespace

extern "C" int mtp_full_attn_fwd(const void* qkv, void* o, float* lse, int dtype, const float* rel_h, const float* rel_w,
                                 int64_t B, int64_t Hp, int64_t Wp, int64_t heads, int64_t hd, float scale, mtp_stream_t stream) {
    if (!qkv || !o || !lse || !rel_h || !rel_w || B <= 0 || Hp <= 0 || Wp <= 0 || heads <= 0) return MTP_ERR_ARG;
    if (hd != HD) return MTP_ERR_UNSUPPORTED;
    if (dtype == MTP_BF16 && mtp_use_mfma_attn()) {
        const int rc = mtp_full_fwd_mfma_launch(qkv, o, lse, rel_h, rel_w, B, Hp, Wp, heads, scale, (hipStream_t)stream);
        if (rc != MTP_ERR_UNSUPPORTED) return rc;   // larger token grids fall through to the generic kernel
    }
    const int N = (int)(Hp * Wp);
    const size_t lds = sizeof(float) * (size_t)(2 * KT * HD + (Hp + Wp) * 256 + KT * 256);
    if (lds > 160 * 1024) return MTP_ERR_UNSUPPORTED;
    dim3 grid((unsigned)(B * heads), (unsigned)((N + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16) {
        (void)hipFuncSetAttribute((const void*)full_attn_fwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((full_attn_fwd_kernel<bf16_t>), grid, block, lds, s, (const bf16_t*)qkv, (bf16_t*)o, lse, rel_h, rel_w, N, (int)Hp, (int)Wp, (int)heads, scale);
    } else {
        (void)hipFuncSetAttribute((const void*)full_attn_fwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((full_attn_fwd_kernel<float>), grid, block, lds, s, (const float*)qkv, (float*)o, lse, rel_h, rel_w, N, (int)Hp, (int)Wp, (int)heads, scale);
    }
    return mtp_launch_status();
}

extern "C" int64_t mtp_full_attn_bwd_workspace_floats(int64_t B, int64_t Hp, int64_t Wp, int64_t heads) {
    const int64_t N = Hp * Wp;
    return N > 256 ? B * heads * (2 * (Hp + Wp) + 1) * N : 0;
}

extern "C" int mtp_full_attn_bwd(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, int dtype,
                                 const float* rel_h, const float* rel_w, float* drel_part, float* workspace,
                                 int64_t B, int64_t Hp, int64_t Wp, int64_t heads, int64_t hd, float scale, mtp_stream_t stream) {
    if (!qkv || !o || !dout || !lse || !dqkv || !rel_h || !rel_w || !drel_part || B <= 0 || heads <= 0) return MTP_ERR_ARG;
    if (hd != HD) return MTP_ERR_UNSUPPORTED;
    if (dtype == MTP_BF16 && mtp_use_mfma_attn()) {
        int rc = mtp_full_bwd_mfma_launch(qkv, o, dout, lse, dqkv, rel_h, rel_w, drel_part, B, Hp, Wp, heads, scale, (hipStream_t)stream);
        if (rc != MTP_ERR_UNSUPPORTED) return rc;
        rc = mtp_full_bwd_flash_launch(qkv, o, dout, lse, dqkv, rel_h, rel_w, drel_part, workspace, B, Hp, Wp, heads, scale, (hipStream_t)stream);
        if (rc != MTP_ERR_UNSUPPORTED) return rc;
    }
    const int N = (int)(Hp * Wp);
    if (N > 256) {   // multi-workgroup three-pass backward (448^2 / 512^2 inputs); needs the caller's workspace
        if (!workspace) return MTP_ERR_ARG;
        const int HW = (int)(Hp + Wp), RT = (int)(2 * Hp - 1 + 2 * Wp - 1);
        const size_t lds_q = sizeof(float) * (size_t)(2 * HW * 256 + 2 * KT * HD);
        const size_t lds_k = sizeof(float) * (size_t)(2 * KT * HD + HW * KT + 2 * KT);
        if (lds_q > 160 * 1024) return MTP_ERR_UNSUPPORTED;
        hipStream_t s = (hipStream_t)stream;
        dim3 gq((unsigned)(B * heads), (unsigned)((N + 255) / 256)), gt((unsigned)(B * heads), (unsigned)((RT + 3) / 4)), block(256);
        if (dtype == MTP_BF16) {
            (void)hipFuncSetAttribute((const void*)full_attn_bwdN_q_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
            hipLaunchKernelGGL((full_attn_bwdN_q_kernel<bf16_t>), gq, block, lds_q, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse, (bf16_t*)dqkv,
                               rel_h, rel_w, workspace, N, (int)Hp, (int)Wp, (int)heads, scale);
            hipLaunchKernelGGL((full_attn_bwdN_t_kernel<bf16_t>), gt, block, 0, s, (const bf16_t*)qkv, (const float*)workspace, drel_part, N, (int)Hp, (int)Wp, (int)heads, scale);
            hipLaunchKernelGGL((full_attn_bwdN_k_kernel<bf16_t>), gq, block, lds_k, s, (const bf16_t*)qkv, (const bf16_t*)dout, lse, (bf16_t*)dqkv,
                               (const float*)workspace, N, (int)Hp, (int)Wp, (int)heads, scale);
        } else {
            (void)hipFuncSetAttribute((const void*)full_attn_bwdN_q_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);
            hipLaunchKernelGGL((full_attn_bwdN_q_kernel<float>), gq, block, lds_q, s, (const float*)qkv, (const float*)o, (const float*)dout, lse, (float*)dqkv,
                               rel_h, rel_w, workspace, N, (int)Hp, (int)Wp, (int)heads, scale);
            hipLaunchKernelGGL((full_attn_bwdN_t_kernel<float>), gt, block, 0, s, (const float*)qkv, (const float*)workspace, drel_part, N, (int)Hp, (int)Wp, (int)heads, scale);
            hipLaunchKernelGGL((full_attn_bwdN_k_kernel<float>), gq, block, lds_k, s, (const float*)qkv, (const float*)dout, lse, (float*)dqkv,
                               (const float*)workspace, N, (int)Hp, (int)Wp, (int)heads, scale);
        }
        return mtp_launch_status();
    }
    if ((2 * Hp - 1) + (2 * Wp - 1) > 4 * MAXR) return MTP_ERR_UNSUPPORTED;   // single-workgroup backward (224^2..256^2 inputs)
    const size_t lds = sizeof(float) * (size_t)(2 * (Hp + Wp) * 256 + 512 + 2 * KT * HD);
    if (lds > 160 * 1024) return MTP_ERR_UNSUPPORTED;
    dim3 grid((unsigned)(B * heads)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16) {
        (void)hipFuncSetAttribute((const void*)full_attn_bwd_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((full_attn_bwd_kernel<bf16_t>), grid, block, lds, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse, (bf16_t*)dqkv,
                           rel_h, rel_w, drel_part, N, (int)Hp, (int)Wp, (int)heads, scale);
    } else {
        (void)hipFuncSetAttribute((const void*)full_attn_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((full_attn_bwd_kernel<float>), grid, block, lds, s, (const float*)qkv, (const float*)o, (const float*)dout, lse, (float*)dqkv,
                           rel_h, rel_w, drel_part, N, (int)Hp, (int)Wp, (int)heads, scale);
    }
    return mtp_launch_status();
}

extern "C" int mtp_rvsa_attn_fwd(const void* qkv, const float* samp, void* o, float* lse, int dtype,
                                 const float* rel_h, const float* rel_w, const float* bias_table,
                                 int64_t B, int64_t Hp, int64_t Wp, int64_t heads, int64_t hd, float scale, mtp_stream_t stream) {
    if (!qkv || !samp || !o || !lse || !rel_h || !rel_w || !bias_table || B <= 0 || Hp < 7 || Wp < 7 || heads <= 0) return MTP_ERR_ARG;
    if (hd != HD) return MTP_ERR_UNSUPPORTED;
    const RvsaGeom g = make_geom(Hp, Wp, heads);
    dim3 grid((unsigned)(B * g.nh * g.nw * heads)), block(64);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16 && mtp_use_mfma_attn())
        return mtp_rvsa_fwd_mfma_launch(qkv, samp, o, lse, rel_h, rel_w, bias_table, B, Hp, Wp, heads, scale, s);
    if (dtype == MTP_BF16)
        hipLaunchKernelGGL((rvsa_attn_fwd_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)qkv, samp, (bf16_t*)o, lse, rel_h, rel_w, bias_table, g, scale);
    else
        hipLaunchKernelGGL((rvsa_attn_fwd_kernel<float>), grid, block, 0, s, (const float*)qkv, samp, (float*)o, lse, rel_h, rel_w, bias_table, g, scale);
    return mtp_launch_status();
}

extern "C" int mtp_rvsa_attn_bwd(const void* qkv, const float* samp, const void* o, const void* dout, const float* lse,
                                 void* dqkv, float* dkv_f32, float* dsamp, float* rel_part, float* tab_part, int dtype,
                                 const float* rel_h, const float* rel_w, const float* bias_table,
                                 int64_t B, int64_t Hp, int64_t Wp, int64_t heads, int64_t hd, float scale, mtp_stream_t stream) {
    if (!qkv || !samp || !o || !dout || !lse || !dqkv || !dkv_f32 || !dsamp || !rel_part || !tab_part || !rel_h || !rel_w || !bias_table) return MTP_ERR_ARG;
    if (B <= 0 || Hp < 7 || Wp < 7 || heads <= 0) return MTP_ERR_ARG;
    if (hd != HD) return MTP_ERR_UNSUPPORTED;
    const RvsaGeom g = make_geom(Hp, Wp, heads);
    const int64_t Ttok = B * Hp * Wp, C = heads * HD;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_BF16 && mtp_use_mfma_attn() && mtp_rvsa_bwd_mfma_scatter_mode(Hp, Wp, heads) == 4)     // dense-product scatter: dqkv's k / v
        return mtp_rvsa_bwd_mfma_launch(qkv, samp, o, dout, lse, dqkv, dkv_f32, dsamp, rel_part, tab_part, rel_h, rel_w, bias_table, B, Hp, Wp, heads, scale, s);   // parts are written once, no scratch passes
    hipError_t e = hipMemsetAsync(dkv_f32, 0, sizeof(float) * (size_t)(Ttok * 2 * C), s);
    if (e != hipSuccess) return (int)e;
    // every real token is the query of exactly one window, so the q part of dqkv is fully written by the kernel
    dim3 grid((unsigned)(B * g.nh * g.nw * heads)), block(64);
    int64_t cb = (Ttok * 2 * C / 4 + 255) / 256;
    dim3 cgrid((unsigned)(cb > 8192 ? 8192 : cb)), cblock(256);
    if (dtype == MTP_BF16 && mtp_use_mfma_attn()) {
        const int rc = mtp_rvsa_bwd_mfma_launch(qkv, samp, o, dout, lse, dqkv, dkv_f32, dsamp, rel_part, tab_part, rel_h, rel_w, bias_table, B, Hp, Wp, heads, scale, s);
        if (rc) return rc;
        hipLaunchKernelGGL((dkv_convert_kernel<bf16_t>), cgrid, cblock, 0, s, dkv_f32, (bf16_t*)dqkv, Ttok, (int)C);
    } else if (dtype == MTP_BF16) {
        hipLaunchKernelGGL((rvsa_attn_bwd_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)qkv, samp, (const bf16_t*)o, (const bf16_t*)dout, lse, (bf16_t*)dqkv,
                           dkv_f32, dsamp, rel_part, tab_part, rel_h, rel_w, bias_table, g, scale);
        hipLaunchKernelGGL((dkv_convert_kernel<bf16_t>), cgrid, cblock, 0, s, dkv_f32, (bf16_t*)dqkv, Ttok, (int)C);
    } else {
        hipLaunchKernelGGL((rvsa_attn_bwd_kernel<float>), grid, block, 0, s, (const float*)qkv, samp, (const float*)o, (const float*)dout, lse, (float*)dqkv,
                           dkv_f32, dsamp, rel_part, tab_part, rel_h, rel_w, bias_table, g, scale);
        hipLaunchKernelGGL((dkv_convert_kernel<float>), cgrid, cblock, 0, s, dkv_f32, (float*)dqkv, Ttok, (int)C);
    }
    return mtp_launch_status();
}
