// Pieces shared by the GEMM translation units (gemm.hip: 128-wide tiles, both precisions; gemm_p8.hip: the 256 x 256
// 8-wave bf16 NT kernel): MFMA wrappers, kernel argument block, XCD-aware tile remap, fused epilogues.
#pragma once
#include "common.h"

struct KArgs {
    const char* A;
    const char* B;
    char* C;
    int M, N, K;
    int64_t lda, ldb, ldc;  // in elements
    const float* bias;
    int bias_mod;
    const float* res;
    int64_t res_ld;
    int res_mod;
    const float* rowscale;
    int rows_per_sample;
    char* aux;
    int64_t aux_ld;
    int tiles_n;
    int k_tiles;          // total k tiles
    int k_tiles_per_split;
    int atomic_out;
    int order;              // tile order experiment: bit0 = no XCD remap, bit1 = M-fastest instead of N-fastest
    int64_t split_stride;   // TN split-K with workspace: partial tile of split z lives at C + z*split_stride (f32 elements)
    float* colsum;          // TN (transpose-read kernel): colsum[m] += sum_k A[k][m], or nullptr
};

// gemm_p8.hip: 256 x 256 x 64 tile, 8 waves, 8-phase LDS-DMA pipeline (bf16 NT, complete K tiles).  flags: bit0 = 224-row tiles,
// bit2 = 256-row tiles (neither: picked per problem), bit1 = plain tile order (no XCD remap / grouping), bits 8 / 9 = force / forbid persistent tiles, bits 13-14 = store policy of the epilogue.  Returns MTP_ERR_UNSUPPORTED when the
// problem does not fit the kernel's preconditions (the caller then uses the 128-wide kernels of gemm.hip).
int mtp_nt_p8_launch(const KArgs& k, int out_dtype, int epilogue, int flags, hipStream_t stream);
int mtp_nt_p8_fits(const KArgs& k, int out_dtype, int epilogue);   // 1 when mtp_nt_p8_launch would run the problem
// gemm_s8.hip: 128 x 256 x 64 strips, 8 waves, two accumulator sets: the epilogue of a strip runs under the next strip's K loop
// (bf16 NT, K >= 704 in whole K-tiles).  flags: bit1 = plain strip order.  MTP_ERR_UNSUPPORTED when the problem does not fit.
int mtp_nt_s8_launch(const KArgs& k, int out_dtype, int epilogue, int flags, hipStream_t stream);
int mtp_nt_s8_fits(const KArgs& k, int out_dtype, int epilogue);

namespace {

template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <>
struct Mma<float> {
    __device__ static __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// 16 zero bytes in HBM: out-of-range tile elements load from here (address select BEFORE the load keeps it branch-free)
__device__ __attribute__((aligned(16))) const uint4 g_zero16 = {0u, 0u, 0u, 0u};

// bijective XCD-aware remap: hardware places block b on XCD b % 8; give each XCD a contiguous range of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    int xcd = bid & 7, idx = bid >> 3, q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


// ---- epilogue: the wave's accumulators acc[ni][mi] (NF x MF MFMA tiles); lane holds
//      C[mrow0 + mi*16 + (lane&15)][ncol0 + ni*16 + 4*(lane>>4) + 0..3]  (mrow0 / ncol0 = the wave's origin in C)
template <typename Tout, int EPI, int MF>
__device__ __forceinline__ void epilogue(const KArgs& p, f32x4_t (&acc)[4][MF], int mrow0, int ncol0, int lane) {
    const int fr = lane & 15, g = lane >> 4;
    Tout* C = reinterpret_cast<Tout*>(p.C);
    // All epilogue LOADS are unconditional on clamped addresses (a branch around a load makes hipcc wait for it inside
    // the branch: 16 serialised HBM latencies per tile); only the stores are predicated.
    int nn[4];
    bool nok[4];
    float4 bias[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = ncol0 + ni * 16 + g * 4;
        nok[ni] = n < p.N;
        nn[ni] = nok[ni] ? n : 0;
        const float* bp = (EPI != MTP_EPI_DGELU && EPI != MTP_EPI_MUL && p.bias) ? p.bias + (p.bias_mod > 0 ? nn[ni] % p.bias_mod : nn[ni]) : reinterpret_cast<const float*>(&g_zero16);
        const uint4 b = ldg16(bp);
        bias[ni] = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
    }
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
        const int m = mrow0 + mi * 16 + fr;
        const bool mok = m < p.M;
        const int mc = mok ? m : 0;
        if (p.atomic_out) {   // split-K weight gradient: f32 atomics into a zeroed buffer
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                if (mok && nok[ni]) {
                    float* c = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + nn[ni];
                    atomicAdd(c + 0, acc[ni][mi][0]); atomicAdd(c + 1, acc[ni][mi][1]); atomicAdd(c + 2, acc[ni][mi][2]); atomicAdd(c + 3, acc[ni][mi][3]);
                }
            continue;
        }
        float rs = 1.0f;
        float4 side[4];
        if (EPI == MTP_EPI_BIAS_RES) {
            const float* rsp = p.rowscale ? p.rowscale + mc / p.rows_per_sample : nullptr;
            const float* resrow = p.res + (int64_t)(p.res_mod > 0 ? mc % p.res_mod : mc) * p.res_ld;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const uint4 r = ldg16(resrow + nn[ni]);
                side[ni] = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            }
            if (rsp) rs = *rsp;
        } else if (EPI == MTP_EPI_DGELU || EPI == MTP_EPI_MUL) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) side[ni] = load4(reinterpret_cast<const Tout*>(p.aux) + (int64_t)mc * p.aux_ld + nn[ni]);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            float4 v = make_float4(acc[ni][mi][0] + bias[ni].x, acc[ni][mi][1] + bias[ni].y, acc[ni][mi][2] + bias[ni].z, acc[ni][mi][3] + bias[ni].w);
            const bool ok = mok && nok[ni];
            if (EPI == MTP_EPI_BIAS_GELU) {
                if (ok) store4(reinterpret_cast<Tout*>(p.aux) + (int64_t)m * p.aux_ld + nn[ni], v);
                v = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
            } else if (EPI == MTP_EPI_BIAS_GELU_DG) {
                float4 d;
                gelu_pair_f(v.x, v.x, d.x); gelu_pair_f(v.y, v.y, d.y); gelu_pair_f(v.z, v.z, d.z); gelu_pair_f(v.w, v.w, d.w);
                if (ok) store4(reinterpret_cast<Tout*>(p.aux) + (int64_t)m * p.aux_ld + nn[ni], d);
            } else if (EPI == MTP_EPI_DGELU) {
                v = make_float4(v.x * dgelu_f(side[ni].x), v.y * dgelu_f(side[ni].y), v.z * dgelu_f(side[ni].z), v.w * dgelu_f(side[ni].w));
            } else if (EPI == MTP_EPI_MUL) {
                v = make_float4(v.x * side[ni].x, v.y * side[ni].y, v.z * side[ni].z, v.w * side[ni].w);
            } else if (EPI == MTP_EPI_BIAS_RES) {
                v = make_float4(side[ni].x + rs * v.x, side[ni].y + rs * v.y, side[ni].z + rs * v.z, side[ni].w + rs * v.w);
            }
            if (ok) store4(C + (int64_t)m * p.ldc + nn[ni], v);
        }
    }
}

}  // namespace
