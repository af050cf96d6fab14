// helpers shared by the full-attention MFMA kernels (attn_full_mfma.hip: <= 256 tokens and the flash forward;
// attn_full_flash_bwd.hip: the flash backward beyond 256 tokens).  Internal linkage: every translation unit gets its own copy.
#pragma once
#include "common.h"

namespace {

constexpr int HD = 64;

__device__ __attribute__((aligned(16))) const uint4 g_zero16f = {0u, 0u, 0u, 0u};

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint4 ld16(const char* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 ld8x2(const char* p0, const char* p1) {
    const uint2 a = *reinterpret_cast<const uint2*>(p0), b = *reinterpret_cast<const uint2*>(p1);
    return make_uint4(a.x, a.y, b.x, b.y);
}
__device__ __forceinline__ uint4 row_frag(const bf16_t* __restrict__ rows, int64_t ld, int tok, bool ok, int e0) {
    return ldg16(ok ? reinterpret_cast<const char*>(rows + (int64_t)tok * ld + e0) : reinterpret_cast<const char*>(&g_zero16f));
}
__device__ __forceinline__ uint4 table_frag(const float* __restrict__ tab, int r, int rows, int e0) {
    if (r >= rows) return make_uint4(0, 0, 0, 0);
    const float4 a = *reinterpret_cast<const float4*>(tab + r * HD + e0), b = *reinterpret_cast<const float4*>(tab + r * HD + e0 + 4);
    return pack_bf16x8(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
__device__ __forceinline__ uint4 table_frag_t(const float* __restrict__ tab, int d, int rows, int r0) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (r0 + e) < rows ? tab[(r0 + e) * HD + d] : 0.f;
    return pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}

struct FGeom {
    int N, Hp, Wp, heads, NT, NP, NP2, KK, TPV, RH, RW;
};

// K (or any 64-wide row block of qkv) -> swizzled row-major LDS image, rows >= N zeroed, up to `rows` rows
__device__ __forceinline__ void stage_rows_swz(const bf16_t* __restrict__ src, int64_t ld, int N, int rows, char* img, int tid, int nthreads = 256) {
    for (int idx = tid; idx < rows * 8; idx += nthreads) {
        const int row = idx >> 3, c = idx & 7;
        *reinterpret_cast<uint4*>(img + swz(row, c)) = row_frag(src, ld, row, row < N, 8 * c);
    }
}
// K^T fragment (MFMA A operand: row d = 16 dt + fr, k = keys key0 .. key0+3 and key0+16 .. key0+19) out of the swizzled row-major K image:
// in each 16-lane group, lane i supplies the address of row i >> 2, columns 4 (i & 3) .. +3 of a [4 keys][16 d] block and receives
// column i of its four rows
typedef short tr4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 kt_frag_tr(const char* Ks, int key0, int dt, int fr) {
    const int c = 16 * dt + 4 * (fr & 3);
    const int ra = key0 + (fr >> 2), rb = ra + 16;
    const int oa = ra * 128 + ((((c >> 3) ^ (ra & 7))) << 4) + (c & 7) * 2, ob = rb * 128 + ((((c >> 3) ^ (rb & 7))) << 4) + (c & 7) * 2;
    const tr4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4_t*)(Ks + oa));
    const tr4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4_t*)(Ks + ob));
    const uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, h.x, h.y);
}

// the same in two steps, for a software-pipelined block loop: the global loads of the NEXT block are issued into registers right behind the barrier that
// publishes the current one (prefetch_rows) and written to LDS at the top of the next trip (commit_rows) -- their latency runs under the block's MFMAs
template <int ROWS>
__device__ __forceinline__ void prefetch_rows(const bf16_t* __restrict__ src, int64_t ld, int N, int tid, uint4 (&r)[ROWS * 8 / 256]) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / 256; ++i) {
        const int idx = tid + 256 * i, row = idx >> 3, c = idx & 7;
        r[i] = row_frag(src, ld, row, row < N, 8 * c);
    }
}
template <int ROWS>
__device__ __forceinline__ void commit_rows(char* img, int tid, const uint4 (&r)[ROWS * 8 / 256]) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / 256; ++i) {
        const int idx = tid + 256 * i, row = idx >> 3, c = idx & 7;
        *reinterpret_cast<uint4*>(img + swz(row, c)) = r[i];
    }
}
// 64-wide rows -> transposed image img[d][row] (pitch TPV bytes), columns >= N zeroed, up to `cols` columns
__device__ __forceinline__ void stage_rows_t(const bf16_t* __restrict__ src, int64_t ld, int N, int cols, int TPV, char* img, int tid, int nthreads = 256) {
    for (int idx = tid; idx < cols * 8; idx += nthreads) {
        const int row = idx >> 3, c = idx & 7;
        const uint4 v = row_frag(src, ld, row, row < N, 8 * c);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<uint16_t*>(img + (8 * c + 2 * e) * TPV + row * 2) = (uint16_t)(w[e] & 0xffffu);
            *reinterpret_cast<uint16_t*>(img + (8 * c + 2 * e + 1) * TPV + row * 2) = (uint16_t)(w[e] >> 16);
        }
    }
}

}  // namespace
