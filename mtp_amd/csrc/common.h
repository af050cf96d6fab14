// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of the MTP ViT+RVSA backbone path.
// Written for gfx950 only: 64-wide wavefronts, MFMA 16x16x32 bf16 / 16x16x4 f32, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mtp_hip.h"

#define MTP_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct bf16_t {  // storage-only 16-bit brain float
    uint16_t bits;
};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {  // round-to-nearest-even, NaN preserved
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// one v_cvt_pk_bf16_f32 (round-to-nearest-even) for two values
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint4 pack_bf16x8(float a, float b, float c, float d, float e, float f, float g, float h) {
    return make_uint4(pack_bf16x2(a, b), pack_bf16x2(c, d), pack_bf16x2(e, f), pack_bf16x2(g, h));
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
    static constexpr int kPerChunk = 4;  // elements per 16-byte chunk
    static constexpr int kDtype = MTP_F32;
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_t> {
    static constexpr int kPerChunk = 8;
    static constexpr int kDtype = MTP_BF16;
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_bits_to_f32(p->bits); }
    __device__ static __forceinline__ void store(bf16_t* p, float v) { p->bits = (uint16_t)f32_to_bf16_bits(v); }
};

// 4 consecutive elements <-> float4 (vectorised: 16 B for f32, 8 B for bf16)
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    return make_float4(bf16_bits_to_f32(v.x & 0xffffu), bf16_bits_to_f32(v.x >> 16), bf16_bits_to_f32(v.y & 0xffffu), bf16_bits_to_f32(v.y >> 16));
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, float4 v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}
// 8 consecutive elements (32 B f32 / 16 B bf16)
__device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
    float4 a = load4(p), b = load4(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&o)[8]) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    o[0] = bf16_bits_to_f32(v.x & 0xffffu); o[1] = bf16_bits_to_f32(v.x >> 16);
    o[2] = bf16_bits_to_f32(v.y & 0xffffu); o[3] = bf16_bits_to_f32(v.y >> 16);
    o[4] = bf16_bits_to_f32(v.z & 0xffffu); o[5] = bf16_bits_to_f32(v.z >> 16);
    o[6] = bf16_bits_to_f32(v.w & 0xffffu); o[7] = bf16_bits_to_f32(v.w >> 16);
}
__device__ __forceinline__ void store8(float* p, const float (&o)[8]) {
    store4(p, make_float4(o[0], o[1], o[2], o[3]));
    store4(p + 4, make_float4(o[4], o[5], o[6], o[7]));
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&o)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
}

// 16-byte load that is guaranteed to be a global_load (not flat_load): needed when the pointer comes out of a select
// (a flat load also counts on lgkmcnt and forces a vmcnt(0) before the next LDS access)
__device__ __forceinline__ uint4 ldg16(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const u32x4_t v = *(const __attribute__((address_space(1))) u32x4_t*)(uintptr_t)p;
    return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}

// 16-byte global load at a WAVE-UNIFORM base plus a 32-bit per-lane byte offset: the compiler selects `global_load_dwordx4 v, v_off, s[base]`,
// so the per-lane address costs one 32-bit multiply-add instead of a 64-bit one (v_mad_u64_u32 + v_lshl_add_u64) -- for gathers of token rows out of
// buffers smaller than 4 GiB (every activation of the path)
__device__ __forceinline__ uint4 ldg16_at(const void* base, uint32_t byte_off) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const u32x4_t v = *(const __attribute__((address_space(1))) u32x4_t*)((uintptr_t)base + (uintptr_t)byte_off);
    return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + byte_off);
#endif
}

// exact-erf GELU (nn.GELU default) and its derivative
// erf via Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. f32-roundoff class like erff itself) -- 1 rcp + 1 exp + 5 fma
// instead of ocml's ~35-instruction erff: the GELU / GELU' GEMM epilogues were spending ~65 us per fc1-sized launch in erff.
// `e` returns exp(-z^2) so GELU' can reuse it (exp(-x^2/2) with z = x/sqrt(2)).
__device__ __forceinline__ float erf_as(float z, float& e) {
    const float az = fabsf(z);
    // v_rcp_f32 (1 ulp) and v_exp_f32 directly: `__frcp_rn` expands to the full IEEE division sequence (2 x v_div_scale,
    // v_div_fmas, v_div_fixup, 4 fma) -- 28 -> 18 VALU instructions per GELU, and the fc1 / GELU' epilogues are VALU-bound
    // (51 M elements per launch); the approximation's own error is 1.5e-7, far above 1 ulp of t
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * az);
    const float y = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    e = __builtin_amdgcn_exp2f(-1.4426950408889634f * az * az);
    return copysignf(1.0f - y * e, z);
}
__device__ __forceinline__ float gelu_f(float x) {
    float e;
    return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f, e));
}
// gelu(x) and gelu'(x) together: one erf, one exponential
__device__ __forceinline__ void gelu_pair_f(float x, float& g, float& dg) {
    float e;
    const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f, e));
    g = x * cdf;
    dg = cdf + x * 0.39894228040143267794f * e;
}
__device__ __forceinline__ float dgelu_f(float x) {
    float e;
    const float er = erf_as(x * 0.70710678118654752440f, e);
    return 0.5f * (1.0f + er) + x * 0.39894228040143267794f * e;
}

// ---- lane exchanges without the LDS crossbar (round 5; probed in round 4: tools/probes/dpp_probe.hip, profiles/r04_dpp_probe.txt) --------------------
// hipcc turns every __shfl_xor into ds_bpermute_b32: an LDS instruction and an LDS round trip on the dependent chain (312 clocks per dependent 16-lane
// butterfly against 88 with DPP).  The partners lane ^ 1 / 2 / 4 / 8 sit inside a 16-lane row and are reachable with DPP modifiers (xor 1, 2, 3 =
// quad_perm; xor 7 = row_half_mirror; xor 15 = row_mirror; xor 4 = 7 o 3, xor 8 = 15 o 7); lane ^ 16 and lane ^ 32 cross rows: v_permlane16_swap /
// v_permlane32_swap of a register with itself leave the pair (even rows | odd rows) resp. (lower half | upper half) replicated in both results, so the
// SUM / MAX with the partner is one swap and one add / max -- the same two operands as v + v[lane ^ 16], hence the same bits.
// ONLY where every lane of the wave is active (a DPP / permlane read of a disabled lane does not return that lane's register as ds_bpermute does):
// every call site is straight-line code behind wave-uniform control flow (audit: tools/ablation/README.md, round 4).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int M>
__device__ __forceinline__ float lane_xor(float v) {   // v[lane ^ M], M = 1, 2, 4, 8
    static_assert(M == 1 || M == 2 || M == 4 || M == 8, "in-row partners only");
    if constexpr (M == 1) return dpp_mov<0xB1>(v);
    else if constexpr (M == 2) return dpp_mov<0x4E>(v);
    else if constexpr (M == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));
    else return dpp_mov<0x141>(dpp_mov<0x140>(v));
}
__device__ __forceinline__ float xor16_sum(float v) {   // v + v[lane ^ 16]
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {   // v + v[lane ^ 32]
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// over the four lanes that share (lane & 15): the MFMA layout's "same row, the four 4-column groups" reductions of the attention kernels
__device__ __forceinline__ float rows_sum(float v) { return xor32_sum(xor16_sum(v)); }
__device__ __forceinline__ float rows_max(float v) { return xor32_max(xor16_max(v)); }
// wave-wide reductions over 64 lanes (same order of additions as the __shfl_xor butterflies 32, 16, 8, 4, 2, 1 they replace)
__device__ __forceinline__ float wave_sum(float v) {
    v = xor32_sum(v); v = xor16_sum(v);
    v += lane_xor<8>(v); v += lane_xor<4>(v); v += lane_xor<2>(v); v += lane_xor<1>(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = xor32_max(v); v = xor16_max(v);
    v = fmaxf(v, lane_xor<8>(v)); v = fmaxf(v, lane_xor<4>(v)); v = fmaxf(v, lane_xor<2>(v)); v = fmaxf(v, lane_xor<1>(v));
    return v;
}

// opt-in for more than 64 KiB of dynamic LDS, once per (kernel, device): `mask` is the call site's static bit set of devices already done
// (a process-wide bool would skip a second device of the same process; setting the attribute twice in a race is harmless)
static inline int mtp_optin_lds(const void* kern, int bytes, unsigned long long& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 64;
    if (dev < 64 && ((mask >> dev) & 1ull)) return 0;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    if (dev < 64) mask |= 1ull << dev;
    return 0;
}

#define MTP_CHECK_ARG(cond) \
    do {                    \
        if (!(cond)) return MTP_ERR_ARG; \
    } while (0)

// CUs a stream may use: what mtp_stream_create_cu_mask registered for it (the two half-batch schedule), else every CU of the device.  The GEMM
// dispatch sizes tiles, strips and persistent grids by it (a 128-CU stream runs M / 2 rows with the tile quantisation of M rows on 256 CUs).
int mtp_stream_cus(hipStream_t stream);

static inline int mtp_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
