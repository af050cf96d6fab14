// The 8-phase LDS-DMA pipeline shared by the 256 x 256 x 64 GEMM kernels (gemm_p8.hip: NT, activations x weights^T;
// gemm_tn_p8.hip: TN, grouped weight gradients).  The schedule (which half tile is read / issued / multiplied in which phase,
// the counted waits, the stagger between the two wave groups) lives HERE, once; an operand policy `OP` supplies how a half
// tile gets from global memory into LDS (DMA source addressing) and from LDS into MFMA fragments (ds_read_b128 of a
// K-contiguous image, or ds_read_b64_tr_b16 of an untransposed one).  Design notes: head of gemm_p8.hip.
#pragma once
#include "gemm_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

constexpr int P8_BM = 256, P8_BN = 256, P8_THREADS = 512;
constexpr int P8_HALF = 16384;          // one half tile: 128 rows x 128 B
constexpr int P8_BUF = 4 * P8_HALF;     // one K-tile buffer
constexpr int P8_LDS = 2 * P8_BUF;      // 128 KiB
enum : int { KA0 = 0, KB0 = 1, KB1 = 2, KA1 = 3, KNONE = -1 };   // slot of a half tile inside its buffer

// two LDS-DMA pieces (1 KiB each per wave): LDS[m0 + 16*lane] <- global[base + voff]
__device__ __forceinline__ void glds2(uint32_t voff, const char* p0, const char* p1, uint32_t l0, uint32_t l1) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(l0), "s"(p0), "s"(l1), "s"(p1)
        : "memory");
}

__device__ __forceinline__ void mma(f32x4_t& acc, const u32x4_t& w, const u32x4_t& x) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), acc, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}


// one phase: read half tile RK of buffer RBUF, issue half tile SK into buffer SBUF, keep at most VM loads in flight,
// then the 16 MFMAs of output quadrant (QI, QJ) with B register set QJ
// XP = ablation switches for tools/ab_gemm.py (0 in the product): bit0 no s_setprio, bit1 no stagger between the wave groups,
// bit2 epilogue stores nothing, bit3 no MFMAs (the DMA / ds_read stream alone), bit4 epilogue straight from the MFMA layout
template <class OP, int RK, int RBUF, int SK, int SBUF, int VM, int QI, int QJ, int XP>
__device__ __forceinline__ void phase(typename OP::Ctx& c, u32x4_t (&a)[2][4], u32x4_t (&b0)[2][2], u32x4_t (&b1)[2][2], f32x4_t (&acc)[4][8]) {
    if constexpr (RK == KA0 || RK == KA1) OP::template read_a<RK, RBUF>(c, a);
    if constexpr (RK == KB0) OP::template read_b<RK, RBUF>(c, b0);
    if constexpr (RK == KB1) OP::template read_b<RK, RBUF>(c, b1);
    if constexpr (SK != KNONE) OP::template stage<SK, SBUF>(c);
    wait_vm<VM * OP::kLoadsPerPiecePair / 2>();
    if constexpr (RK == KA0 || RK == KA1) OP::template retire_a<RK, RBUF>(c, a);
    if constexpr (RK == KB0) OP::retire_b(b0);
    if constexpr (RK == KB1) OP::retire_b(b1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(XP & 1)) __builtin_amdgcn_s_setprio(1);
    u32x4_t(&b)[2][2] = QJ ? b1 : b0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mi = 0; mi < (QI ? OP::kMi1 : 4); ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if constexpr (XP & 8)
                    asm volatile("" ::"v"(b[ks][ni]), "v"(a[ks][mi]));
                else
                    mma(acc[QJ * 2 + ni][QI * 4 + mi], b[ks][ni], a[ks][mi]);
            }
    if constexpr (!(XP & 1)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// 8 phases = K-tiles e (even, buffer 0) and e+1 (odd, buffer 1); the DMA issued here belongs to tiles e+2 / e+3
template <class OP, bool LAST, int XP>
__device__ __forceinline__ void two_tiles(typename OP::Ctx& c, u32x4_t (&a)[2][4], u32x4_t (&b0)[2][2], u32x4_t (&b1)[2][2], f32x4_t (&acc)[4][8]) {
    if constexpr (!LAST) {
        phase<OP, KA0, 0, KB1, 0, 12, 0, 1, XP>(c, a, b0, b1, acc);
        phase<OP, KB0, 0, KA0, 0, 12, 0, 0, XP>(c, a, b0, b1, acc);
        phase<OP, KA1, 0, KB0, 0, 12, 1, 0, XP>(c, a, b0, b1, acc);
        phase<OP, KB0, 1, KA1, 0, 12, 1, 1, XP>(c, a, b0, b1, acc);
        OP::next_ktile(c);
        phase<OP, KA0, 1, KB0, 1, 12, 0, 0, XP>(c, a, b0, b1, acc);
        phase<OP, KB1, 1, KA0, 1, 12, 0, 1, XP>(c, a, b0, b1, acc);
        phase<OP, KA1, 1, KB1, 1, 12, 1, 1, XP>(c, a, b0, b1, acc);
        phase<OP, KB1, 0, KA1, 1, 12, 1, 0, XP>(c, a, b0, b1, acc);
        OP::next_ktile(c);
    } else {   // everything has been issued: S_{g+3} .. S_{4nk-1} may stay in flight = 5 - q half tiles in phase q
        phase<OP, KA0, 0, KNONE, 0, 10, 0, 1, XP>(c, a, b0, b1, acc);
        phase<OP, KB0, 0, KNONE, 0, 8, 0, 0, XP>(c, a, b0, b1, acc);
        phase<OP, KA1, 0, KNONE, 0, 6, 1, 0, XP>(c, a, b0, b1, acc);
        phase<OP, KB0, 1, KNONE, 0, 4, 1, 1, XP>(c, a, b0, b1, acc);
        phase<OP, KA0, 1, KNONE, 0, 2, 0, 0, XP>(c, a, b0, b1, acc);
        phase<OP, KB1, 1, KNONE, 0, 0, 0, 1, XP>(c, a, b0, b1, acc);
        phase<OP, KA1, 1, KNONE, 0, 0, 1, 1, XP>(c, a, b0, b1, acc);
        phase<OP, KNONE, 0, KNONE, 0, 0, 1, 0, XP>(c, a, b0, b1, acc);
    }
}

// ---- read-ahead form of the phase (round 5) ---------------------------------------------------------------------------------
// Measured (profiles/r05_ab_p8_loop_ablations.txt): the loop is not DMA-bound (4 instead of 6 half tiles in flight: same time) -- what a
// wave's MFMAs wait for is its OWN fragment reads: in `phase` above the ds_reads of a half tile are issued at the start of the read part
// and retired (lgkmcnt(0)) before the barrier that starts the 16 MFMAs consuming them, so their issue time plus the LDS round trip sit
// on the wave's critical path (NT: 8 / 4 reads per phase, the read part ~1.15 x the MFMA part in the A phases; TN: 16 / 8 transpose
// reads, the read part ~2 x the MFMA part -- the "LDS-read-issue bound" 0.50 of the weight-gradient kernel).  Used by gemm_tn_p8.hip.
// Here the fragments of phase g + 1 are read INSIDE the MFMA block of phase g: every register is overwritten right after the last
// MFMA that uses it has been issued (a[ks][mi] after its two MFMAs; the b set in use after the last row of its k-step; an idle b set
// spread over the first groups), so the LDS instructions issue in the shadow of the matrix pipe and their latency runs under the rest of
// the block, the closing barrier and the next read part, which shrinks to { DMA issue, counted wait, lgkmcnt(0), barrier }.
// Hazards: the half tile read in the MFMA block of phase g is S_{g+2}; under the stagger group 0 multiplies in the interval in which
// group 1 runs R_g, so S_{g+2} must have been covered by the waits of R_{g-1}: every counted wait is ONE HALF TILE stricter
// (vmcnt(10): S_{g+3} .. S_{g+8} in flight).  The reads still retire in R_{g+1} (before its barrier), the slot is refilled in R_{g+2}: the
// WAR margin of the original schedule.  An LDS return lands >= 64 cycles after its issue, long after the preceding MFMAs have read
// their operands.
// (Measured on the grouped TN kernel, profiles/r05_ab_wgrad_readahead.txt: 5-11 % ahead of the plain phases.  A SPLIT form -- only the
// k-step-0 fragments under the MFMAs, the k-step-1 fragments at the start of the next read part -- was built too: bit-identical, 2-4 %
// behind this one.  On the NT kernel, whose phases read 8 / 4 b128 fragments, read-ahead changes nothing: -1 ... +3 %.)
template <class OP, int NRK, int NRBUF, int QJ, int KS, int MI, int MIC>
__device__ __forceinline__ void ra_reads(typename OP::Ctx& c, u32x4_t (&a)[2][4], u32x4_t (&b0)[2][2], u32x4_t (&b1)[2][2]) {
    if constexpr (NRK == KA0 || NRK == KA1) {
        constexpr int NMI = (NRK == KA1) ? OP::kMi1 : 4;       // fragments per k-step of the half tile being read
        if constexpr (MI < NMI) OP::template read_a_frag<NRK, NRBUF, KS, MI>(c, a);
        if constexpr (MI == MIC - 1 && MIC < NMI) OP::template read_a_frag<NRK, NRBUF, KS, (MIC < 4 ? MIC : 3)>(c, a);   // a row the current block does not multiply
    } else if constexpr (NRK == KB0 || NRK == KB1) {
        constexpr bool SAME = (NRK == KB0) == (QJ == 0);       // the set the current MFMAs use
        u32x4_t(&nb)[2][2] = NRK == KB0 ? b0 : b1;
        if constexpr (SAME) {
            if constexpr (MI == MIC - 1) {
                OP::template read_b_frag<NRK, NRBUF, KS, 0>(c, nb);
                OP::template read_b_frag<NRK, NRBUF, KS, 1>(c, nb);
            }
        } else {
            constexpr int G = KS * MIC + MI;                   // idle set: one fragment after each of the first four groups
            if constexpr (G < 4) OP::template read_b_frag<NRK, NRBUF, (G >> 1), (G & 1)>(c, nb);
        }
    }
}
template <class OP, int QI, int QJ, int NRK, int NRBUF, int KS, int MI, int MIC>
__device__ __forceinline__ void ra_group(typename OP::Ctx& c, u32x4_t (&a)[2][4], u32x4_t (&b0)[2][2], u32x4_t (&b1)[2][2], f32x4_t (&acc)[4][8]) {
    if constexpr (MI < MIC) {
        u32x4_t(&b)[2][2] = QJ ? b1 : b0;
        mma(acc[QJ * 2 + 0][QI * 4 + MI], b[KS][0], a[KS][MI]);
        mma(acc[QJ * 2 + 1][QI * 4 + MI], b[KS][1], a[KS][MI]);
        if constexpr (NRK != KNONE) {
            __builtin_amdgcn_sched_barrier(0);
            ra_reads<OP, NRK, NRBUF, QJ, KS, MI, MIC>(c, a, b0, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        ra_group<OP, QI, QJ, NRK, NRBUF, KS, MI + 1, MIC>(c, a, b0, b1, acc);
    }
}
// phase g: retire the fragments read under phase g - 1 (half tile RK of buffer RBUF), issue half tile SK into buffer SBUF, keep at most VM
// loads in flight; then the 16 MFMAs of quadrant (QI, QJ) with the reads of half tile NRK of buffer NRBUF (phase g + 1's operands) between them
template <class OP, int RK, int RBUF, int SK, int SBUF, int VM, int QI, int QJ, int NRK, int NRBUF>
__device__ __forceinline__ void phase_ra(typename OP::Ctx& c, u32x4_t (&a)[2][4], u32x4_t (&b0)[2][2], u32x4_t (&b1)[2][2], f32x4_t (&acc)[4][8]) {
    if constexpr (SK != KNONE) OP::template stage<SK, SBUF>(c);
    wait_vm<VM * OP::kLoadsPerPiecePair / 2>();
    if constexpr (RK == KA0 || RK == KA1) OP::template retire_a<RK, RBUF>(c, a);
    if constexpr (RK == KB0) OP::retire_b(b0);
    if constexpr (RK == KB1) OP::retire_b(b1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    constexpr int MIC = QI ? OP::kMi1 : 4;
    ra_group<OP, QI, QJ, NRK, NRBUF, 0, 0, MIC>(c, a, b0, b1, acc);
    ra_group<OP, QI, QJ, NRK, NRBUF, 1, 0, MIC>(c, a, b0, b1, acc);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
// the caller has read S_0 (B1 of the first even tile) into b1 and S_1 (its A0) into `a` and covered S_2 (prologue wait vmcnt(10))
template <class OP, bool LAST>
__device__ __forceinline__ void two_tiles_ra(typename OP::Ctx& c, u32x4_t (&a)[2][4], u32x4_t (&b0)[2][2], u32x4_t (&b1)[2][2], f32x4_t (&acc)[4][8]) {
    if constexpr (!LAST) {
        phase_ra<OP, KA0, 0, KB1, 0, 10, 0, 1, KB0, 0>(c, a, b0, b1, acc);
        phase_ra<OP, KB0, 0, KA0, 0, 10, 0, 0, KA1, 0>(c, a, b0, b1, acc);
        phase_ra<OP, KA1, 0, KB0, 0, 10, 1, 0, KB0, 1>(c, a, b0, b1, acc);
        phase_ra<OP, KB0, 1, KA1, 0, 10, 1, 1, KA0, 1>(c, a, b0, b1, acc);
        OP::next_ktile(c);
        phase_ra<OP, KA0, 1, KB0, 1, 10, 0, 0, KB1, 1>(c, a, b0, b1, acc);
        phase_ra<OP, KB1, 1, KA0, 1, 10, 0, 1, KA1, 1>(c, a, b0, b1, acc);
        phase_ra<OP, KA1, 1, KB1, 1, 10, 1, 1, KB1, 0>(c, a, b0, b1, acc);
        phase_ra<OP, KB1, 0, KA1, 1, 10, 1, 0, KA0, 0>(c, a, b0, b1, acc);
        OP::next_ktile(c);
    } else {
        phase_ra<OP, KA0, 0, KNONE, 0, 8, 0, 1, KB0, 0>(c, a, b0, b1, acc);
        phase_ra<OP, KB0, 0, KNONE, 0, 6, 0, 0, KA1, 0>(c, a, b0, b1, acc);
        phase_ra<OP, KA1, 0, KNONE, 0, 4, 1, 0, KB0, 1>(c, a, b0, b1, acc);
        phase_ra<OP, KB0, 1, KNONE, 0, 2, 1, 1, KA0, 1>(c, a, b0, b1, acc);
        phase_ra<OP, KA0, 1, KNONE, 0, 0, 0, 0, KB1, 1>(c, a, b0, b1, acc);
        phase_ra<OP, KB1, 1, KNONE, 0, 0, 0, 1, KA1, 1>(c, a, b0, b1, acc);
        phase_ra<OP, KA1, 1, KNONE, 0, 0, 1, 1, KNONE, 0>(c, a, b0, b1, acc);
        phase_ra<OP, KNONE, 0, KNONE, 0, 0, 1, 0, KNONE, 0>(c, a, b0, b1, acc);
    }
}

// ---- epilogue through LDS ---------------------------------------------------------------------------------------------
// Measured (tools/ab_gemm.py, ablation "nostore"): storing straight out of the MFMA layout (lane = 4 columns of one row; a wave
// store = 16 rows x 32 B) costs 29 of 92 us at N = 3072, K = 1024 -- ~7 B/clk/CU, store-issue-bound, nothing to overlap it with at
// one workgroup per CU.  Here every wave transposes its own 128 x 64 block through its PRIVATE 16 KiB of the (now idle) LDS, in
// two passes of 64 rows of f32: written as the accumulators lie (ds_write_b128, 16-B unit ^ (row & 7): conflict-free for the
// 8-lane write groups and for both read patterns below), read back row-major, so that a lane owns 8 (bf16 out) or 4 (f32 out)
// consecutive columns and a wave instruction covers 8 x 128 B or 4 x 256 B of whole cache lines -- for the stores AND for the
// epilogue's side inputs (residual rows, GELU pre-activations).  The elementwise math runs in the row layout.  No barriers:
// the region is the wave's own; LDS operations of one wave execute in order.
__device__ __forceinline__ float4 ld4f(const float* p) {
    const uint4 v = ldg16(p);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// The wave's 128 x 64 block may be two row strips / two column strips of the tile: pass h (accumulator rows 4h .. 4h+3) starts at
// tile row mrow0 + h * RSTRIDE, block columns [32s, 32s + 32) sit at tile column ncol0 + s * CSTRIP (NT kernel: one solid block,
// RSTRIDE = 64, CSTRIP = 32; TN kernel: strips 128 apart, so that its DMA reads whole 256-B runs of the token-major operands).
// ROWS = rows of the wave's block that exist (128; 112 in the 224-row NT tile, whose second pass has 48 rows).
// 16-byte output store with a cache policy (A/B of the epilogue's write path): 0 = plain (the line stays in the XCD's
// L2 until evicted), 1 = nt (streaming hint), 2 = sc1 (write-through: the line is not kept in L2, so a tile's 229 KiB of output
// do not push the next tile's operand panels out of the 4-MiB L2)
typedef __attribute__((ext_vector_type(4))) unsigned int p8_u32v4_t;
template <int POL>
__device__ __forceinline__ void p8_st16(void* ptr, const uint4& v) {
    if constexpr (POL == 1) {
        __builtin_nontemporal_store(p8_u32v4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<p8_u32v4_t*>(ptr));
    } else if constexpr (POL == 2) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(p8_u32v4_t{v.x, v.y, v.z, v.w}) : "memory");
    } else {
        *reinterpret_cast<uint4*>(ptr) = v;
    }
}
template <int POL>
__device__ __forceinline__ void p8_store8(bf16_t* p, const float (&o)[8]) {
    p8_st16<POL>(p, make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])));
}
template <int POL>
__device__ __forceinline__ void p8_store8(float* p, const float (&o)[8]) {
    p8_st16<POL>(p, make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])));
    p8_st16<POL>(p + 4, make_uint4(__float_as_uint(o[4]), __float_as_uint(o[5]), __float_as_uint(o[6]), __float_as_uint(o[7])));
}
template <int POL>
__device__ __forceinline__ void p8_store4(float* p, float4 v) {
    p8_st16<POL>(p, make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)));
}
template <int POL>
__device__ __forceinline__ void p8_store4(bf16_t* p, float4 v) { store4(p, v); }

template <typename Tout, int EPI, int RSTRIDE = 64, int CSTRIP = 32, int ROWS = 128>
__device__ __forceinline__ void epilogue_lds(const KArgs& p, f32x4_t (&acc)[4][8], char* wsm, int mrow0, int ncol0, int lane) {
    const int fr = lane & 15, g = lane >> 4;
    const uint32_t wbase = (uint32_t)(fr * 256);
    constexpr bool WIDE = sizeof(Tout) == 2;          // bf16 out: 8 columns per lane, 8 rows per wave instruction
    constexpr int ITS = WIDE ? 8 : 16, RSTEP = WIDE ? 8 : 4, NV = WIDE ? 8 : 4;
    const int cg = WIDE ? (lane & 7) : (lane & 15);   // column group of this lane
    const int rsub = WIDE ? (lane >> 3) : (lane >> 4);
    const int n = ncol0 + ((cg * NV) >> 5) * CSTRIP + ((cg * NV) & 31);
    const bool nok = n < p.N;
    const int nc = nok ? n : 0;
    // bias of the lane's columns (row-independent)
    float bias[NV];
#pragma unroll
    for (int q = 0; q < NV / 4; ++q) {
        const int nb = nc + 4 * q;
        const float* bp = (EPI != MTP_EPI_DGELU && EPI != MTP_EPI_MUL && p.bias) ? p.bias + (p.bias_mod > 0 ? nb % p.bias_mod : nb) : reinterpret_cast<const float*>(&g_zero16);
        const float4 b = ld4f(bp);
        bias[4 * q + 0] = b.x; bias[4 * q + 1] = b.y; bias[4 * q + 2] = b.z; bias[4 * q + 3] = b.w;
    }
    Tout* C = reinterpret_cast<Tout*>(p.C);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // ---- accumulators of rows [h*64, h*64+64) -> LDS as they lie: row = mfl*16 + fr, unit = nf*4 + g
#pragma unroll
        for (int mfl = 0; mfl < (h ? (ROWS - 64) / 16 : 4); ++mfl)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf)
                *reinterpret_cast<f32x4_t*>(wsm + mfl * 4096 + wbase + (((nf * 4 + g) ^ (fr & 7)) << 4)) = acc[nf][h * 4 + mfl];
        // ---- back in row layout, 8 wave instructions (= 64 or 32 rows) per batch
#pragma unroll
        for (int bt = 0; bt < ITS / 8; ++bt) {
            const int mb = mrow0 + h * RSTRIDE + bt * 8 * RSTEP + rsub;     // first row of this lane in the batch; rows mb + it * RSTEP
            // side inputs of the whole batch first (unconditional loads on clamped rows), so that they are all in flight together
            float4 side[(EPI == MTP_EPI_BIAS_RES) ? 8 : 1];
            uint4 sideb[(EPI == MTP_EPI_DGELU || EPI == MTP_EPI_MUL) ? 8 : 1];
            float rsv[(EPI == MTP_EPI_BIAS_RES) ? 8 : 1];
            if constexpr (EPI == MTP_EPI_BIAS_RES) {
                // row -> sample and row % res_mod advance incrementally: one division per batch, not per row
                const int mc0 = mb < p.M ? mb : p.M - 1;
                int smp = mc0 / p.rows_per_sample, srem = mc0 - smp * p.rows_per_sample;
                int rrow = p.res_mod > 0 ? mc0 % p.res_mod : mc0;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    side[it] = ld4f(p.res + (int64_t)rrow * p.res_ld + nc);
                    rsv[it] = p.rowscale ? p.rowscale[smp] : 1.0f;
                    if (mb + (it + 1) * RSTEP < p.M) {     // (rows past the edge keep re-reading the last valid row)
                        srem += RSTEP;
                        while (srem >= p.rows_per_sample) { srem -= p.rows_per_sample; ++smp; }
                        rrow += RSTEP;
                        if (p.res_mod > 0) while (rrow >= p.res_mod) rrow -= p.res_mod;
                    }
                }
            } else if constexpr (EPI == MTP_EPI_DGELU || EPI == MTP_EPI_MUL) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    int m = mb + it * RSTEP;
                    m = m < p.M ? m : p.M - 1;
                    sideb[it] = ldg16(reinterpret_cast<const Tout*>(p.aux) + (int64_t)m * p.aux_ld + nc);
                }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = (bt * 8 + it) * RSTEP + rsub;
                const int m = mb + it * RSTEP;
                const bool ok = nok && m < p.M && (ROWS == 128 || h * 64 + r < ROWS);
                float v[NV];
                if constexpr (WIDE) {
                    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(wsm + r * 256 + (((2 * cg) ^ (r & 7)) << 4));
                    const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(wsm + r * 256 + (((2 * cg + 1) ^ (r & 7)) << 4));
                    v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3]; v[4] = a1[0]; v[5] = a1[1]; v[6] = a1[2]; v[7] = a1[3];
                } else {
                    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(wsm + r * 256 + ((cg ^ (r & 7)) << 4));
                    v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3];
                }
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] += bias[e];
                if constexpr (EPI == MTP_EPI_BIAS_GELU) {
                    if (ok) store8(reinterpret_cast<Tout*>(p.aux) + (int64_t)m * p.aux_ld + n, v);
#pragma unroll
                    for (int e = 0; e < NV; ++e) v[e] = gelu_f(v[e]);
                } else if constexpr (EPI == MTP_EPI_BIAS_GELU_DG) {
                    float d[NV];
#pragma unroll
                    for (int e = 0; e < NV; ++e) gelu_pair_f(v[e], v[e], d[e]);
                    if (ok) store8(reinterpret_cast<Tout*>(p.aux) + (int64_t)m * p.aux_ld + n, d);
                } else if constexpr (EPI == MTP_EPI_DGELU) {
                    const uint32_t w[4] = {sideb[it].x, sideb[it].y, sideb[it].z, sideb[it].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] *= dgelu_f(bf16_bits_to_f32(w[e] & 0xffffu));
                        v[2 * e + 1] *= dgelu_f(bf16_bits_to_f32(w[e] >> 16));
                    }
                } else if constexpr (EPI == MTP_EPI_MUL) {
                    const uint32_t w[4] = {sideb[it].x, sideb[it].y, sideb[it].z, sideb[it].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] *= bf16_bits_to_f32(w[e] & 0xffffu);
                        v[2 * e + 1] *= bf16_bits_to_f32(w[e] >> 16);
                    }
                } else if constexpr (EPI == MTP_EPI_BIAS_RES) {
                    v[0] = side[it].x + rsv[it] * v[0]; v[1] = side[it].y + rsv[it] * v[1];
                    v[2] = side[it].z + rsv[it] * v[2]; v[3] = side[it].w + rsv[it] * v[3];
                }
                if (ok) {
                    if constexpr (WIDE)
                        store8(C + (int64_t)m * p.ldc + n, v);
                    else
                        store4(C + (int64_t)m * p.ldc + n, make_float4(v[0], v[1], v[2], v[3]));
                }
            }
        }
    }
}

// The same epilogue through a SMALL private region (4 KiB per wave = 16 rows x 64 f32), for the persistent NT kernel: the ring is
// already being refilled with the next tile's first K-tiles while this runs.  The wave's block is `ROWS` consecutive rows x 64
// columns; side inputs are loaded 8 wave instructions (64 / 32 rows) ahead as above.
template <typename Tout, int EPI, int ROWS, int POL = 0>
__device__ __forceinline__ void epilogue_lds16(const KArgs& p, f32x4_t (&acc)[4][8], char* wsm, int mrow0, int ncol0, int lane) {
    const int fr = lane & 15, g = lane >> 4;
    constexpr bool WIDE = sizeof(Tout) == 2;
    constexpr int RSTEP = WIDE ? 8 : 4, NV = WIDE ? 8 : 4, IPP = 16 / RSTEP, PPB = 8 / IPP, NP = ROWS / 16;
    const int cg = WIDE ? (lane & 7) : (lane & 15);
    const int rsub = WIDE ? (lane >> 3) : (lane >> 4);
    const int n = ncol0 + cg * NV;
    const bool nok = n < p.N;
    const int nc = nok ? n : 0;
    float bias[NV];
#pragma unroll
    for (int q = 0; q < NV / 4; ++q) {
        const int nb = nc + 4 * q;
        const float* bp = (EPI != MTP_EPI_DGELU && EPI != MTP_EPI_MUL && p.bias) ? p.bias + (p.bias_mod > 0 ? nb % p.bias_mod : nb) : reinterpret_cast<const float*>(&g_zero16);
        const float4 b = ld4f(bp);
        bias[4 * q + 0] = b.x; bias[4 * q + 1] = b.y; bias[4 * q + 2] = b.z; bias[4 * q + 3] = b.w;
    }
    Tout* C = reinterpret_cast<Tout*>(p.C);
#pragma unroll
    for (int b0 = 0; b0 < NP; b0 += PPB) {
        const int mb = mrow0 + 16 * b0 + rsub;     // rows mb + it * RSTEP, it = 0 .. 7
        float4 side[(EPI == MTP_EPI_BIAS_RES) ? 8 : 1];
        uint4 sideb[(EPI == MTP_EPI_DGELU || EPI == MTP_EPI_MUL) ? 8 : 1];
        float rsv[(EPI == MTP_EPI_BIAS_RES) ? 8 : 1];
        if constexpr (EPI == MTP_EPI_BIAS_RES) {
            const int mc0 = mb < p.M ? mb : p.M - 1;
            int smp = mc0 / p.rows_per_sample, srem = mc0 - smp * p.rows_per_sample;
            int rrow = p.res_mod > 0 ? mc0 % p.res_mod : mc0;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                side[it] = ld4f(p.res + (int64_t)rrow * p.res_ld + nc);
                rsv[it] = p.rowscale ? p.rowscale[smp] : 1.0f;
                if (mb + (it + 1) * RSTEP < p.M) {
                    srem += RSTEP;
                    while (srem >= p.rows_per_sample) { srem -= p.rows_per_sample; ++smp; }
                    rrow += RSTEP;
                    if (p.res_mod > 0) while (rrow >= p.res_mod) rrow -= p.res_mod;
                }
            }
        } else if constexpr (EPI == MTP_EPI_DGELU || EPI == MTP_EPI_MUL) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                int m = mb + it * RSTEP;
                m = m < p.M ? m : p.M - 1;
                sideb[it] = ldg16(reinterpret_cast<const Tout*>(p.aux) + (int64_t)m * p.aux_ld + nc);
            }
        }
#pragma unroll
        for (int pp = 0; pp < PPB; ++pp) {
            const int mfl = b0 + pp;
            if (mfl < NP) {
#pragma unroll
                for (int nf = 0; nf < 4; ++nf)
                    *reinterpret_cast<f32x4_t*>(wsm + fr * 256 + (((nf * 4 + g) ^ (fr & 7)) << 4)) = acc[nf][mfl];
#pragma unroll
                for (int i = 0; i < IPP; ++i) {
                    const int it = pp * IPP + i;
                    const int r = i * RSTEP + rsub;
                    const int m = mb + it * RSTEP;
                    const bool ok = nok && m < p.M;
                    float v[NV];
                    if constexpr (WIDE) {
                        const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(wsm + r * 256 + (((2 * cg) ^ (r & 7)) << 4));
                        const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(wsm + r * 256 + (((2 * cg + 1) ^ (r & 7)) << 4));
                        v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3]; v[4] = a1[0]; v[5] = a1[1]; v[6] = a1[2]; v[7] = a1[3];
                    } else {
                        const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(wsm + r * 256 + ((cg ^ (r & 7)) << 4));
                        v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3];
                    }
#pragma unroll
                    for (int e = 0; e < NV; ++e) v[e] += bias[e];
                    if constexpr (EPI == MTP_EPI_BIAS_GELU) {
                        if (ok) p8_store8<POL>(reinterpret_cast<Tout*>(p.aux) + (int64_t)m * p.aux_ld + n, v);
#pragma unroll
                        for (int e = 0; e < NV; ++e) v[e] = gelu_f(v[e]);
                    } else if constexpr (EPI == MTP_EPI_BIAS_GELU_DG) {
                        float d[NV];
#pragma unroll
                        for (int e = 0; e < NV; ++e) gelu_pair_f(v[e], v[e], d[e]);
                        if (ok) p8_store8<POL>(reinterpret_cast<Tout*>(p.aux) + (int64_t)m * p.aux_ld + n, d);
                    } else if constexpr (EPI == MTP_EPI_DGELU) {
                        const uint32_t w[4] = {sideb[it].x, sideb[it].y, sideb[it].z, sideb[it].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] *= dgelu_f(bf16_bits_to_f32(w[e] & 0xffffu));
                            v[2 * e + 1] *= dgelu_f(bf16_bits_to_f32(w[e] >> 16));
                        }
                    } else if constexpr (EPI == MTP_EPI_MUL) {
                        const uint32_t w[4] = {sideb[it].x, sideb[it].y, sideb[it].z, sideb[it].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] *= bf16_bits_to_f32(w[e] & 0xffffu);
                            v[2 * e + 1] *= bf16_bits_to_f32(w[e] >> 16);
                        }
                    } else if constexpr (EPI == MTP_EPI_BIAS_RES) {
                        v[0] = side[it].x + rsv[it] * v[0]; v[1] = side[it].y + rsv[it] * v[1];
                        v[2] = side[it].z + rsv[it] * v[2]; v[3] = side[it].w + rsv[it] * v[3];
                    }
                    if (ok) {
                        if constexpr (WIDE)
                            p8_store8<POL>(C + (int64_t)m * p.ldc + n, v);
                        else
                            p8_store4<POL>(C + (int64_t)m * p.ldc + n, make_float4(v[0], v[1], v[2], v[3]));
                    }
                }
            }
        }
    }
}

// tile index -> (tile row, tile column): panels of 8 tile rows, rows fastest inside a panel (the 32 tiles an XCD runs at a
// time cover 8 rows x 4 columns: 12 operand panels for 32 tiles)
__device__ __forceinline__ void tile_coords(int tile, int tiles_m, int tiles_n, int plain, int& tm, int& tn) {
    if (plain) {
        tm = tile / tiles_n;
        tn = tile - tm * tiles_n;
        return;
    }
    const int per = 8 * tiles_n;
    const int grp = tile / per, r = tile - grp * per;
    const int gm = (tiles_m - grp * 8) < 8 ? (tiles_m - grp * 8) : 8;
    tn = r / gm;
    tm = grp * 8 + (r - tn * gm);
}

}  // namespace
