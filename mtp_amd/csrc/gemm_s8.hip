// bf16 NT GEMM for gfx950 on 128 x 256 x 64 STRIPS with TWO accumulator sets per wave: the epilogue of strip i runs inside the
// K loop of strip i + 1.   C[m][n] = epilogue( sum_k A[m][k] * B[n][k] ), A (M, K) activations, B (N, K) weights.
// nn.Linear forward / dgrad at the training shapes with a short contraction (VIT:55-62, 97, 390, 428-431: K = C = 1024).
//
// Why: in gemm_p8.hip a 224 x 256 tile at K = 1024 spends 16 us in its main loop (0.87 of the MFMA issue rate)
// and 7-13 us in its epilogue, during which all eight waves of the CU store and no matrix pipe runs -- 30-45 % of every K = 1024
// launch.  Overlapping tile i's epilogue with tile i + 1's loop needs a second set of accumulators; with 128 of the wave's 256
// registers already holding the 128 x 64 block of gemm_p8 there is none.  Here the unit of work is HALF as tall:
//   * 8 waves = 2 (M) x 4 (N), a wave owns 64 x 64 outputs = 64 accumulator registers; the finished strip's 64 registers
//     (`prev`) stay live while the next strip accumulates into `acc`: 128 registers, as before.
//   * the price is 1.5 x the L2 -> LDS bytes per flop (a strip K-tile is 16 KiB of A + 32 KiB of B for 4.2 MFLOP); consecutive
//     strips of a workgroup's XCD share their B panel in L2, so the miss traffic is that of the 256-row tiles.
//   * the epilogue needs no LDS: two v_permlane16_swap_b32 turn the MFMA layout (lane = 4 consecutive columns of one row) into
//     8 consecutive bf16 columns per lane (16-byte stores, 64 contiguous bytes per row and instruction); it is cut into
//     8 (bf16 out) / 16 (f32 out) SLICES of a few dozen VALU instructions + one store, each issued in the read part of one odd
//     phase of strip i + 1's first ten K-tiles (schedule: S8Epi::use_ktile), its side input loaded one K-tile earlier.
// MEASURED (round 5, profiles/r05_ab_gemm_s8_strip*.txt): bit-identical to the other NT kernels on every epilogue, and 10-38 % SLOWER than the
// 8-wave kernel on the ViT-L shapes -- a phase takes ~1100 cycles instead of ~600: the strip's read part carries 12 / 4 fragment reads and
// three DMA pieces per wave (the 256-row tile: 8 / 4 and two), and the read part, not the matrix pipe, is what both kernels' phases wait
// for (profiles/r05_ab_p8_loop_ablations.txt).  It WINS where the 256 x 256 tiles leave most CUs idle: problems of 40-128 tiles (ViT-B at
// batch 32 with N = C, InternImage-XL's 768- / 1536-channel levels) run 8-19 % faster on twice as many, half as long work units, and
// that is where mtp_gemm_nt dispatches it (gemm.hip: nt_s8_mode).
//
// Pipeline (same two-wave-group stagger, barriers and counted waits as gemm_p8.h; re-derived for 3 half tiles per K-tile):
//   LDS ring = 3 K-tile buffers x { A (128 rows), B0, B1 (128 columns each: the 32-column sub-tile h of every wave column) },
//   16 KiB each = 144 KiB.  K-tile T lives in buffer T % 3.  Two phases per K-tile:
//     even 2T  : R  ds_read A_T (8) and B0_T (4);  DMA  A_{T+2} piece 1, B1_{T+2} (2 pieces);  s_waitcnt vmcnt(12)
//                M  16 MFMAs  a x b -> columns  0 .. 31 of the wave's block
//     odd  2T+1: R  ds_read B1_T (4);              DMA  B0_{T+3} (2 pieces), A_{T+3} piece 0;  [epilogue slice]  vmcnt(11)
//                M  16 MFMAs  a x b -> columns 32 .. 63
//   A piece = 1 KiB per wave (8 rows x 128 B, global_load_lds_dwordx4); a half tile = 16 pieces = 2 per wave.
//   WAR: a slot is refilled one phase after its last read (A_T, B0_T read in 2T, refilled from 2T+1; B1_T read in 2T+1, refilled
//        in 2T+2) -- under the stagger the same margin as gemm_p8.h (reads retire before the reading phase's barrier).
//   RAW: the wait of phase g covers what phase g + 1 reads.  Loads issued after A_{T+1} piece 1 (first of even phase 2T-2) at
//        the wait of odd phase 2T+1: 2 + 3 + 3 + 3 = 11; after B1_{T} piece 1 (last of even phase 2T-4) at the wait of even
//        phase 2T: 3 + 3 + 3 + 3 = 12.  A piece is in flight for >= 3 phases (~0.8 us).
//   The stream NEVER drains: the DMA of the next strip's first K-tiles follows the last K-tile of the current one (the issue
//   side runs 3 K-tiles ahead of the MFMAs and switches strips by itself); after a workgroup's last strip it re-fetches that
//   strip's first K-tiles into free slots (never read) so that the counted waits stay uniform.
//   Side inputs (EPI_MUL's factor, the f32 residual) and the bias are ordinary global loads issued by inline asm INTO the same
//   in-order vmcnt stream.  They are not counted in the waits above (a wait that ignores k younger loads is k loads stricter,
//   never weaker); a load issued after the DMA of phase g has 12 younger DMA pieces at the wait of phase g + 4 and is
//   therefore complete there.  Stores are compiler-visible (they need no wait) and only make a wait stricter.
//   Accumulation order per element = k ascending, one MFMA per 32 k: bit-identical to every other NT kernel of the library.
#include "gemm_p8.h"

namespace {

constexpr int S8_BM = 128, S8_BN = 256, S8_THREADS = 512;
constexpr int S8_HALF = 16384, S8_BUF = 3 * S8_HALF, S8_LDS = 3 * S8_BUF;   // 144 KiB
constexpr int SA = 0, SB0 = S8_HALF, SB1 = 2 * S8_HALF;                     // slots inside a K-tile buffer
constexpr int S8_NPEEL = 10;                                                // K-tiles of a strip that carry epilogue slices
__device__ __attribute__((aligned(16))) const float g_s8_one[4] = {1.0f, 1.0f, 1.0f, 1.0f};

__device__ __forceinline__ uint32_t s8_ring_next(uint32_t off) {
    const uint32_t n = off + S8_BUF;
    return n == 3u * S8_BUF ? 0u : n;
}

// three LDS-DMA pieces (1 KiB each per wave): one of operand X, two of operand Y, in the given order
__device__ __forceinline__ void glds_x_yy(uint32_t vx, const char* px, uint32_t lx, uint32_t vy, const char* py0, const char* py1, uint32_t ly) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %4\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %6\n\t"
        "s_mov_b32 m0, %7\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %8\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(vx), "v"(vy), "s"(lx), "s"(px), "s"(ly), "s"(py0), "s"(ly + 1024u), "s"(py1)
        : "memory");
}
__device__ __forceinline__ void glds_yy_x(uint32_t vy, const char* py0, const char* py1, uint32_t ly, uint32_t vx, const char* px, uint32_t lx) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %4\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %6\n\t"
        "s_mov_b32 m0, %7\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, %8\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(vy), "v"(vx), "s"(ly), "s"(py0), "s"(ly + 1024u), "s"(py1), "s"(lx), "s"(px)
        : "memory");
}

template <int IMM>
__device__ __forceinline__ void s8_dsr(u32x4_t& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(IMM));
}
// 16-byte global load through the in-order vmcnt stream, invisible to hipcc's own wait bookkeeping (see the header): the
// destination is valid only after a counted wait that has >= 12 younger loads behind it, or after vmcnt(0)
__device__ __forceinline__ void s8_ldg(u32x4_t& d, const void* base, uint32_t off) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void s8_ldg1(uint32_t& d, const void* base, uint32_t off) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(d) : "v"(off), "s"(base) : "memory");
}

struct S8Ctx {
    uint32_t lpA0, lpA1, lpB0, lpB1;   // ds_read addresses of the wave's fragments in buffer 0 (per lane; k-step 0 / 1)
    uint32_t rbuf, nbuf, pbuf;         // ring: buffer of the K-tile being multiplied, the next one, the previous one
    uint32_t voffA, voffB;             // per-lane DMA source offset incl. the k of the K-tile being issued
    uint32_t voffA0, voffB0;
    const char* pA[2];                 // wave-uniform DMA source rows of the strip being issued [piece]
    const char* pB[2][2];              // [half][piece]
    uint32_t m0base;                   // LDS address of this wave's first piece in slot 0 of buffer 0
    int ikt;                           // K-tile (of the strip being issued) the next triple belongs to
    int istrip;                        // its strip id
};

struct S8Geo {
    int tiles_m, tiles_n, nstrips, plain, kt, grid;
};

__device__ __forceinline__ void s8_strip_origin(const S8Geo& G, int strip, int& m0, int& n0) {
    int tm, tn;
    tile_coords(G.plain ? strip : xcd_remap(strip, G.nstrips), G.tiles_m, G.tiles_n, G.plain, tm, tn);
    m0 = tm * S8_BM;
    n0 = tn * S8_BN;
}

// DMA source rows of this wave for the strip at (m0, n0).  Piece q = 2 * wave + i holds rows [8q, 8q + 8) of a half-tile image.
//   A: image row r <-> strip row r (wave row wr reads image rows [64 wr, 64 wr + 64))
//   B-half h: image row 32 wc' + x <-> strip column 64 wc' + 32 h + x   (wc' = q >> 2)
// Rows past the matrix edge are clamped to the last complete 8-row piece (their outputs are never stored).
__device__ __forceinline__ void s8_sources(const KArgs& p, S8Ctx& c, int wave, int m0, int n0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave * 2 + i;
        int ra = m0 + q * 8;
        ra = ra < p.M - 8 ? ra : p.M - 8;
        c.pA[i] = p.A + (int64_t)ra * p.lda * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int rb = n0 + (q >> 2) * 64 + h * 32 + (q & 3) * 8;
            rb = rb < p.N - 8 ? rb : p.N - 8;
            c.pB[h][i] = p.B + (int64_t)rb * p.ldb * 2;
        }
    }
}

// the issue side moves on by one K-tile; at the end of a strip it switches to the workgroup's next strip (or re-walks the last one)
__device__ __forceinline__ void s8_advance(const KArgs& p, const S8Geo& G, S8Ctx& c, int wave) {
    c.voffA += 128;
    c.voffB += 128;
    if (++c.ikt == G.kt) {
        c.ikt = 0;
        c.voffA = c.voffA0;
        c.voffB = c.voffB0;
        if (c.istrip + G.grid < G.nstrips) {
            c.istrip += G.grid;
            int m0, n0;
            s8_strip_origin(G, c.istrip, m0, n0);
            s8_sources(p, c, wave, m0, n0);
        }
    }
}
// a K-tile's triple (B0, A, B1) goes out in two halves, in this order: B0 pieces 0 / 1, A piece 0 | A piece 1, B1 pieces 0 / 1
__device__ __forceinline__ void s8_issue_first(const S8Ctx& c, uint32_t buf) {
    const uint32_t l = c.m0base + buf;
    glds_yy_x(c.voffB, c.pB[0][0], c.pB[0][1], l + SB0, c.voffA, c.pA[0], l + SA);
}
__device__ __forceinline__ void s8_issue_second(const S8Ctx& c, uint32_t buf) {
    const uint32_t l = c.m0base + buf;
    glds_x_yy(c.voffA, c.pA[1], l + SA + 1024u, c.voffB, c.pB[1][0], c.pB[1][1], l + SB1);
}

__device__ __forceinline__ void s8_read_ab0(const S8Ctx& c, u32x4_t (&a)[2][4], u32x4_t (&b)[2][2]) {
    const uint32_t a0 = c.lpA0 + c.rbuf, a1 = c.lpA1 + c.rbuf, b0 = c.lpB0 + c.rbuf, b1 = c.lpB1 + c.rbuf;
    s8_dsr<SA + 0 * 2048>(a[0][0], a0); s8_dsr<SA + 1 * 2048>(a[0][1], a0); s8_dsr<SA + 2 * 2048>(a[0][2], a0); s8_dsr<SA + 3 * 2048>(a[0][3], a0);
    s8_dsr<SB0 + 0 * 2048>(b[0][0], b0); s8_dsr<SB0 + 1 * 2048>(b[0][1], b0);
    s8_dsr<SA + 0 * 2048>(a[1][0], a1); s8_dsr<SA + 1 * 2048>(a[1][1], a1); s8_dsr<SA + 2 * 2048>(a[1][2], a1); s8_dsr<SA + 3 * 2048>(a[1][3], a1);
    s8_dsr<SB0 + 0 * 2048>(b[1][0], b1); s8_dsr<SB0 + 1 * 2048>(b[1][1], b1);
}
__device__ __forceinline__ void s8_read_b1(const S8Ctx& c, u32x4_t (&b)[2][2]) {
    const uint32_t b0 = c.lpB0 + c.rbuf, b1 = c.lpB1 + c.rbuf;
    s8_dsr<SB1 + 0 * 2048>(b[0][0], b0); s8_dsr<SB1 + 1 * 2048>(b[0][1], b0);
    s8_dsr<SB1 + 0 * 2048>(b[1][0], b1); s8_dsr<SB1 + 1 * 2048>(b[1][1], b1);
}
__device__ __forceinline__ void s8_retire_ab(u32x4_t (&a)[2][4], u32x4_t (&b)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]),
                   "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
}
__device__ __forceinline__ void s8_retire_b(u32x4_t (&b)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
}

// 16 MFMAs: the wave's 64 rows x columns [32 QJ, 32 QJ + 32) x K = 64.  ZERO: the strip's first K-tile starts from C = 0.
template <int QJ, bool ZERO>
__device__ __forceinline__ void s8_mfma16(f32x4_t (&acc)[4][4], const u32x4_t (&a)[2][4], const u32x4_t (&b)[2][2]) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                if (ZERO && ks == 0) {
                    const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
                    acc[QJ * 2 + ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[ks][ni]), __builtin_bit_cast(bf16x8_t, a[ks][mi]), z, 0, 0, 0);
                } else {
                    mma(acc[QJ * 2 + ni][mi], b[ks][ni], a[ks][mi]);
                }
            }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// ---- the finished strip's epilogue, in slices ---------------------------------------------------------------------------
// Lane (fr, g) of a wave holds prev[nf][mi][c] = C[em0 + 64 wr + 16 mi + fr][en0 + 64 wc + 16 nf + 4 g + c].
// bf16 out, slice S = (mi = S & 3, j = S >> 2): the pair nf = 2j, 2j + 1.  With P = the packed pair of dwords of nf = 2j and Q of
// nf = 2j + 1, v_permlane16_swap(P_d, Q_d) (odd 16-lane rows of P <-> even rows of Q) leaves lane g with 8 consecutive columns
// of nf = 2j + (g & 1), starting at column 8 (g >> 1): (P'_0, P'_1, Q'_0, Q'_1) is one 16-byte store.  The swap is its own
// inverse, so a bf16 side input loaded in the store layout is brought to the accumulator layout by the same two swaps.
// f32 out, slice S = (mi = S & 3, nf = S >> 2): the lane's 4 columns are one 16-byte store as they lie.
template <typename Tout, int EPI, bool HASB>
struct S8Epi {
    static constexpr bool WIDE = sizeof(Tout) == 2;
    static constexpr int NS = WIDE ? 8 : 16;
    static constexpr bool SIDE = EPI == MTP_EPI_MUL || EPI == MTP_EPI_DGELU || EPI == MTP_EPI_BIAS_RES;
    static constexpr int NSIDE = SIDE ? (WIDE ? 2 : 4) : 1;
    // Schedule (K-tile of the RUNNING strip in whose odd phase slice S of the FINISHED strip is computed and stored):
    //   bf16 out: S = 0..3 (columns 0..31 of the wave's block) in K-tiles 1..4, S = 4..7 (columns 32..63) in K-tiles 6..9;
    //   f32 out : two slices per K-tile, S / 2 + 1 (K-tiles 1..8).
    // A slice's side input is loaded at the START of the even phase of the K-tile before (ahead of that phase's DMA: 12 younger
    // pieces at the wait of the odd phase three phases later, where it is used).  The bias of the columns a group of slices covers
    // is loaded the same way, one group at a time into ONE register set: bf16 out 2 x float4 at K-tiles 0 and 5, f32 out
    // 1 x float4 at K-tiles 0, 2, 4, 6 (two sets alternating); the drop-path factors of the lane's four rows at K-tile 0.
    static constexpr int use_ktile(int S) { return WIDE ? (S < 4 ? S + 1 : S + 2) : S / 2 + 1; }
    static constexpr int NBIAS = HASB ? 2 : 1;

    f32x4_t prev[4][4];
    u32x4_t bias[NBIAS];            // bf16 out: bias of nf = 2q, 2q + 1 of the current group; f32 out: of nf (set nf & 1)
    u32x4_t side[NSIDE];
    uint32_t rs[(EPI == MTP_EPI_BIAS_RES) ? 4 : 1];   // drop-path factor of the lane's row of each mi (f32 bits)
    int em0, en0;                   // origin of the finished strip
    bool live;                      // false until the workgroup's first strip has finished: slices store nothing
    int rbase, cown, cst;           // lane constants: row in the strip, own first column (nf = 0), store column (pair 0)

    __device__ __forceinline__ void init(int wr, int wc, int fr, int g) {
        rbase = wr * 64 + fr;
        cown = wc * 64 + g * 4;
        cst = wc * 64 + (g & 1) * 16 + (g >> 1) * 8;
        live = false;
    }
    template <int NF, int SET>
    __device__ __forceinline__ void load_bias(const KArgs& p) {
        if constexpr (HASB) {
            int n = en0 + cown + NF * 16;
            n = n < p.N ? n : 0;
            if (p.bias_mod > 0) n %= p.bias_mod;
            s8_ldg(bias[SET], p.bias, (uint32_t)n * 4u);
        }
    }
    __device__ __forceinline__ void load_rs(const KArgs& p) {
        if constexpr (EPI == MTP_EPI_BIAS_RES) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                int m = em0 + rbase + mi * 16;
                m = m < p.M ? m : p.M - 1;
                const uint32_t smp = p.rowscale ? (uint32_t)m / (uint32_t)p.rows_per_sample : 0u;
                s8_ldg1(rs[mi], p.rowscale ? (const void*)p.rowscale : (const void*)&g_s8_one, smp * 4u);
            }
        }
    }

    template <int S>
    __device__ __forceinline__ void load(const KArgs& p) {
        if constexpr (SIDE) {
            constexpr int mi = S & 3, q = S >> 2;
            int m = em0 + rbase + mi * 16;
            m = m < p.M ? m : p.M - 1;
            if constexpr (WIDE) {
                int n = en0 + cst + q * 32;
                n = n < p.N ? n : 0;
                s8_ldg(side[S % NSIDE], p.aux, ((uint32_t)m * (uint32_t)p.aux_ld + (uint32_t)n) * 2u);
            } else {
                int n = en0 + cown + q * 16;
                n = n < p.N ? n : 0;
                s8_ldg(side[S % NSIDE], p.res, ((uint32_t)m * (uint32_t)p.res_ld + (uint32_t)n) * 4u);
            }
        }
    }

    template <int S, int POL>
    __device__ __forceinline__ void use(const KArgs& p) {
        constexpr int mi = S & 3, q = S >> 2;
        const int m = em0 + rbase + mi * 16;
        if constexpr (WIDE) {
            constexpr int n0f = 2 * q, n1f = 2 * q + 1;
            float v0[4], v1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v0[c] = prev[n0f][mi][c];
                v1[c] = prev[n1f][mi][c];
                if constexpr (HASB) {
                    v0[c] += __uint_as_float(bias[0][c]);
                    v1[c] += __uint_as_float(bias[1][c]);
                }
            }
            const int n = en0 + cst + q * 32;
            const bool ok = live && m < p.M && n < p.N;
            if constexpr (EPI == MTP_EPI_BIAS_GELU_DG) {
                float d0[4], d1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    gelu_pair_f(v0[c], v0[c], d0[c]);
                    gelu_pair_f(v1[c], v1[c], d1[c]);
                }
                const uint4 dv = swap_pack(d0, d1);
                if (ok) p8_st16<POL>(reinterpret_cast<bf16_t*>(p.aux) + ((size_t)m * p.aux_ld + n), dv);
            } else if constexpr (EPI == MTP_EPI_MUL || EPI == MTP_EPI_DGELU) {
                const u32x4_t s = side[S % NSIDE];
                const auto r0 = __builtin_amdgcn_permlane16_swap(s[0], s[2], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(s[1], s[3], false, false);
                const uint32_t w0[2] = {r0[0], r1[0]}, w1[2] = {r0[1], r1[1]};   // accumulator layout: nf = 2q (w0), nf = 2q + 1 (w1)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float f00 = bf16_bits_to_f32(w0[e] & 0xffffu), f01 = bf16_bits_to_f32(w0[e] >> 16);
                    float f10 = bf16_bits_to_f32(w1[e] & 0xffffu), f11 = bf16_bits_to_f32(w1[e] >> 16);
                    if constexpr (EPI == MTP_EPI_DGELU) { f00 = dgelu_f(f00); f01 = dgelu_f(f01); f10 = dgelu_f(f10); f11 = dgelu_f(f11); }
                    v0[2 * e] *= f00; v0[2 * e + 1] *= f01;
                    v1[2 * e] *= f10; v1[2 * e + 1] *= f11;
                }
            }
            const uint4 ov = swap_pack(v0, v1);
            if (ok) p8_st16<POL>(reinterpret_cast<bf16_t*>(p.C) + ((size_t)m * p.ldc + n), ov);
        } else {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] = prev[q][mi][c];
                if constexpr (HASB) v[c] += __uint_as_float(bias[q & 1][c]);
            }
            if constexpr (EPI == MTP_EPI_BIAS_RES) {
                const u32x4_t s = side[S % NSIDE];
                const float r = __uint_as_float(rs[mi]);
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = __uint_as_float(s[c]) + r * v[c];
            }
            const int n = en0 + cown + q * 16;
            const bool ok = live && m < p.M && n < p.N;
            if (ok) p8_st16<POL>(reinterpret_cast<float*>(p.C) + ((size_t)m * p.ldc + n), make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
        }
    }
    // accumulator layout (4 columns of nf = 2q in x, of nf = 2q + 1 in y) -> 8 consecutive bf16 columns per lane
    static __device__ __forceinline__ uint4 swap_pack(const float (&x)[4], const float (&y)[4]) {
        const auto r0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[0], x[1]), pack_bf16x2(y[0], y[1]), false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(x[2], x[3]), pack_bf16x2(y[2], y[3]), false, false);
        return make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }

    // the loads K-tile KI of the running strip issues for the finished one (start of its even phase, ahead of the phase's DMA)
    template <int KI, int S = 0>
    __device__ __forceinline__ void even_items(const KArgs& p) {
        if constexpr (KI >= 0 && S == 0) {
            if constexpr (KI == 0) load_rs(p);
            if constexpr (WIDE) {
                if constexpr (KI == 0) { load_bias<0, 0>(p); load_bias<1, 1>(p); }
                if constexpr (KI == 5) { load_bias<2, 0>(p); load_bias<3, 1>(p); }
            } else {
                if constexpr (KI == 0 || KI == 2 || KI == 4 || KI == 6) load_bias<(KI >= 0 ? KI / 2 : 0), ((KI >= 0 ? KI / 2 : 0) & 1)>(p);
            }
        }
        if constexpr (SIDE && KI >= 0 && S < NS) {
            if constexpr (use_ktile(S) - 1 == KI) load<S>(p);
            even_items<KI, S + 1>(p);
        }
    }
    // ... and the slices it computes and stores (odd phase, behind that phase's counted wait)
    template <int KI, int POL, int S = 0>
    __device__ __forceinline__ void odd_items(const KArgs& p) {
        if constexpr (KI >= 1 && S < NS) {
            if constexpr (use_ktile(S) == KI) use<S, POL>(p);
            odd_items<KI, POL, S + 1>(p);
        }
    }
    // after the workgroup's last strip: the same items back to back, K-tile by K-tile, with a drain in place of the counted waits
    template <int KI, int POL>
    __device__ __forceinline__ void flush_from(const KArgs& p) {
        if constexpr (KI < S8_NPEEL) {
            even_items<KI>(p);
            flush_wait();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (KI >= 1) odd_items<KI, POL>(p);
            __builtin_amdgcn_sched_barrier(0);
            flush_from<KI + 1, POL>(p);
        }
    }
    __device__ __forceinline__ void flush_wait() {
        if constexpr (NSIDE == 2)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(side[0]), "+v"(side[1]));
        else if constexpr (NSIDE == 4)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(side[0]), "+v"(side[1]), "+v"(side[2]), "+v"(side[3]));
        if constexpr (HASB) asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias[0]), "+v"(bias[1]));
        if constexpr (EPI == MTP_EPI_BIAS_RES) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rs[0]), "+v"(rs[1]), "+v"(rs[2]), "+v"(rs[3]));
        if constexpr (!HASB && NSIDE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
};

// one K-tile of the running strip = an even and an odd phase.  KI = position in the strip when it carries slices of the finished one
// (0 .. S8_NPEEL - 1), -1 otherwise; FIRST: the accumulators start from zero.
template <typename Tout, int EPI, bool HASB, int KI, bool FIRST, int POL>
__device__ __forceinline__ void s8_ktile(const KArgs& p, const S8Geo& G, S8Ctx& c, S8Epi<Tout, EPI, HASB>& e, int wave, u32x4_t (&a)[2][4], u32x4_t (&b)[2][2],
                                         f32x4_t (&acc)[4][4]) {
    e.template even_items<KI>(p);
    s8_read_ab0(c, a, b);
    s8_issue_second(c, c.pbuf);
    s8_advance(p, G, c, wave);
    wait_vm<12>();
    s8_retire_ab(a, b);
    s8_mfma16<0, FIRST>(acc, a, b);

    s8_read_b1(c, b);
    s8_issue_first(c, c.rbuf);
    wait_vm<11>();
    e.template odd_items<KI, POL>(p);
    s8_retire_b(b);
    s8_mfma16<1, FIRST>(acc, a, b);
    c.pbuf = c.rbuf;
    c.rbuf = c.nbuf;
    c.nbuf = s8_ring_next(c.nbuf);
}

template <typename Tout, int EPI, bool HASB, int POL>
__global__ __launch_bounds__(S8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_s8_kernel(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, g = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
    S8Geo G;
    G.tiles_m = (p.M + S8_BM - 1) / S8_BM;
    G.tiles_n = p.tiles_n;
    G.nstrips = G.tiles_m * G.tiles_n;
    G.plain = p.order & 1;
    G.kt = p.k_tiles;
    G.grid = gridDim.x;

    S8Ctx c;
    {
        const uint32_t lanepart = (uint32_t)(fr * 128 + ((g ^ (fr & 7)) << 4));
        c.lpA0 = lds0 + wr * 8192 + lanepart;
        c.lpA1 = lds0 + wr * 8192 + (lanepart ^ 64u);
        c.lpB0 = lds0 + wc * 4096 + lanepart;
        c.lpB1 = lds0 + wc * 4096 + (lanepart ^ 64u);
        c.m0base = lds0 + wave * 2048;
        const uint32_t lanesrc = (uint32_t)(((lane & 7) ^ (lane >> 3)) << 4);
        c.voffA0 = (uint32_t)((lane >> 3) * (int)p.lda * 2) + lanesrc;
        c.voffB0 = (uint32_t)((lane >> 3) * (int)p.ldb * 2) + lanesrc;
        c.voffA = c.voffA0;
        c.voffB = c.voffB0;
        c.rbuf = 0;
        c.nbuf = S8_BUF;
        c.pbuf = 2 * S8_BUF;
        c.ikt = 0;
        c.istrip = blockIdx.x;
    }
    int cstrip = blockIdx.x, cm0, cn0;
    s8_strip_origin(G, cstrip, cm0, cn0);
    s8_sources(p, c, wave, cm0, cn0);
    // prologue: triples 0 and 1 whole, the first half of triple 2 (15 pieces per wave)
    s8_issue_first(c, 0u);
    s8_issue_second(c, 0u);
    s8_advance(p, G, c, wave);
    s8_issue_first(c, (uint32_t)S8_BUF);
    s8_issue_second(c, (uint32_t)S8_BUF);
    s8_advance(p, G, c, wave);
    s8_issue_first(c, 2u * S8_BUF);

    S8Epi<Tout, EPI, HASB> e;
    e.init(wr, wc, fr, g);
    e.em0 = cm0;
    e.en0 = cn0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) e.prev[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    u32x4_t a[2][4], b[2][2];
    f32x4_t acc[4][4];

    wait_vm<11>();   // B0_0 and A_0 have landed (this lane's pieces)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind wave row 0
    __builtin_amdgcn_sched_barrier(0);

    const int nmy = (G.nstrips - (int)blockIdx.x + G.grid - 1) / G.grid;
    for (int j = 0; j < nmy; ++j) {
        s8_ktile<Tout, EPI, HASB, 0, true, POL>(p, G, c, e, wave, a, b, acc);
        s8_ktile<Tout, EPI, HASB, 1, false, POL>(p, G, c, e, wave, a, b, acc);
        s8_ktile<Tout, EPI, HASB, 2, false, POL>(p, G, c, e, wave, a, b, acc);
        s8_ktile<Tout, EPI, HASB, 3, false, POL>(p, G, c, e, wave, a, b, acc);
        s8_ktile<Tout, EPI, HASB, 4, false, POL>(p, G, c, e, wave, a, b, acc);
        s8_ktile<Tout, EPI, HASB, 5, false, POL>(p, G, c, e, wave, a, b, acc);
        s8_ktile<Tout, EPI, HASB, 6, false, POL>(p, G, c, e, wave, a, b, acc);
        s8_ktile<Tout, EPI, HASB, 7, false, POL>(p, G, c, e, wave, a, b, acc);
        s8_ktile<Tout, EPI, HASB, 8, false, POL>(p, G, c, e, wave, a, b, acc);
        s8_ktile<Tout, EPI, HASB, 9, false, POL>(p, G, c, e, wave, a, b, acc);
        for (int k = S8_NPEEL; k < G.kt; ++k) s8_ktile<Tout, EPI, HASB, -1, false, POL>(p, G, c, e, wave, a, b, acc);
        // the strip is complete: its accumulators become `prev`, its slices ride on the next strip's K-tiles
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) e.prev[i][q] = acc[i][q];
        e.em0 = cm0;
        e.en0 = cn0;
        e.live = true;
        cstrip += G.grid;
        if (j + 1 < nmy) s8_strip_origin(G, cstrip, cm0, cn0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 0) __builtin_amdgcn_s_barrier();   // re-align the two wave rows
    __builtin_amdgcn_sched_barrier(0);
    // the last strip's epilogue has nothing to hide under
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    e.template flush_from<0, POL>(p);
}

template <typename Tout, int EPI, bool HASB, int POL>
int launch_s8_kernel(const KArgs& k, int flags, hipStream_t stream) {
    static unsigned long long optin = 0;   // 144 KiB of dynamic LDS needs the opt-in once per kernel and device
    if (const int e = mtp_optin_lds((const void*)gemm_nt_s8_kernel<Tout, EPI, HASB, POL>, S8_LDS, optin)) return e;
    KArgs a = k;
    a.tiles_n = (k.N + S8_BN - 1) / S8_BN;
    a.k_tiles = k.K / 64;
    a.order = (flags >> 1) & 1;
    const int nstrips = ((k.M + S8_BM - 1) / S8_BM) * a.tiles_n;
    const int cus = mtp_stream_cus(stream);
    const int grid = nstrips < cus ? nstrips : cus;
    hipLaunchKernelGGL((gemm_nt_s8_kernel<Tout, EPI, HASB, POL>), dim3(grid), dim3(S8_THREADS), S8_LDS, stream, a);
    return mtp_launch_status();
}

}  // namespace

// preconditions: whole K-tiles and at least S8_NPEEL + 1 of them (the slices of a strip ride on the first S8_NPEEL K-tiles of the
// next one), 8-row DMA pieces, every byte offset of the operands / outputs / side inputs in 32 bits, an instantiation
int mtp_nt_s8_fits(const KArgs& k, int out_dtype, int epi) {
    if ((k.K % 64) || k.K / 64 < S8_NPEEL + 1 || (k.M % 8) || (k.N % 8) || k.M < 8 || k.N < 8) return 0;
    const uint64_t lim = 1ull << 31;
    if ((uint64_t)k.lda * 2 * 8 + (uint64_t)k.K * 2 >= lim || (uint64_t)k.ldb * 2 * 8 + (uint64_t)k.K * 2 >= lim) return 0;
    if ((uint64_t)k.M * (uint64_t)k.ldc * 4 >= lim) return 0;
    const bool hasb = k.bias != nullptr;
    if (epi == MTP_EPI_BIAS_RES) return out_dtype == MTP_F32 && k.res && k.res_mod <= 0 && (uint64_t)k.M * (uint64_t)k.res_ld * 4 < lim;      // (with or without a bias: the data-gradient GEMMs of InternImage add a residual gradient and have none)
    if (epi == MTP_EPI_BIAS) return out_dtype == MTP_BF16 || out_dtype == MTP_F32;
    if (out_dtype != MTP_BF16) return 0;
    if (epi == MTP_EPI_BIAS_GELU_DG) return hasb && k.aux && (uint64_t)k.M * (uint64_t)k.aux_ld * 2 < lim;
    if (epi == MTP_EPI_MUL) return k.aux && (uint64_t)k.M * (uint64_t)k.aux_ld * 2 < lim;
    return 0;
}

int mtp_nt_s8_launch(const KArgs& k, int out_dtype, int epi, int flags, hipStream_t stream) {
    if (!mtp_nt_s8_fits(k, out_dtype, epi)) return MTP_ERR_UNSUPPORTED;
    const bool hasb = k.bias != nullptr;
    if (epi == MTP_EPI_BIAS_RES) return hasb ? launch_s8_kernel<float, MTP_EPI_BIAS_RES, true, 2>(k, flags, stream) : launch_s8_kernel<float, MTP_EPI_BIAS_RES, false, 2>(k, flags, stream);
    if (epi == MTP_EPI_BIAS_GELU_DG) return launch_s8_kernel<bf16_t, MTP_EPI_BIAS_GELU_DG, true, 1>(k, flags, stream);
    if (epi == MTP_EPI_MUL) return launch_s8_kernel<bf16_t, MTP_EPI_MUL, false, 1>(k, flags, stream);
    if (out_dtype == MTP_BF16) return hasb ? launch_s8_kernel<bf16_t, MTP_EPI_BIAS, true, 1>(k, flags, stream) : launch_s8_kernel<bf16_t, MTP_EPI_BIAS, false, 1>(k, flags, stream);
    return hasb ? launch_s8_kernel<float, MTP_EPI_BIAS, true, 2>(k, flags, stream) : launch_s8_kernel<float, MTP_EPI_BIAS, false, 2>(k, flags, stream);
}
