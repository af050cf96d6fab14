// bf16 NT GEMM for gfx950, 256 x 256 x 64 tile, 8 waves (2 in M x 4 in N, 128 x 64 outputs each), ONE workgroup per CU.
//   C[m][n] = epilogue( sum_k A[m][k] * B[n][k] )      A = activations (M, K), B = weights (N, K), both K-contiguous
// nn.Linear forward and dgrad (VIT:50-52, 78, 87), patch-embed / ConvTranspose2d as GEMM (VIT:529, 642-649) at the
// training shapes (M = B*196 tokens, K and N multiples of 128 / 256).
//
// Why this kernel exists: the 128 x 128 single-stage kernels of gemm.hip drain `vmcnt(0)` and double-barrier every K step and
// rely on 4 resident workgroups per CU to cover it -- 0.30 of the bf16 MFMA peak in the training step.  Here the main loop is a
// software pipeline in which no wait ever drains the load queue and the matrix pipe of every SIMD always has a wave to run:
//
// * LDS: 8 slots of 16 KiB = 2 K-tile buffers x 4 "half tiles" {A0, B0, B1, A1}.  A half tile is 128 rows x 64 k (128-B rows,
//   16-B slot = chunk ^ (row & 7): the measured conflict-free image of gemm.hip).  A-half h holds the tile rows with
//   (row >> 6) & 1 == h, i.e. the rows that are "sub-tile h" of BOTH wave rows; B-half h the columns with (col >> 5) & 1 == h.
//   So ONE half tile is exactly what all 8 waves read in ONE phase, and its slot can be refilled one phase later.
// * A phase = { R: ds_read one half tile into registers (8 or 4 x ds_read_b128), issue the LDS-DMA of ONE future half tile
//   (2 x global_load_lds_dwordx4 per lane), s_waitcnt vmcnt(12), s_waitcnt lgkmcnt(0); s_barrier;  M: 16 MFMAs = one
//   64 x 32 quadrant of the wave's outputs x K = 64; s_barrier }.  8 phases = 2 K-tiles per loop iteration:
//       even tile (buffer 0): read A0 | B0 | A1 | B0' ; quadrants (0,1) (0,0) (1,0) (1,1)
//       odd  tile (buffer 1): read A0'| B1'| A1'| B1''; quadrants (0,0) (0,1) (1,1) (1,0)        (' = odd tile, '' = next even)
//   The B fragments alternate between two register sets so that the half tile a phase reads is never needed by that phase's
//   own MFMAs' *other* operand set: every phase reads exactly one half tile (8, 4, 8, 4 ... reads) and issues one.
// * The stream of half tiles S_0, S_1, ... is issued in the order it is read: S_{g+1} is read in phase g, S_{g+8} is issued
//   in phase g (into the slot S_g was read from in phase g-1).  After the issue, `vmcnt(12)` leaves S_{g+3} .. S_{g+8} in
//   flight (6 half tiles = 96 KiB per workgroup, ~7 phases ~ 1.5 us ahead of their use) and guarantees S_{g+2}, which the
//   NEXT phase reads.  Waits are counted; the queue never drains in the loop.
// * The two wave groups (waves 0-3 = wave row 0, waves 4-7 = wave row 1; one wave of each group per SIMD) run ONE barrier
//   apart: while one group issues its 16 MFMAs the other does its R part, so each SIMD's matrix pipe is handed from one wave
//   to the other at every barrier.  Hazards under that stagger (interval = span between two barriers; group 0 does R_g in
//   interval 2g, group 1 in 2g+1):
//     RAW  every lane waits for its own DMA pieces of S_{g+2} in R_g, i.e. by the end of interval 2g+1 all pieces have landed;
//          S_{g+2} is read in R_{g+1} = interval 2g+2 (group 0) / 2g+3 (group 1).
//     WAR  the reads of R_g are retired (lgkmcnt(0)) before R_g's barrier, i.e. by the end of interval 2g+1 for both groups;
//          the slot is overwritten by DMA issued in R_{g+1} >= interval 2g+2.
// * ds_read / LDS-DMA / waits are inline asm: hipcc neither counts nor drains them (its own bookkeeping would put vmcnt(0)
//   in front of every ds_read that follows an LDS-DMA).  Data dependencies are made explicit with "+v" ties on the
//   lgkmcnt(0) statement; sched_barrier(0) pins the MFMA blocks between their barriers.
// * Accumulation order per output element is the same as in gemm.hip (k ascending, one MFMA per 32 k): results are
//   bit-identical to the 128-wide kernels.
#include "gemm_p8.h"

namespace {

struct P8Ctx {
    uint32_t addrA[2][2];    // ds_read base of the wave's A fragments, [buffer][k-step]  (per lane)
    uint32_t addrA1[2][2];   // same for A-half 1 (its rows are packed 48 per wave row in the 224-row tile)
    uint32_t addrB[2][2];
    uint32_t voffA, voffB;   // per-lane byte offset of the DMA source (row-in-piece, swizzled chunk, current k)
    const char* pA[2][2];    // wave-uniform DMA source row bases [half][piece]
    const char* pB[2][2];
    uint32_t m0base;         // LDS address of this wave's first DMA piece in slot 0
};

template <int IMM>
__device__ __forceinline__ void dsr(u32x4_t& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(IMM));
}

template <int SK, int SBUF>
__device__ __forceinline__ void nt_stage(const P8Ctx& c) {
    constexpr int h = (SK == KA1 || SK == KB1) ? 1 : 0;
    const uint32_t l0 = c.m0base + SBUF * P8_BUF + SK * P8_HALF;
    if constexpr (SK == KA0 || SK == KA1)
        glds2(c.voffA, c.pA[h][0], c.pA[h][1], l0, l0 + 1024);
    else
        glds2(c.voffB, c.pB[h][0], c.pB[h][1], l0, l0 + 1024);
}

// read one A half tile (8 fragments; 6 for the 48-row half of the 224-row tile) / one B half tile (4 fragments) of buffer BUF, slot K
template <int K, int BUF, int MI>
__device__ __forceinline__ void nt_read_a(const P8Ctx& c, u32x4_t (&a)[2][4]) {
    constexpr int o = K * P8_HALF;
    const uint32_t a0 = K == KA1 ? c.addrA1[BUF][0] : c.addrA[BUF][0], a1 = K == KA1 ? c.addrA1[BUF][1] : c.addrA[BUF][1];
    dsr<o + 0 * 2048>(a[0][0], a0); dsr<o + 1 * 2048>(a[0][1], a0); dsr<o + 2 * 2048>(a[0][2], a0);
    if constexpr (MI == 4) dsr<o + 3 * 2048>(a[0][3], a0);
    dsr<o + 0 * 2048>(a[1][0], a1); dsr<o + 1 * 2048>(a[1][1], a1); dsr<o + 2 * 2048>(a[1][2], a1);
    if constexpr (MI == 4) dsr<o + 3 * 2048>(a[1][3], a1);
}
template <int K, int BUF>
__device__ __forceinline__ void nt_read_b(const P8Ctx& c, u32x4_t (&b)[2][2]) {
    constexpr int o = K * P8_HALF;
    dsr<o + 0 * 2048>(b[0][0], c.addrB[BUF][0]); dsr<o + 1 * 2048>(b[0][1], c.addrB[BUF][0]);
    dsr<o + 0 * 2048>(b[1][0], c.addrB[BUF][1]); dsr<o + 1 * 2048>(b[1][1], c.addrB[BUF][1]);
}
// retire the ds_reads; the "+v" ties make every consumer of the fragments depend on this statement
template <int MI>
__device__ __forceinline__ void wait_a(u32x4_t (&a)[2][4]) {
    if constexpr (MI == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]));
}
__device__ __forceinline__ void wait_b(u32x4_t (&b)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
}

// operand policy of the pipeline (gemm_p8.h) for K-contiguous operands; MI1 = 16-row fragments in A sub-tile 1 (4: 256-row
// tile, 3: 224-row tile)
template <int MI1>
struct NtOps {
    typedef P8Ctx Ctx;
    static constexpr int kLoadsPerPiecePair = 2;   // DMA instructions per lane and half tile
    static constexpr int kMi1 = MI1;
    template <int K, int BUF> static __device__ __forceinline__ void read_a(Ctx& c, u32x4_t (&a)[2][4]) { nt_read_a<K, BUF, (K == KA1 ? MI1 : 4)>(c, a); }
    template <int K, int BUF> static __device__ __forceinline__ void read_b(Ctx& c, u32x4_t (&b)[2][2]) { nt_read_b<K, BUF>(c, b); }
    template <int SK, int SBUF> static __device__ __forceinline__ void stage(Ctx& c) { nt_stage<SK, SBUF>(c); }
    template <int K, int BUF> static __device__ __forceinline__ void retire_a(Ctx&, u32x4_t (&a)[2][4]) { wait_a<(K == KA1 ? MI1 : 4)>(a); }
    static __device__ __forceinline__ void retire_b(u32x4_t (&b)[2][2]) { wait_b(b); }
    static __device__ __forceinline__ void next_ktile(Ctx& c) { c.voffA += 128; c.voffB += 128; }
};

// MI1 = 4: 256 x 256 tile; MI1 = 3: 224 x 256 tile (wave rows of 112 = 64 + 48 rows).  M = 12544 tokens = 49 x 256 = 56 x 224: with
// 256-row tiles every ViT-L shape runs 0.766 of a whole number of rounds on the 256 CUs (196 / 588 / 784 tiles), with 224-row tiles
// 0.875 (224 / 672 / 896) at 7/8 of the time per tile -- the host picks the cheaper one per problem (p8_pick_bm).
// DMA source rows of this wave for output tile (m0, n0).  Piece q = 2 * wave + i holds rows [8q, 8q + 8) of the half tile image.
//   A-half 0 (64 rows per wave row): image row 64 wr' + x  <->  tile row WROWS wr' + x         (wr' = q >> 3)
//   A-half 1 (16 MI1 rows per wave row): image row 16 MI1 wr' + x  <->  tile row WROWS wr' + 64 + x; at MI1 = 3 the last four
//     pieces (waves 6, 7) are past the 96 image rows: they fetch a valid row into the unused tail of the 16-KiB slot, so
//     that every wave still issues two DMA instructions per half tile (the counted waits rely on it)
//   B-half h: image row 32 wc' + x  <->  tile column 64 wc' + 32 h + x                             (wc' = q >> 2)
// Rows past the matrix edge are clamped to the last complete 8-row piece (their outputs are never stored).
template <int MI1>
__device__ __forceinline__ void p8_tile_sources(const KArgs& p, P8Ctx& c, int wave, int m0, int n0, uint32_t voffA0, uint32_t voffB0) {
    constexpr int WROWS = 64 + 16 * MI1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave * 2 + i;
        int ra0 = m0 + (q >> 3) * WROWS + (q & 7) * 8;
        const int qw = q / (2 * MI1), qx = q - qw * (2 * MI1);
        int ra1 = qw < 2 ? m0 + qw * WROWS + 64 + qx * 8 : m0;
        ra0 = ra0 < p.M - 8 ? ra0 : p.M - 8;
        ra1 = ra1 < p.M - 8 ? ra1 : p.M - 8;
        c.pA[0][i] = p.A + (int64_t)ra0 * p.lda * 2;
        c.pA[1][i] = p.A + (int64_t)ra1 * p.lda * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int rb = n0 + (q >> 2) * 64 + h * 32 + (q & 3) * 8;
            rb = rb < p.N - 8 ? rb : p.N - 8;
            c.pB[h][i] = p.B + (int64_t)rb * p.ldb * 2;
        }
    }
    c.voffA = voffA0;
    c.voffB = voffB0;
}
// prologue: S_0 .. S_7 = B1 A0 B0 A1 of K-tile 0 (buffer 0), B0 A0 B1 A1 of K-tile 1 (buffer 1)
__device__ __forceinline__ void p8_issue_prologue(P8Ctx& c) {
    nt_stage<KB1, 0>(c); nt_stage<KA0, 0>(c); nt_stage<KB0, 0>(c); nt_stage<KA1, 0>(c);
    c.voffA += 128; c.voffB += 128;
    nt_stage<KB0, 1>(c); nt_stage<KA0, 1>(c); nt_stage<KB1, 1>(c); nt_stage<KA1, 1>(c);
    c.voffA += 128; c.voffB += 128;
}

// MI1 = 4: 256 x 256 tile; MI1 = 3: 224 x 256 tile (wave rows of 112 = 64 + 48 rows).  M = 12544 tokens = 49 x 256 = 56 x 224: with
// 256-row tiles every ViT-L shape runs 0.766 of a whole number of rounds on the 256 CUs (196 / 588 / 784 tiles), with 224-row tiles
// 0.875 (224 / 672 / 896) at 7/8 of the time per tile -- the host picks the cheaper one per problem (p8_pick_bm).
// PS = 1: persistent -- gridDim.x workgroups (one per CU) walk the tiles t = blockIdx.x, + gridDim.x, ...; when a tile's main loop
// ends, the NEXT tile's first two K-tiles are issued into the (now idle) ring before the epilogue runs, and the epilogue transposes
// through 4 KiB per wave beyond the ring: the first-load latency of a tile and the workgroup hand-over are hidden behind the
// previous tile's stores.  The load queue is drained (vmcnt(0)) once per tile, after the epilogue: the counted waits of the main
// loop assume that only the DMA stream is in flight.
template <typename Tout, int EPI, int MI1, int XP, int PS, int SP = 0>
__global__ __launch_bounds__(P8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_p8_kernel(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WROWS = 64 + 16 * MI1, BM = 2 * WROWS;      // rows per wave row, rows per tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, g = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.tiles_n, ntiles = tiles_m * tiles_n;
    const int plain = p.order & 1;
    const int pairs = p.k_tiles >> 1;

    P8Ctx c;
    {
        // fragment row fr of a 16-row group, 16-B slot (chunk ^ (row & 7)) with chunk = 4 * kstep + g: k-step 1 flips byte bit 6
        const uint32_t lanepart = (uint32_t)(fr * 128 + ((g ^ (fr & 7)) << 4));
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            c.addrA[b][0] = lds0 + b * P8_BUF + wr * 8192 + lanepart;
            c.addrA[b][1] = lds0 + b * P8_BUF + wr * 8192 + (lanepart ^ 64u);
            c.addrA1[b][0] = lds0 + b * P8_BUF + wr * (MI1 * 2048) + lanepart;     // A-half 1: MI1 * 16 rows per wave row
            c.addrA1[b][1] = lds0 + b * P8_BUF + wr * (MI1 * 2048) + (lanepart ^ 64u);
            c.addrB[b][0] = lds0 + b * P8_BUF + wc * 4096 + lanepart;
            c.addrB[b][1] = lds0 + b * P8_BUF + wc * 4096 + (lanepart ^ 64u);
        }
        c.m0base = lds0 + wave * 2048;
    }
    const uint32_t lanesrc = (uint32_t)(((lane & 7) ^ (lane >> 3)) << 4);
    const uint32_t voffA0 = (uint32_t)((lane >> 3) * (int)p.lda * 2) + lanesrc;
    const uint32_t voffB0 = (uint32_t)((lane >> 3) * (int)p.ldb * 2) + lanesrc;

    int tile = blockIdx.x;
    int tm, tn;
    tile_coords(plain ? tile : xcd_remap(tile, ntiles), tiles_m, tiles_n, plain, tm, tn);
    int m0 = tm * BM, n0 = tn * P8_BN;
    p8_tile_sources<MI1>(p, c, wave, m0, n0, voffA0, voffB0);
    p8_issue_prologue(c);

    u32x4_t a[2][4], b0[2][2], b1[2][2];
    f32x4_t acc[4][8];
    for (;;) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        wait_vm<12>();   // S_0, S_1 have landed (this lane's pieces)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        nt_read_b<KB1, 0>(c, b1);
        wait_b(b1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (!(XP & 2) && wr == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind wave row 0
        __builtin_amdgcn_sched_barrier(0);

        for (int it = 0; it < pairs - 1; ++it) two_tiles<NtOps<MI1>, false, XP>(c, a, b0, b1, acc);
        two_tiles<NtOps<MI1>, true, XP>(c, a, b0, b1, acc);

        __builtin_amdgcn_sched_barrier(0);
        if (!(XP & 2) && wr == 0) __builtin_amdgcn_s_barrier();   // re-align: every wave has finished its last phase behind this barrier
        __builtin_amdgcn_sched_barrier(0);

        if constexpr (PS) {
            const int em0 = m0, en0 = n0;
            tile += gridDim.x;
            const bool more = tile < ntiles;
            if (more) {   // the ring is idle: the next tile's first two K-tiles go out before this tile's stores
                tile_coords(plain ? tile : xcd_remap(tile, ntiles), tiles_m, tiles_n, plain, tm, tn);
                m0 = tm * BM; n0 = tn * P8_BN;
                p8_tile_sources<MI1>(p, c, wave, m0, n0, voffA0, voffB0);
                p8_issue_prologue(c);
            }
            __builtin_amdgcn_sched_barrier(0);
            epilogue_lds16<Tout, EPI, WROWS, SP>(p, acc, smem + P8_LDS + wave * 4096, em0 + wr * WROWS, en0 + wc * 64, lane);
            if (!more) break;
            __builtin_amdgcn_sched_barrier(0);
            wait_vm<0>();   // stores and side loads of the epilogue retired: only the DMA stream is counted from here on
        } else {
            if constexpr (SP != 0) {   // (store-policy A/B: the one-shot kernel through the same small-region epilogue)
                epilogue_lds16<Tout, EPI, WROWS, SP>(p, acc, smem + P8_LDS + wave * 4096, m0 + wr * WROWS, n0 + wc * 64, lane);
            } else if (!(XP & 4) || p.M < 0) {
                if constexpr (XP & 16)   // A/B: straight out of the MFMA layout (the epilogue of the 128-wide kernels)
                    epilogue<Tout, EPI, 8>(p, acc, m0 + wr * 128, n0 + wc * 64, lane);
                else
                    epilogue_lds<Tout, EPI, 64, 32, WROWS>(p, acc, smem + wave * P8_HALF, m0 + wr * WROWS, n0 + wc * 64, lane);
            }
            break;
        }
    }
}

template <typename Tout, int EPI, int MI1, int XP, int PS = 0, int SP = 0>
int launch_p8_kernel(const KArgs& a, int ntiles, hipStream_t stream) {
    constexpr int LDS = P8_LDS + ((PS || SP) ? 8 * 4096 : 0);
    static unsigned long long optin = 0;   // 128 / 160 KiB of dynamic LDS needs the opt-in once per kernel and device
    if (const int e = mtp_optin_lds((const void*)gemm_nt_p8_kernel<Tout, EPI, MI1, XP, PS, SP>, LDS, optin)) return e;
    const int cus = mtp_stream_cus(stream);
    const int grid = PS ? (ntiles < cus ? ntiles : cus) : ntiles;
    hipLaunchKernelGGL((gemm_nt_p8_kernel<Tout, EPI, MI1, XP, PS, SP>), dim3(grid), dim3(P8_THREADS), LDS, stream, a);
    return mtp_launch_status();
}

// rows per tile: whole rounds of one workgroup per CU cost (rows per tile) each -- take the cheaper of 256 and 224
int p8_pick_bm(int64_t M, int64_t N, int64_t cus) {
    const int64_t tn = (N + P8_BN - 1) / P8_BN;
    const int64_t t256 = ((M + 255) / 256) * tn, t224 = ((M + 223) / 224) * tn;
    const int64_t c256 = ((t256 + cus - 1) / cus) * 256, c224 = ((t224 + cus - 1) / cus) * 224;
    return c224 < c256 ? 224 : 256;
}

template <typename Tout, int EPI>
int launch_p8(const KArgs& k, int flags, hipStream_t stream) {
    const int bm = (flags & 1) ? 224 : (flags & 4) ? 256 : p8_pick_bm(k.M, k.N, mtp_stream_cus(stream));
    const int tiles_m = (k.M + bm - 1) / bm, tiles_n = (k.N + P8_BN - 1) / P8_BN;
    KArgs a = k;
    a.tiles_n = tiles_n;
    a.k_tiles = k.K / 64;
    a.order = (flags >> 1) & 1;
    a.atomic_out = 0;
    const int ntiles = tiles_m * tiles_n;
    // persistent tiles: default for problems of more than one round of 224-row tiles (measured, tools/ab_gemm.py: +3...4 % at N = 3072 /
    // 4096, K = 1024 and on the FPN GEMM, nothing to gain on one-round problems; the 256-row instantiations spill 2-17 VGPRs with the
    // second tile loop and stay opt-in).  flags bit 8 forces it, bit 9 forbids it (A/B).
    const bool persist = (flags & 256) || (!(flags & 512) && bm == 224 && ntiles > mtp_stream_cus(stream));
    // store policy of the epilogue (224-row tiles; flags bits 13-14: 0 = by epilogue, 1 = nt, 2 = sc1 write-through, 3 = plain).  Measured with
    // rotating output buffers (tools/ab_gemm.py, MTP_AB_ROTATE=8: in the training step every GEMM writes fresh memory) and in the step
    // itself (profiles/r03_ab_store_policy.txt): nt wins for the bf16 outputs (+2...4 %), sc1 for the f32 residual epilogue (+1...5 %:
    // its 103 MB of output per launch do not evict the operand panels from the 4-MiB L2s); whole step +1.0 %.
    int sp = (flags >> 13) & 3;
    if (sp == 0) sp = (EPI == MTP_EPI_BIAS_RES) ? 2 : 1;
    if (bm == 224 && sp == 1) return persist ? launch_p8_kernel<Tout, EPI, 3, 0, 1, 1>(a, ntiles, stream) : launch_p8_kernel<Tout, EPI, 3, 0, 0, 1>(a, ntiles, stream);
    if (bm == 224 && sp == 2) return persist ? launch_p8_kernel<Tout, EPI, 3, 0, 1, 2>(a, ntiles, stream) : launch_p8_kernel<Tout, EPI, 3, 0, 0, 2>(a, ntiles, stream);
    if (persist) {
        if (bm == 224) return launch_p8_kernel<Tout, EPI, 3, 0, 1>(a, ntiles, stream);
        return launch_p8_kernel<Tout, EPI, 4, 0, 1>(a, ntiles, stream);
    }
    if (bm == 224) return launch_p8_kernel<Tout, EPI, 3, 0>(a, ntiles, stream);
    return launch_p8_kernel<Tout, EPI, 4, 0>(a, ntiles, stream);
}

template <typename Tout>
int dispatch_p8(const KArgs& k, int epi, int flags, hipStream_t s) {
    switch (epi) {
        case MTP_EPI_BIAS: return launch_p8<Tout, MTP_EPI_BIAS>(k, flags, s);
        case MTP_EPI_BIAS_GELU: return launch_p8<Tout, MTP_EPI_BIAS_GELU>(k, flags, s);
        case MTP_EPI_DGELU: return launch_p8<Tout, MTP_EPI_DGELU>(k, flags, s);
        case MTP_EPI_BIAS_GELU_DG: return launch_p8<Tout, MTP_EPI_BIAS_GELU_DG>(k, flags, s);
        case MTP_EPI_MUL: return launch_p8<Tout, MTP_EPI_MUL>(k, flags, s);
        default: return MTP_ERR_UNSUPPORTED;
    }
}

}  // namespace

int mtp_nt_p8_fits(const KArgs& k, int out_dtype, int epi) {
    // preconditions: whole K tiles in pairs, 8-row DMA pieces, 32-bit DMA offsets, an epilogue instantiation
    if (k.K < 128 || (k.K % 128) || (k.M % 8) || (k.N % 8) || k.M < 8 || k.N < 8) return 0;
    if ((uint64_t)k.lda * 2 * 8 + (uint64_t)k.K * 2 >= (1ull << 31) || (uint64_t)k.ldb * 2 * 8 + (uint64_t)k.K * 2 >= (1ull << 31)) return 0;
    if (epi == MTP_EPI_BIAS_RES) return out_dtype == MTP_F32;
    if (out_dtype == MTP_BF16) return epi == MTP_EPI_BIAS || epi == MTP_EPI_BIAS_GELU || epi == MTP_EPI_DGELU || epi == MTP_EPI_BIAS_GELU_DG || epi == MTP_EPI_MUL;
    return out_dtype == MTP_F32 && epi == MTP_EPI_BIAS;
}

int mtp_nt_p8_launch(const KArgs& k, int out_dtype, int epi, int flags, hipStream_t stream) {
    if (!mtp_nt_p8_fits(k, out_dtype, epi)) return MTP_ERR_UNSUPPORTED;
    if (epi == MTP_EPI_BIAS_RES) return launch_p8<float, MTP_EPI_BIAS_RES>(k, flags, stream);
    if (out_dtype == MTP_BF16) return dispatch_p8<bf16_t>(k, epi, flags, stream);
    return launch_p8<float, MTP_EPI_BIAS>(k, flags, stream);
}
