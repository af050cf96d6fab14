// bf16 NT GEMM for gfx950, 256 x 128 x 64 tile, FOUR waves (2 in M x 2 in N, 128 x 64 outputs each), TWO workgroups per CU.
//   C[m][n] = epilogue( sum_k A[m][k] * B[n][k] )      A = activations (M, K), B = weights (N, K), both K-contiguous
// Same fragments, LDS image, MFMA chain and epilogue as the 8-wave kernel of gemm_p8.hip (results are bit-identical); what
// changes is WHO shares a SIMD.  In gemm_p8.hip one workgroup owns the CU: its two wave rows hand the matrix pipe to each
// other inside the main loop, but prologue (first loads of a tile) and epilogue (LDS transpose, bias / GELU pair / residual
// math, the stores) run on all 8 waves at once with the matrix pipes idle -- at K = 1024 that is 6-15 us of a 22-31 us tile
// (round 3: fc1 forward with the GELU pair 0.33 of the MFMA peak, fc2 dgrad 0.36).  Here two INDEPENDENT workgroups sit on
// each CU (one wave of each per SIMD, 256 VGPRs per lane each, 72 KiB of LDS each): they drift apart by themselves, so one
// workgroup's epilogue VALU / store work and its next tile's first loads run beside the other's main loop.  (lever 2a.)
//
// Pipeline: a stream of half tiles S_0, S_1, ... in the order the MFMAs need them (the 8-phase schedule of gemm_p8.h: per pair of
// K-tiles e, e+1 the phases read A0(e) B0(e) A1(e) B0(e+1) A0(e+1) B1(e+1) A1(e+1) B1(e+2) and multiply the quadrants (0,1) (0,0)
// (1,0) (1,1) (0,0) (0,1) (1,1) (1,0) with the B fragments alternating between two register sets).  An A half tile is 128 rows x 64 k
// = 16 KiB (the 64-row sub-tile h of both wave rows), a B half tile 64 rows = 8 KiB (the 32-column sub-tile h of both wave columns);
// odd stream elements are A halves, even ones B halves.  LDS = a ring of 3 A slots + a ring of 3 B slots (72 KiB):
//   phase g:  ds_read S_{g+1} (landed: guaranteed by phase g-1);  issue the LDS-DMA of S_{g+6} into the slot S_g
//             was read from in phase g-1 (same type, three positions back in that type's ring);  s_waitcnt vmcnt(12) -- leaves
//             S_{g+3} .. S_{g+6} (2 A + 2 B halves = 12 DMA instructions per lane) in flight and guarantees this lane's pieces of
//             S_{g+2};  lgkmcnt(0);  s_barrier;  16 MFMAs.
// One barrier per phase: RAW -- every lane waited for its own pieces of S_{g+2} before barrier g, S_{g+2} is read in phase g+1;
// WAR -- the reads of phase g retire (lgkmcnt(0)) before barrier g, their slot is refilled by DMA issued in phase g+1.
// Five phases (~1.5 us) of look-ahead per half tile; waits are counted, the queue never drains inside a tile.
#include "gemm_p8.h"

namespace {

constexpr int C2_BM = 256, C2_BN = 128, C2_THREADS = 256;
constexpr int C2_ASLOT = 16384, C2_BSLOT = 8192, C2_BOFF = 3 * C2_ASLOT;   // B ring behind the A ring
constexpr int C2_LDS = 3 * C2_ASLOT + 3 * C2_BSLOT;                        // 72 KiB

struct C2Ctx {
    uint32_t aA[2], aB[2];   // ds_read bases (slot 0) of the wave's A / B fragments, k-step 0 / 1 (per lane)
    uint32_t voffA, voffB;   // per-lane byte offset of the DMA source (row in piece, swizzled chunk, K-tile e)
    const char* pA[2][4];    // wave-uniform DMA source rows [half][piece]
    const char* pB[2][2];
    uint32_t mA, mB;         // LDS address of this wave's first DMA piece in slot 0 of the A / B ring
    uint32_t ra, rb, wa, wb;   // byte offsets of the next slot to read / to fill in each ring (wave-uniform)
};

// LDS-DMA pieces (1 KiB each per wave): LDS[l + 16 * lane] <- global[p + voff + OFF]  (OFF is added to the VGPR offset: the
// instruction's immediate offset also moves the LDS destination -- measured, round 4: every result wrong with `offset:128`)
template <int OFF>
__device__ __forceinline__ void c2_glds2(uint32_t voff, const char* p0, const char* p1, uint32_t l0) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %4 \n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %5 \n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff + (uint32_t)OFF), "s"(l0), "s"(l0 + 1024u), "s"(p0), "s"(p1)
        : "memory");
}
template <int OFF>
__device__ __forceinline__ void c2_glds4(uint32_t voff, const char* p0, const char* p1, const char* p2, const char* p3, uint32_t l0) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %6 \n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %7 \n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %8 \n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %9 \n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff + (uint32_t)OFF), "s"(l0), "s"(l0 + 1024u), "s"(l0 + 2048u), "s"(l0 + 3072u), "s"(p0), "s"(p1), "s"(p2), "s"(p3)
        : "memory");
}

template <int IMM>
__device__ __forceinline__ void c2_dsr(u32x4_t& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(IMM));
}
__device__ __forceinline__ uint32_t ring_next(uint32_t off, uint32_t slot) {   // (off + slot) mod (3 * slot), wave-uniform
    const uint32_t n = off + slot;
    return n == 3 * slot ? 0u : n;
}

__device__ __forceinline__ void c2_read_a(C2Ctx& c, u32x4_t (&a)[2][4]) {
    const uint32_t a0 = c.aA[0] + c.ra, a1 = c.aA[1] + c.ra;
    c2_dsr<0 * 2048>(a[0][0], a0); c2_dsr<1 * 2048>(a[0][1], a0); c2_dsr<2 * 2048>(a[0][2], a0); c2_dsr<3 * 2048>(a[0][3], a0);
    c2_dsr<0 * 2048>(a[1][0], a1); c2_dsr<1 * 2048>(a[1][1], a1); c2_dsr<2 * 2048>(a[1][2], a1); c2_dsr<3 * 2048>(a[1][3], a1);
    c.ra = ring_next(c.ra, C2_ASLOT);
}
__device__ __forceinline__ void c2_read_b(C2Ctx& c, u32x4_t (&b)[2][2]) {
    const uint32_t b0 = c.aB[0] + c.rb, b1 = c.aB[1] + c.rb;
    c2_dsr<0 * 2048>(b[0][0], b0); c2_dsr<1 * 2048>(b[0][1], b0);
    c2_dsr<0 * 2048>(b[1][0], b1); c2_dsr<1 * 2048>(b[1][1], b1);
    c.rb = ring_next(c.rb, C2_BSLOT);
}
__device__ __forceinline__ void c2_wait_a(u32x4_t (&a)[2][4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]));
}
__device__ __forceinline__ void c2_wait_b(u32x4_t (&b)[2][2]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]));
}
// issue one half tile: H = half (0 / 1), KT = K-tile relative to the pair's first one
template <int H, int KT>
__device__ __forceinline__ void c2_issue_a(C2Ctx& c) {
    c2_glds4<KT * 128>(c.voffA, c.pA[H][0], c.pA[H][1], c.pA[H][2], c.pA[H][3], c.mA + c.wa);
    c.wa = ring_next(c.wa, C2_ASLOT);
}
template <int H, int KT>
__device__ __forceinline__ void c2_issue_b(C2Ctx& c) {
    c2_glds2<KT * 128>(c.voffB, c.pB[H][0], c.pB[H][1], c.mB + c.wb);
    c.wb = ring_next(c.wb, C2_BSLOT);
}

// phase P of a pair of K-tiles (e, e + 1).  Reads: even phases an A half, odd phases a B half (into b0 in phases 1, 3; b1 in 5, 7).
// Issues S_{g+6}: P0 B1(e+1)  P1 A1(e+1)  P2 B1(e+2)  P3 A0(e+2)  P4 B0(e+2)  P5 A1(e+2)  P6 B0(e+3)  P7 A0(e+3); the LAST pair issues
// only in its first two phases and counts its waits down (10, 6, 4, 0: what is still to come after this lane's pieces of S_{g+2}).
template <int P, bool LAST>
__device__ __forceinline__ void c2_phase(C2Ctx& c, u32x4_t (&a)[2][4], u32x4_t (&b0)[2][2], u32x4_t (&b1)[2][2], f32x4_t (&acc)[4][8]) {
    constexpr bool RA = (P & 1) == 0;
    constexpr bool RB = !RA && !(LAST && P == 7);
    u32x4_t(&br)[2][2] = (P == 1 || P == 3) ? b0 : b1;
    if constexpr (RA) c2_read_a(c, a);
    if constexpr (RB) c2_read_b(c, br);
    if constexpr (!LAST || P < 2) {
        if constexpr (P == 0) c2_issue_b<1, 1>(c);
        if constexpr (P == 1) c2_issue_a<1, 1>(c);
        if constexpr (P == 2) c2_issue_b<1, 2>(c);
        if constexpr (P == 3) c2_issue_a<0, 2>(c);
        if constexpr (P == 4) c2_issue_b<0, 2>(c);
        if constexpr (P == 5) c2_issue_a<1, 2>(c);
        if constexpr (P == 6) c2_issue_b<0, 3>(c);
        if constexpr (P == 7) c2_issue_a<0, 3>(c);
    }
    constexpr int VM = !LAST ? 12 : P < 2 ? 12 : P == 2 ? 10 : P == 3 ? 6 : P == 4 ? 4 : 0;
    wait_vm<VM>();
    if constexpr (RA) c2_wait_a(a);
    if constexpr (RB) c2_wait_b(br);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // quadrant (QI, QJ): rows 64 QI .., columns 32 QJ .. of the wave's block
    constexpr int QI = (P == 2 || P == 3 || P == 6 || P == 7) ? 1 : 0;
    constexpr int QJ = (P == 0 || P == 3 || P == 5 || P == 6) ? 1 : 0;
    u32x4_t(&b)[2][2] = QJ ? b1 : b0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) mma(acc[QJ * 2 + ni][QI * 4 + mi], b[ks][ni], a[ks][mi]);
    __builtin_amdgcn_sched_barrier(0);
}

template <bool LAST>
__device__ __forceinline__ void c2_two_tiles(C2Ctx& c, u32x4_t (&a)[2][4], u32x4_t (&b0)[2][2], u32x4_t (&b1)[2][2], f32x4_t (&acc)[4][8]) {
    c2_phase<0, LAST>(c, a, b0, b1, acc);
    c2_phase<1, LAST>(c, a, b0, b1, acc);
    c2_phase<2, LAST>(c, a, b0, b1, acc);
    c2_phase<3, LAST>(c, a, b0, b1, acc);
    c2_phase<4, LAST>(c, a, b0, b1, acc);
    c2_phase<5, LAST>(c, a, b0, b1, acc);
    c2_phase<6, LAST>(c, a, b0, b1, acc);
    c2_phase<7, LAST>(c, a, b0, b1, acc);
    c.voffA += 256;
    c.voffB += 256;
}

// DMA source rows of this wave for output tile (m0, n0).  A half h, image row R = 32 wave + 8 i + x  <->  tile row
// 128 (R >> 6) + 64 h + (R & 63);  B half h, image row R = 16 wave + 8 i + x  <->  tile column 64 (R >> 5) + 32 h + (R & 31).
// Rows past the matrix edge are clamped to the last complete 8-row piece (their outputs are never stored).
__device__ __forceinline__ void c2_tile_sources(const KArgs& p, C2Ctx& c, int wave, int m0, int n0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int R = 32 * wave + 8 * i;
            int ra = m0 + 128 * (R >> 6) + 64 * h + (R & 63);
            ra = ra < p.M - 8 ? ra : p.M - 8;
            c.pA[h][i] = p.A + (int64_t)ra * p.lda * 2;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int R = 16 * wave + 8 * i;
            int rb = n0 + 64 * (R >> 5) + 32 * h + (R & 31);
            rb = rb < p.N - 8 ? rb : p.N - 8;
            c.pB[h][i] = p.B + (int64_t)rb * p.ldb * 2;
        }
    }
}

template <typename Tout, int EPI, int SP>
__global__ __launch_bounds__(C2_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_c2_kernel(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, g = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);
    const int tiles_m = (p.M + C2_BM - 1) / C2_BM, tiles_n = p.tiles_n, ntiles = tiles_m * tiles_n;
    const int plain = p.order & 1;
    const int pairs = p.k_tiles >> 1;

    C2Ctx c;
    {
        // fragment row fr of a 16-row group, 16-B slot (chunk ^ (row & 7)) with chunk = 4 * kstep + g: k-step 1 flips byte bit 6
        const uint32_t lanepart = (uint32_t)(fr * 128 + ((g ^ (fr & 7)) << 4));
        c.aA[0] = lds0 + wr * 8192 + lanepart;
        c.aA[1] = lds0 + wr * 8192 + (lanepart ^ 64u);
        c.aB[0] = lds0 + C2_BOFF + wc * 4096 + lanepart;
        c.aB[1] = lds0 + C2_BOFF + wc * 4096 + (lanepart ^ 64u);
        c.mA = lds0 + wave * 4096;
        c.mB = lds0 + C2_BOFF + wave * 2048;
    }
    const uint32_t lanesrc = (uint32_t)(((lane & 7) ^ (lane >> 3)) << 4);
    c.voffA = (uint32_t)((lane >> 3) * (int)p.lda * 2) + lanesrc;
    c.voffB = (uint32_t)((lane >> 3) * (int)p.ldb * 2) + lanesrc;

    int tm, tn;
    tile_coords(plain ? (int)blockIdx.x : xcd_remap((int)blockIdx.x, ntiles), tiles_m, tiles_n, plain, tm, tn);
    const int m0 = tm * C2_BM, n0 = tn * C2_BN;
    c2_tile_sources(p, c, wave, m0, n0);
    // prologue: S_0 .. S_5 = B1(0) A0(0) B0(0) A1(0) B0(1) A0(1) fill the two rings
    c.wa = 0; c.wb = 0;
    c2_issue_b<1, 0>(c); c2_issue_a<0, 0>(c); c2_issue_b<0, 0>(c); c2_issue_a<1, 0>(c); c2_issue_b<0, 1>(c); c2_issue_a<0, 1>(c);
    c.ra = 0; c.rb = 0;      // (wa = wb = 0 again: three issues each)

    u32x4_t a[2][4], b0[2][2], b1[2][2];
    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    wait_vm<12>();   // S_0, S_1 have landed (this lane's pieces): 18 issued, B 2 + A 4 retired
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    c2_read_b(c, b1);
    c2_wait_b(b1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();   // every wave has read S_0: phase 0 refills its slot
    __builtin_amdgcn_sched_barrier(0);

    for (int it = 0; it < pairs - 1; ++it) c2_two_tiles<false>(c, a, b0, b1, acc);
    c2_two_tiles<true>(c, a, b0, b1, acc);

    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();   // every wave is past its last ds_read: the rings become the epilogue's transpose space
    __builtin_amdgcn_sched_barrier(0);
    epilogue_lds16<Tout, EPI, 128, SP>(p, acc, smem + wave * 4096, m0 + wr * 128, n0 + wc * 64, lane);
}

template <typename Tout, int EPI, int SP>
int launch_c2_kernel(const KArgs& a, int ntiles, hipStream_t stream) {
    static unsigned long long optin = 0;
    if (const int e = mtp_optin_lds((const void*)gemm_nt_c2_kernel<Tout, EPI, SP>, C2_LDS, optin)) return e;
    hipLaunchKernelGGL((gemm_nt_c2_kernel<Tout, EPI, SP>), dim3(ntiles), dim3(C2_THREADS), C2_LDS, stream, a);
    return mtp_launch_status();
}

template <typename Tout, int EPI>
int launch_c2(const KArgs& k, int flags, hipStream_t stream) {
    KArgs a = k;
    a.tiles_n = (k.N + C2_BN - 1) / C2_BN;
    a.k_tiles = k.K / 64;
    a.order = (flags >> 1) & 1;
    a.atomic_out = 0;
    const int ntiles = ((k.M + C2_BM - 1) / C2_BM) * a.tiles_n;
    // store policy of the epilogue as in gemm_p8.hip: nt for the bf16 outputs, sc1 (write-through) for the f32 residual form
    int sp = (flags >> 13) & 3;
    if (sp == 0) sp = (EPI == MTP_EPI_BIAS_RES) ? 2 : 1;
    if (sp == 1) return launch_c2_kernel<Tout, EPI, 1>(a, ntiles, stream);
    if (sp == 2) return launch_c2_kernel<Tout, EPI, 2>(a, ntiles, stream);
    return launch_c2_kernel<Tout, EPI, 0>(a, ntiles, stream);
}

}  // namespace

// same preconditions as the 8-wave kernel (whole K-tile pairs, 8-row DMA pieces, 32-bit DMA offsets); flags: bit1 = plain tile
// order, bits 13-14 = store policy
int mtp_nt_c2_launch(const KArgs& k, int out_dtype, int epi, int flags, hipStream_t stream) {
    if (!mtp_nt_p8_fits(k, out_dtype, epi)) return MTP_ERR_UNSUPPORTED;
    if (epi == MTP_EPI_BIAS_RES) return launch_c2<float, MTP_EPI_BIAS_RES>(k, flags, stream);
    if (out_dtype == MTP_BF16) {
        switch (epi) {
            case MTP_EPI_BIAS: return launch_c2<bf16_t, MTP_EPI_BIAS>(k, flags, stream);
            case MTP_EPI_BIAS_GELU: return launch_c2<bf16_t, MTP_EPI_BIAS_GELU>(k, flags, stream);
            case MTP_EPI_DGELU: return launch_c2<bf16_t, MTP_EPI_DGELU>(k, flags, stream);
            case MTP_EPI_BIAS_GELU_DG: return launch_c2<bf16_t, MTP_EPI_BIAS_GELU_DG>(k, flags, stream);
            case MTP_EPI_MUL: return launch_c2<bf16_t, MTP_EPI_MUL>(k, flags, stream);
            default: return MTP_ERR_UNSUPPORTED;
        }
    }
    return launch_c2<float, MTP_EPI_BIAS>(k, flags, stream);
}
