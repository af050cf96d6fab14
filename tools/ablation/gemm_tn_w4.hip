// Grouped weight-gradient GEMM, second form (round 3):  dW_j[m][n] = sum_t dY_j[t][m] * X_j[t][n]  (nn.Linear backward w.r.t. the
// weight, VIT:50-52, 78, 87; bf16 in, f32 out), 256 x 256 output tile per workgroup, the whole contraction inside the workgroup --
// same contract as gemm_tn_p8.hip (mtp_gemm_tn_grouped), different machine mapping:
//
//   FOUR waves, ONE per SIMD, each owning a 128 x 128 quadrant as 4 x 4 tiles of v_mfma_f32_32x32x16_bf16 (256 accumulator registers
//   of the wave's 512).
//
// Why: both operands are token-major, so every MFMA fragment comes out of the transpose read ds_read_b64_tr_b16 (8 bytes per lane:
// two instructions per fragment).  With 8 waves of 128 x 64 (gemm_tn_p8.hip) a K-tile costs a wave 48 DS instructions + 4 LDS-DMA
// instructions next to 64 MFMAs of 16 passes -- the "read" half of a phase (16 DS x ~17 cycles + 2 DMA x ~150) is LONGER than the
// partner wave's 16 MFMAs (272 cycles), and the matrix pipe idles 40 % of the time (SQ_VALU_MFMA_BUSY 60 %, profiles/r02_pmc_sq_p8_kernels.txt).
// A 128 x 128 quadrant needs 64 DS instructions for 64 MFMAs of 32 passes: ONE transpose read per 32-cycle MFMA, which fits the
// ~28 issue cycles a single wave has free under each MFMA (MI355X guide: <= 5 single-issue instructions per 32x32x16 gap), and the
// instruction rate of 32x32x16 (32 cycles for twice the flops of a 16x16x32 at 17) is 6 % better on top.
//
// LDS: 2 K-tile buffers of 64 KiB = A tile [64 t][256 m] | B tile [64 t][256 n] (512-byte rows, untransposed; the 32-byte slot of a row
// XOR-ed by 2 (t & 3) on the per-lane SOURCE address: the 4 rows x 2 adjacent slots a 32-lane half reads in one transpose read then
// cover all 64 banks).  One barrier per K-tile.
// Staging goes through REGISTERS, not LDS-DMA: a wave alone on its SIMD pays every cycle of an instruction's issue time with matrix-pipe
// idle time, and an LDS-DMA piece costs 60-185 issue cycles (MI355X guide) -- 16 of them per K-tile made the first version of this
// kernel slower than the 8-wave one (1182 vs 1038 us per 4 blocks, same box).  global_load_dwordx4 + ds_write_b128 cost ~8 + 13
// cycles and fit under the 32-cycle MFMAs; the 64 staging VGPRs are free here (a 512-register wave).  Tile k + 1 is loaded during
// k-step 0 of tile k and written to the other buffer during k-steps 2 and 3.
// Bias gradient (column sums of dY): the workgroups of a tile row share it -- workgroup (tm, tn) sums the K-tiles kt == tn (mod tiles_n)
// out of LDS and adds its partial with one f32 atomic per column (the p8 form gave it all to column 0: +15 % on a quarter of the tiles).
// Accumulation order differs from the 16x16x32 kernels (16 k per MFMA instead of 32): results agree to f32 rounding, not bit for bit.
#include "gemm_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef short w4_tr4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char w4_lds_t;

constexpr int W4_THREADS = 256, W4_BM = 256, W4_BN = 256;
constexpr int W4_BUF = 65536, W4_BOFF = 32768, W4_LDS = 2 * W4_BUF;
constexpr int W4_MAX_PROBLEMS = MTP_MAX_GROUPED_GEMMS;

struct W4Prob {
    const char* A;      // dY (Kc, M) bf16, lda
    const char* B;      // X  (Kc, N) bf16, ldb
    float* C;           // dW (M, N) f32, ldc
    float* colsum;      // += column sums of A (M entries) or nullptr
    int M, N, K;
    int lda, ldb, ldc;
    int tile0, tiles_m, tiles_n;
    int pad_;
};
struct W4Group {
    W4Prob p[W4_MAX_PROBLEMS];
    int nprob, ntiles, plain;
};

// MFMA operand (32 columns x 16 t) out of an untransposed tile: two transpose reads, 4 t each
__device__ __forceinline__ bf16x8_t w4_frag(w4_lds_t* s) {
    const w4_tr4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w4_tr4_t*)(s));
    const w4_tr4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w4_tr4_t*)(s + 2048));
    const uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return __builtin_bit_cast(bf16x8_t, make_uint4(l.x, l.y, h.x, h.y));
}

// Sources as buffer loads: resource (matrix base) in SGPRs + a persistent 32-bit per-lane offset + a scalar offset for (K-tile, piece).
// (With 64-bit flat addresses hipcc recomputed an address pair per load into registers that earlier loads were still landing in, and
//  waited for those loads first: the 16 loads of a tile went out one latency after the other.)
struct W4Src {
    __amdgpu_buffer_rsrc_t rA, rB;
    uint32_t voffA[2], voffB[2];   // per-lane source offsets for even / odd pieces
    uint32_t soffA, soffB;         // scalar byte offset of this wave's first piece (rows 16 wave ..) of the NEXT K-tile to load
    uint32_t stepA, stepB;         // bytes per piece step (2 rows); a K-tile = 32 piece steps
};
typedef __attribute__((ext_vector_type(4))) uint32_t w4_u32x4_t;
// staged piece s (0-7: A piece s, 8-15: B piece s - 8 of this wave): 16 bytes per lane
__device__ __forceinline__ w4_u32x4_t w4_load(const W4Src& d, int s) {
    const int i = s & 7;
    return s < 8 ? __builtin_amdgcn_raw_buffer_load_b128(d.rA, d.voffA[i & 1], d.soffA + i * d.stepA, 0)
                 : __builtin_amdgcn_raw_buffer_load_b128(d.rB, d.voffB[i & 1], d.soffB + i * d.stepB, 0);
}
__device__ __forceinline__ void w4_store(w4_lds_t* tile, int wave, int lane, int s, const w4_u32x4_t& v) {   // tile: the buffer's A tile
    w4_lds_t* dst = tile + (s < 8 ? 0 : W4_BOFF) + (8 * wave + (s & 7)) * 1024 + lane * 16;
    *reinterpret_cast<__attribute__((address_space(3))) w4_u32x4_t*>(dst) = v;
}

__global__ __launch_bounds__(W4_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_tn_w4_kernel(W4Group grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    w4_lds_t* const sm = (w4_lds_t*)smem;

    // ---- which problem, which tile
    const int vt = grp.plain ? (int)blockIdx.x : xcd_remap(blockIdx.x, grp.ntiles);
    int pi = 0;
#pragma unroll 1
    while (pi + 1 < grp.nprob && vt >= grp.p[pi + 1].tile0) ++pi;
    const W4Prob& q = grp.p[pi];
    int tm, tn;
    {
        const int t = vt - q.tile0;
        if (grp.plain) {
            tm = t / q.tiles_n; tn = t - tm * q.tiles_n;
        } else {   // panels of 8 tile rows, rows fastest (as tile_coords of gemm_p8.h)
            const int per = 8 * q.tiles_n, g8 = t / per, r = t - g8 * per;
            const int gm = (q.tiles_m - g8 * 8) < 8 ? (q.tiles_m - g8 * 8) : 8;
            tn = r / gm;
            tm = g8 * 8 + (r - tn * gm);
        }
    }
    const int m0 = tm * W4_BM, n0 = tn * W4_BN;
    const int nk = q.K >> 6;

    // ---- DMA addressing: piece p = 8 wave + i holds tile rows 2p, 2p + 1; lane l -> row r = l >> 5, 16-byte unit u = l & 31 of the row;
    // the unit's 32-byte slot u >> 1 holds the source slot (u >> 1) ^ (2 (t & 3)), t & 3 = 2 (i & 1) + r
    W4Src d;
    {
        const int r = lane >> 5, u = lane & 31;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int t3 = 2 * par + r;
            const int chunk = ((((u >> 1) ^ (t3 << 1)) << 1) | (u & 1));
            d.voffA[par] = (uint32_t)(r * q.lda * 2 + chunk * 16);
            d.voffB[par] = (uint32_t)(r * q.ldb * 2 + chunk * 16);
        }
        d.stepA = (uint32_t)q.lda * 4u;
        d.stepB = (uint32_t)q.ldb * 4u;
        d.rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(q.A), 0, -1, 0x00020000);
        d.rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(q.B), 0, -1, 0x00020000);
        d.soffA = ((uint32_t)(16 * wave) * (uint32_t)q.lda + (uint32_t)m0) * 2u;       // (host: K * ld * 2 < 2^32)
        d.soffB = ((uint32_t)(16 * wave) * (uint32_t)q.ldb + (uint32_t)n0) * 2u;
    }
    const uint32_t ktA = (uint32_t)q.lda * 128u, ktB = (uint32_t)q.ldb * 128u;      // bytes per K-tile (64 rows)

    // ---- fragment addressing (see the header): lane = 16 j + i supplies row 8 (j >> 1) + (i >> 2) (+ 4: second read), 8 bytes at column
    // 16 (j & 1) + 4 (i & 3) of its 32-column tile; the row's swizzle is 2 (i >> 2)
    uint32_t offA[4], offB[4];
    {
        const int i = lane & 15, j = lane >> 4;
        const uint32_t rowpart = (uint32_t)((8 * (j >> 1) + (i >> 2)) * 512 + 8 * (i & 3));
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            offA[x] = rowpart + (uint32_t)((8 * wr + 2 * (x ^ (i >> 2)) + (j & 1)) * 32);
            offB[x] = W4_BOFF + rowpart + (uint32_t)((8 * wc + 2 * (x ^ (i >> 2)) + (j & 1)) * 32);
        }
    }

    // ---- bias gradient share of this workgroup: K-tiles kt == tn (mod tiles_n); thread = 16-byte unit u of rows t == rsel (mod 4)
    const bool do_cs = q.colsum != nullptr;
    const int cs_u = tid & 31, cs_rsel = (tid >> 5) & 3, cs_half = tid >> 7;
    float cs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = 0.f;

    f32x16_t acc[4][4];      // [n tile][m tile]: D^T, so that a lane ends with 4 consecutive n of one row m
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // ---- prologue: K-tile 0 -> buffer 0
    w4_u32x4_t st[16];
#pragma unroll
    for (int x = 0; x < 16; ++x) st[x] = w4_load(d, x);
#pragma unroll
    for (int x = 0; x < 16; ++x) w4_store(sm, wave, lane, x, st[x]);
    d.soffA += ktA; d.soffB += ktB;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        w4_lds_t* const sb = sm + buf * W4_BUF;
        const bool more = kt + 1 < nk;
        bf16x8_t fa[2][4], fb[2][4];      // [register set][tile]: k-step s uses set s & 1, the next step's fragments load into the other
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            fa[0][x] = w4_frag(sb + offA[x]);
            fb[0][x] = w4_frag(sb + offB[x]);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
#pragma unroll
            for (int g = 0; g < 4; ++g) {          // group g: the 4 MFMAs of m tile g; two next-step fragments and (k-steps 0, 1) two DMA pieces
                if (ks < 3) {      // what the next step needs first is read first: all four B fragments (groups 0, 1), then the A fragments
                    if (g < 2) {
                        fb[nxt][2 * g] = w4_frag(sb + offB[2 * g] + (ks + 1) * 8192);
                        fb[nxt][2 * g + 1] = w4_frag(sb + offB[2 * g + 1] + (ks + 1) * 8192);
                    } else {
                        fa[nxt][2 * g - 4] = w4_frag(sb + offA[2 * g - 4] + (ks + 1) * 8192);
                        fa[nxt][2 * g - 3] = w4_frag(sb + offA[2 * g - 3] + (ks + 1) * 8192);
                    }
                }
                if (more) {        // tile kt + 1: 16 loads under k-step 0, 16 LDS stores (other buffer: free since the last barrier) under k-steps 2, 3
                    if (ks == 0) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) st[4 * g + x] = w4_load(d, 4 * g + x);
                    }
                    if (ks >= 2) {
#pragma unroll
                        for (int x = 0; x < 2; ++x) w4_store(sm + (buf ^ 1) * W4_BUF, wave, lane, 8 * (ks - 2) + 2 * g + x, st[8 * (ks - 2) + 2 * g + x]);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[cur][nt], fa[cur][g], acc[nt][g], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (do_cs && (kt % q.tiles_n) == tn) {      // (workgroup-uniform) this K-tile's share of the bias gradient, out of the A tile in LDS
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t = 4 * (8 * cs_half + i) + cs_rsel;
                typedef __attribute__((address_space(3))) const w4_u32x4_t w4_lds_u32x4_t;
                const w4_u32x4_t w = *reinterpret_cast<w4_lds_u32x4_t*>(sb + t * 512 + cs_u * 16);
                const uint32_t ww[4] = {w[0], w[1], w[2], w[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cs[2 * e] += __uint_as_float(ww[e] << 16);
                    cs[2 * e + 1] += __uint_as_float(ww[e] & 0xffff0000u);
                }
            }
        }
        d.soffA += ktA; d.soffB += ktB;
        // this wave's pieces of tile kt + 1 are in LDS, its reads of buffer `buf` are retired; then everybody's
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- bias gradient: 8 threads hold partial sums of the same 8 columns -> LDS (f32 atomics, 8 per thread, once) -> one global atomic per column
    if (do_cs) {
        float* red = reinterpret_cast<float*>(smem);
        for (int i = tid; i < W4_BM; i += W4_THREADS) red[i] = 0.f;
        __syncthreads();
        const int col0 = (((cs_u >> 1) ^ (cs_rsel << 1)) << 4) + (cs_u & 1) * 8;      // features of this thread's 16-byte unit
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(red + col0 + e, cs[e]);
        __syncthreads();
        for (int i = tid; i < W4_BM; i += W4_THREADS) atomicAdd(q.colsum + m0 + i, red[i]);
    }

    // ---- epilogue: lane holds, for m = 32 mt + (lane & 31), the columns n = 32 nt + 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3): 16-byte stores
    {
        const int ml = lane & 31, nh = (lane >> 5) * 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float* row = q.C + (int64_t)(m0 + 128 * wr + 32 * mt + ml) * q.ldc + n0 + 128 * wc + nh;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    *reinterpret_cast<float4*>(row + 32 * nt + 8 * r4) =
                        make_float4(acc[nt][mt][4 * r4], acc[nt][mt][4 * r4 + 1], acc[nt][mt][4 * r4 + 2], acc[nt][mt][4 * r4 + 3]);
        }
    }
}

}  // namespace

// same contract as mtp_gemm_tn_grouped (gemm_tn_p8.hip), which dispatches here (variant bit 5 of the first problem keeps the 8-wave form)
int mtp_gemm_tn_grouped_w4(const mtp_gemm_args* args, int count, hipStream_t stream) {
    W4Group g = {};
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        const mtp_gemm_args& a = args[i];
        W4Prob& q = g.p[i];
        q.A = (const char*)a.A; q.B = (const char*)a.B; q.C = (float*)a.C; q.colsum = a.colsum;
        q.M = (int)a.M; q.N = (int)a.N; q.K = (int)a.K;
        q.lda = (int)a.lda; q.ldb = (int)a.ldb; q.ldc = (int)a.ldc;
        q.tiles_m = q.M / W4_BM; q.tiles_n = q.N / W4_BN;
        q.tile0 = tiles;
        tiles += q.tiles_m * q.tiles_n;
    }
    g.nprob = count;
    g.ntiles = tiles;
    g.plain = (args[0].variant >> 1) & 1;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_tn_w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    hipLaunchKernelGGL(gemm_tn_w4_kernel, dim3(tiles), dim3(W4_THREADS), W4_LDS, stream, g);
    return mtp_launch_status();
}
