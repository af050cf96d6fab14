// Grouped weight-gradient GEMM for gfx950 on the 8-phase pipeline of gemm_p8.h:  dW_j[m][n] = sum_t dY_j[t][m] * X_j[t][n]
// (nn.Linear backward w.r.t. the weight, VIT:50-52, 78, 87; f32 out) for a LIST of problems in one launch, 256 x 256 tiles,
// the whole contraction (T = B*196 tokens) inside one workgroup.
//
// Why grouped: one such GEMM has 16 .. 64 tiles of 256 x 256 -- far fewer than the 256 CUs.  The 128-wide kernels of gemm.hip
// therefore split the contraction 4-16 ways and pay for it: 64 MB of f32 partial tiles written and re-read per GEMM plus a
// reduction launch (0.71 of ~1.0 PF/s in the step, 11.6 ms).  The four weight gradients of a transformer block do not depend
// on each other, only on tensors the backward pass keeps anyway (dY of the layer and its saved input), so the host defers them
// and launches the weight gradients of FOUR blocks together: 4 x (48 + 16 + 64 + 64) = 768 tiles = exactly three rounds of one
// workgroup per CU, each tile accumulating all 196 K-tiles in registers and storing its f32 result once.  No split-K, no
// partials, no reduction kernel, no quantisation loss.
//
// Operands are token-major as they lie in memory ([t][feature], feature contiguous): both go HBM -> LDS untransposed by LDS-DMA
// and the MFMA fragments (8 consecutive t per lane) come out of the gfx950 transpose read ds_read_b64_tr_b16 -- the layout of
// gemm_tn_tr_kernel (gemm.hip; probed with tools/probes/tr_probe.hip): half tile = [64 t][128 features] (256-B rows), 32-B slot pair
// (16 features) ^= f(t), f = (t & 3) | ((t >> 3) & 1) << 2, applied on the per-lane DMA SOURCE address.  A-half h holds the
// feature columns "sub-tile h" of the two wave rows, B-half j the columns "sub-tile j" of the four wave columns, as in the NT
// kernel, so the phase schedule is shared -- but here a half tile is 128 CONSECUTIVE columns and a wave's sub-tiles are strips
// 128 columns apart (whole 256-B runs per DMA row).
// Bias gradient (column sums of dY) as a by-product: the workgroups of tile column 0 also add up the dY half tiles out of LDS.
#include "gemm_p8.h"

namespace {

typedef short tr_v4s_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char_t;
constexpr int TN_MAX_PROBLEMS = MTP_MAX_GROUPED_GEMMS;

struct TnProb {
    const char* A;      // dY (Kc, M) bf16, lda
    const char* B;      // X  (Kc, N) bf16, ldb
    float* C;           // dW (M, N) f32, ldc
    float* colsum;      // += column sums of A (M entries) or nullptr
    float* sqn;         // += sum of the squares of this problem's output (unsplit problems; the gradient norm of the clipping step) or nullptr
    int M, N, K;
    int lda, ldb, ldc;
    int tile0, tiles_m, tiles_n;
    int splits;         // the contraction in `splits` pieces of `kchunk` rows (the last one shorter), each an own tile writing its own (M, ldc)
    int kchunk;         //   f32 image at C + piece * M * ldc: few-tile problems with a long contraction (M, N <= 256, K = 131072: InternImage level 0)
    int pad_;
};
struct TnGroup {
    TnProb p[TN_MAX_PROBLEMS];
    int nprob, ntiles, plain;
};

struct T8Ctx {
    lds_char_t* smem;        // LDS base (the transpose reads go through the builtin: compiler-visible LDS loads)
    uint32_t offA[2][4];     // per-lane byte offset of fragment mi (k-step 0, rows 0-3) in buffer b
    uint32_t offB[2][2];
    uint32_t voffA[2], voffB[2];   // per-lane DMA source offsets of half 0 / 1 (row in piece, swizzled 16-B chunk, current k)
    uint32_t kstepA, kstepB; // bytes per K-tile: 64 rows
    const char* pA[2][2];    // wave-uniform DMA source bases [half][piece]
    const char* pB[2][2];
    uint32_t m0base;         // LDS address of this wave's first DMA piece in slot 0
    bool colsum;             // workgroup-uniform: also sum the A half tiles over t
    uint32_t csoff;          // per-lane byte offset of the column-sum reads inside a half tile
    float cs[2][8];          // [half][column of the lane's 16-B chunk]
};

__device__ __forceinline__ u32x4_t tr_frag8(lds_char_t* s) {
    const tr_v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s_t*)(s));
    const tr_v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s_t*)(s + 1024));
    const uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return u32x4_t{l.x, l.y, h.x, h.y};
}

struct TnOps {
    typedef T8Ctx Ctx;
    static constexpr int kLoadsPerPiecePair = 2;
    static constexpr int kMi1 = 4;
    template <int K, int BUF>
    static __device__ __forceinline__ void read_a(Ctx& c, u32x4_t (&a)[2][4]) {
        lds_char_t* s = c.smem + K * P8_HALF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a[ks][mi] = tr_frag8(s + c.offA[BUF][mi] + ks * 8192);
    }
    template <int K, int BUF>
    static __device__ __forceinline__ void read_b(Ctx& c, u32x4_t (&b)[2][2]) {
        lds_char_t* s = c.smem + K * P8_HALF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ks][ni] = tr_frag8(s + c.offB[BUF][ni] + ks * 8192);
    }
    // one fragment = two transpose reads (read-ahead form of the phase, gemm_p8.h)
    template <int K, int BUF, int KS, int MI>
    static __device__ __forceinline__ void read_a_frag(Ctx& c, u32x4_t (&a)[2][4]) { a[KS][MI] = tr_frag8(c.smem + K * P8_HALF + c.offA[BUF][MI] + KS * 8192); }
    template <int K, int BUF, int KS, int NI>
    static __device__ __forceinline__ void read_b_frag(Ctx& c, u32x4_t (&b)[2][2]) { b[KS][NI] = tr_frag8(c.smem + K * P8_HALF + c.offB[BUF][NI] + KS * 8192); }
    template <int SK, int SBUF>
    static __device__ __forceinline__ void stage(Ctx& c) {
        constexpr int h = (SK == KA1 || SK == KB1) ? 1 : 0;
        const uint32_t l0 = c.m0base + SBUF * P8_BUF + SK * P8_HALF;
        if constexpr (SK == KA0 || SK == KA1)
            glds2(c.voffA[h], c.pA[h][0], c.pA[h][1], l0, l0 + 1024);
        else
            glds2(c.voffB[h], c.pB[h][0], c.pB[h][1], l0, l0 + 1024);
    }
    // the fragment reads are compiler-visible LDS loads: the asm wait below (a memory barrier for the compiler) retires them
    // before the phase's barrier, which is what lets the slot be refilled one phase later (WAR rule of gemm_p8.hip)
    template <int K, int BUF>
    static __device__ __forceinline__ void retire_a(Ctx& c, u32x4_t (&)[2][4]) {
        if (c.colsum) {   // rows r and r + 32 of the half tile share the swizzle, i.e. the lane's 16-B piece is the same 8 columns
            constexpr int h = K == KA1 ? 1 : 0;
            typedef __attribute__((address_space(3))) const u32x4_t lds_u32x4_t;
            const u32x4_t w0 = *reinterpret_cast<lds_u32x4_t*>(c.smem + BUF * P8_BUF + K * P8_HALF + c.csoff);
            const u32x4_t w1 = *reinterpret_cast<lds_u32x4_t*>(c.smem + BUF * P8_BUF + K * P8_HALF + c.csoff + 8192);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                c.cs[h][2 * e] += __uint_as_float(w0[e] << 16) + __uint_as_float(w1[e] << 16);
                c.cs[h][2 * e + 1] += __uint_as_float(w0[e] & 0xffff0000u) + __uint_as_float(w1[e] & 0xffff0000u);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    static __device__ __forceinline__ void retire_b(u32x4_t (&)[2][2]) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    static __device__ __forceinline__ void next_ktile(Ctx& c) {
        c.voffA[0] += c.kstepA; c.voffA[1] += c.kstepA;
        c.voffB[0] += c.kstepB; c.voffB[1] += c.kstepB;
    }
};

template <int XP, int RA = 1>
__global__ __launch_bounds__(P8_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn_p8_kernel(TnGroup grp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(smem);

    // ---- which problem, which tile
    const int vt = grp.plain ? (int)blockIdx.x : xcd_remap(blockIdx.x, grp.ntiles);
    int pi = 0;
#pragma unroll 1
    while (pi + 1 < grp.nprob && vt >= grp.p[pi + 1].tile0) ++pi;
    const TnProb& q = grp.p[pi];
    int tm, tn;
    const int per = q.tiles_m * q.tiles_n, piece = (vt - q.tile0) / per;
    tile_coords(vt - q.tile0 - piece * per, q.tiles_m, q.tiles_n, grp.plain, tm, tn);
    const int m0 = tm * P8_BM, n0 = tn * P8_BN;
    const int k0 = piece * q.kchunk, klen = q.K - k0 < q.kchunk ? q.K - k0 : q.kchunk;
    const int pairs = klen >> 7;

    T8Ctx c;
    c.smem = (lds_char_t*)smem;
    {
        const int i = lane & 15, g = lane >> 4;
        const int fr_ = (i >> 2) | ((g & 1) << 2);                                   // f(t) of the rows this lane addresses
        const uint32_t rowoff = (uint32_t)((8 * g + (i >> 2)) * 256 + (i & 3) * 8);   // row 8g + i/4 (+4: second read), columns 4(i&3)..
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) c.offA[b][mi] = b * P8_BUF + rowoff + (uint32_t)((((wr * 4 + mi) ^ fr_)) << 5);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) c.offB[b][ni] = b * P8_BUF + rowoff + (uint32_t)((((wc * 2 + ni) ^ fr_)) << 5);
        }
        c.m0base = lds0 + wave * 2048;
        // DMA: piece (wave, i) = half-tile rows wave*8 + i*4 + (lane >> 4); LDS 16-B position lane & 15 receives global chunk
        // (lane & 15) ^ (f << 1) with f = f(row) = (lane >> 4) | (wave & 1) << 2
        const int fd = (lane >> 4) | ((wave & 1) << 2);
        const int ch = (lane & 15) ^ (fd << 1);
        // a half tile = 128 CONSECUTIVE feature columns (256-B runs: whole cache lines for every DMA row): A-half h = tile
        // columns [128h, 128h + 128), of which wave row wr multiplies [64 wr, 64 wr + 64); B-half j likewise with 32 columns per
        // wave column.  (Splitting the wave's columns in two strips instead of splitting each strip's lines in two: measured
        // 871 us of a 1050-us launch were the DMA / read stream alone when B-half rows were 4 x 64-B pieces.)
        // Edge tiles (M, N multiples of 8, not of 256): a 16-byte chunk past the last column is fetched from the half's first column instead (a
        // valid address; its products land in accumulators the epilogue never stores); a half that starts past the edge reads half 0's columns.
        const int cbA[2] = {m0, m0 + 128 < q.M ? m0 + 128 : m0}, cbB[2] = {n0, n0 + 128 < q.N ? n0 + 128 : n0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            c.voffA[h] = (uint32_t)((lane >> 4) * q.lda * 2 + (cbA[h] + ch * 8 < q.M ? ch * 16 : 0));
            c.voffB[h] = (uint32_t)((lane >> 4) * q.ldb * 2 + (cbB[h] + ch * 8 < q.N ? ch * 16 : 0));
        }
        c.kstepA = (uint32_t)(64 * q.lda * 2);
        c.kstepB = (uint32_t)(64 * q.ldb * 2);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
                c.pA[h][pc] = q.A + ((int64_t)(k0 + wave * 8 + pc * 4) * q.lda + cbA[h]) * 2;
                c.pB[h][pc] = q.B + ((int64_t)(k0 + wave * 8 + pc * 4) * q.ldb + cbB[h]) * 2;
            }
        c.colsum = q.colsum != nullptr && tn == 0;
        c.csoff = (uint32_t)((tid >> 4) * 256 + (tid & 15) * 16);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) c.cs[h][e] = 0.f;
    }

    u32x4_t a[2][4], b0[2][2], b1[2][2];
    f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: S_0 .. S_7 = B1 A0 B0 A1 of K-tile 0 (buffer 0), B0 A0 B1 A1 of K-tile 1 (buffer 1)
    TnOps::stage<KB1, 0>(c); TnOps::stage<KA0, 0>(c); TnOps::stage<KB0, 0>(c); TnOps::stage<KA1, 0>(c);
    TnOps::next_ktile(c);
    TnOps::stage<KB0, 1>(c); TnOps::stage<KA0, 1>(c); TnOps::stage<KB1, 1>(c); TnOps::stage<KA1, 1>(c);
    TnOps::next_ktile(c);
    if constexpr (RA) wait_vm<10>(); else wait_vm<12>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    TnOps::read_b<KB1, 0>(c, b1);
    if constexpr (RA) TnOps::read_a<KA0, 0>(c, a);     // read-ahead form: the first phase's A half too (retired, with its column sums, in that phase)
    TnOps::retire_b(b1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (!(XP & 2) && wr == 1) __builtin_amdgcn_s_barrier();   // wave row 1 runs one barrier behind wave row 0
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (RA) {
        for (int it = 0; it < pairs - 1; ++it) two_tiles_ra<TnOps, false>(c, a, b0, b1, acc);
        two_tiles_ra<TnOps, true>(c, a, b0, b1, acc);
    } else {
        for (int it = 0; it < pairs - 1; ++it) two_tiles<TnOps, false, XP>(c, a, b0, b1, acc);
        two_tiles<TnOps, true, XP>(c, a, b0, b1, acc);
    }

    __builtin_amdgcn_sched_barrier(0);
    if (!(XP & 2) && wr == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    if (c.colsum) {
        // 32 row groups (tid >> 4) hold partial sums of the same 8 columns: reduce through LDS, one atomic per column
        float* red = reinterpret_cast<float*>(smem);
        const int rg = tid >> 4;
        const int f = (rg & 3) | ((rg >> 1) & 4);
        const int ch = (tid & 15) ^ (f << 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4*>(red + (h * 32 + rg) * 128 + ch * 8) = make_float4(c.cs[h][0], c.cs[h][1], c.cs[h][2], c.cs[h][3]);
            *reinterpret_cast<float4*>(red + (h * 32 + rg) * 128 + ch * 8 + 4) = make_float4(c.cs[h][4], c.cs[h][5], c.cs[h][6], c.cs[h][7]);
        }
        __syncthreads();
        if (tid < 256) {
            const int h = tid >> 7, x = tid & 127;     // x = chunk * 8 + e inside the half tile image
            float s = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) s += red[(h * 32 + r) * 128 + x];
            if (m0 + h * 128 + x < q.M) atomicAdd(q.colsum + m0 + h * 128 + x, s);
        }
        __syncthreads();
    }

    if (q.sqn) {
        // squared-norm by-product (round 6): the sum of dW^2 over this tile straight out of the accumulators -- the clipping step then needs no pass over the 97 % of
        // the gradient buffer these launches write.  acc[nf][4 h + mfl][e] = tile row 64 wr + 128 h + 16 mfl + fr, tile column 32 wc + 128 (c >> 5) + (c & 31) with
        // c = 16 nf + 4 g + e (the mapping of epilogue_lds<.., 128, 128>); edge tiles mask what lies beyond (M, N)
        const int fr = lane & 15, gq = lane >> 4;
        const bool full = m0 + P8_BM <= q.M && n0 + P8_BN <= q.N;
        float ss = 0.f;
#pragma unroll
        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
            for (int hm = 0; hm < 8; ++hm)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[nf][hm][e];
                    if (full) {
                        ss += v * v;
                    } else {
                        const int row = m0 + wr * 64 + (hm >> 2) * 128 + (hm & 3) * 16 + fr;
                        const int cc = nf * 16 + gq * 4 + e, col = n0 + wc * 32 + (cc >> 5) * 128 + (cc & 31);
                        ss += (row < q.M && col < q.N) ? v * v : 0.f;
                    }
                }
        ss = wave_sum(ss);
        if (lane == 0) atomicAdd(q.sqn, ss);
    }

    KArgs o = {};
    o.C = reinterpret_cast<char*>(q.C + (int64_t)piece * q.M * q.ldc);
    o.M = q.M; o.N = q.N; o.ldc = q.ldc;
    // accumulator rows 4i .. 4i+3 = tile rows 128 i + 64 wr + ..., accumulator columns 2j, 2j+1 = tile columns 128 j + 32 wc + ...
    epilogue_lds<float, MTP_EPI_BIAS, 128, 128>(o, acc, smem + wave * P8_HALF, m0 + wr * 64, n0 + wc * 32, lane);
}

}  // namespace

// Every problem: bf16 operands, f32 output, M and N multiples of 8 (edge tiles of the 256 x 256 grid are clamped / masked), contraction a
// multiple of 128, 16-byte aligned rows.  split_k > 1 with `aux` (f32, split_k * M * ldc elements): the contraction is cut into split_k pieces
// (multiples of 128 rows), every (tile, piece) is an own workgroup writing its own image into aux, and -- unless defer_sum -- one
// mtp_sum_partials_batch launch per 12 such problems adds the images up into C: for problems of a few tiles with a very long contraction.
extern "C" int mtp_gemm_tn_grouped(const mtp_gemm_args* args, int count, mtp_stream_t stream) {
    if (!args || count <= 0) return MTP_ERR_ARG;
    if (count > TN_MAX_PROBLEMS) return MTP_ERR_UNSUPPORTED;
    TnGroup g = {};
    int64_t tiles = 0;
    for (int i = 0; i < count; ++i) {
        const mtp_gemm_args& a = args[i];
        if (!a.A || !a.B || !a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return MTP_ERR_ARG;
        if (((uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.C) & 15) return MTP_ERR_ARG;
        if (a.in_dtype != MTP_BF16 || a.out_dtype != MTP_F32) return MTP_ERR_UNSUPPORTED;
        if ((a.M % 8) || (a.N % 8) || (a.K % 128) || (a.lda % 8) || (a.ldb % 8) || (a.ldc % 4) || a.lda < a.M || a.ldb < a.N || a.ldc < a.N) return MTP_ERR_UNSUPPORTED;
        if ((uint64_t)a.K * (uint64_t)a.lda * 2 >= (1ull << 32) || (uint64_t)a.K * (uint64_t)a.ldb * 2 >= (1ull << 32)) return MTP_ERR_UNSUPPORTED;
        int splits = a.split_k > 1 ? a.split_k : 1;
        int64_t kchunk = a.K;
        if (splits > 1) {
            if (!a.aux || ((uintptr_t)a.aux & 15)) return MTP_ERR_ARG;
            if (!a.defer_sum && a.ldc != a.N) return MTP_ERR_UNSUPPORTED;      // (the reduction treats an image as M * N contiguous floats; checked BEFORE anything is launched)
            kchunk = ((a.K / 128 + splits - 1) / splits) * 128;
            splits = (int)((a.K + kchunk - 1) / kchunk);
        }
        TnProb& q = g.p[i];
        q.A = (const char*)a.A; q.B = (const char*)a.B; q.C = splits > 1 ? (float*)a.aux : (float*)a.C; q.colsum = a.colsum;
        // `workspace` (>= 4 bytes) = a device float that receives += sum(C^2): only for problems whose tiles hold the whole contraction
        if (a.workspace && splits > 1) return MTP_ERR_UNSUPPORTED;
        if (a.workspace && (a.workspace_bytes < 4 || ((uintptr_t)a.workspace & 3))) return MTP_ERR_ARG;
        q.sqn = (float*)a.workspace;
        q.M = (int)a.M; q.N = (int)a.N; q.K = (int)a.K;
        q.lda = (int)a.lda; q.ldb = (int)a.ldb; q.ldc = (int)a.ldc;
        q.tiles_m = (int)((a.M + P8_BM - 1) / P8_BM); q.tiles_n = (int)((a.N + P8_BN - 1) / P8_BN);
        q.splits = splits; q.kchunk = (int)kchunk;
        q.tile0 = (int)tiles;
        tiles += (int64_t)q.tiles_m * q.tiles_n * splits;
        if (tiles >= (1 << 30)) return MTP_ERR_UNSUPPORTED;
    }
    g.nprob = count;
    g.ntiles = (int)tiles;
    g.plain = (args[0].variant >> 1) & 1;
    // read-ahead phases (gemm_p8.h: the next phase's transpose reads issued under the current phase's MFMAs; round 5: +5-11 %); variant bit 19 = the
    // plain phases of rounds 2-4 (A/B, bit-identical)
    const int plain_phase = (args[0].variant >> 19) & 1;
    void (*kern)(TnGroup) = plain_phase ? gemm_tn_p8_kernel<0, 0> : gemm_tn_p8_kernel<0, 1>;
    static unsigned long long optin[2] = {0, 0};     // 128 KiB of dynamic LDS: opt-in once per kernel INSTANTIATION and device
    if (const int e = mtp_optin_lds((const void*)kern, P8_LDS, optin[plain_phase])) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(P8_THREADS), P8_LDS, (hipStream_t)stream, g);
    int rc = mtp_launch_status();
    // the split problems' images -> C
    const float* parts[12];
    float* outs[12];
    int64_t numel[12];
    int nsplit[12], n = 0;
    for (int i = 0; i < count && rc == 0; ++i) {
        if (g.p[i].splits > 1 && !args[i].defer_sum) {
            parts[n] = (const float*)args[i].aux; outs[n] = (float*)args[i].C; numel[n] = args[i].M * args[i].N; nsplit[n] = g.p[i].splits;
            ++n;
        }
        if (n == 12 || (n > 0 && i == count - 1)) {
            rc = mtp_sum_partials_batch(parts, outs, numel, nsplit, n, stream);
            n = 0;
        }
    }
    return rc;
}
