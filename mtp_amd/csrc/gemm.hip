// GEMM kernels for gfx950: bf16 MFMA (v_mfma_f32_16x16x32_bf16) and exact-f32 MFMA (v_mfma_f32_16x16x4_f32),
// one source for both precisions: everything is expressed in 16-byte "chunks" along the contraction dim
// (8 bf16 or 4 f32).  A K-step = 4 chunks: lane group g = lane>>4 owns chunk g of both operands, which is
// exactly the MFMA operand layout for bf16 (8 consecutive k per lane) and, for f32, a consistent k-permutation
// across 4 chained 16x16x4 MFMAs (k = 16*step + 4*g + e for MFMA e).
//
// Tile: 128 x 128 x 8 chunks (BK = 64 bf16 / 32 f32), 256 threads = 4 waves as 2(M) x 2(N), each wave 64x64 =
// 4x4 MFMA tiles.  LDS image per operand: [128 rows][8 chunk slots] (128 B rows), slot = chunk ^ (row & 7):
// conflict-free for the ds_read_b128 fragment reads (16-lane groups) and for the staging writes.
// MFMA operands are swapped (A-operand = weight rows n, B-operand = activation rows m) so each lane ends up with
// 4 consecutive output columns of one row -> 8/16-byte epilogue accesses.
//
// NT  C[m][n] = sum_k A[m][k] B[n][k]   : both operands K-contiguous -> direct-to-LDS loads (global_load_lds x16B),
//                                         swizzle applied on the per-lane SOURCE address (LDS image is lane-linear).
//                                         Default kernel: ONE 32 KiB stage, 4 workgroups per CU (gemm_nt_sb_kernel).
// TN  C[m][n] = sum_k A[k][m] B[k][n]   : weight gradients, split-K over gridDim.y.  bf16 + complete tiles: both tiles go to
//                                         LDS untransposed (LDS-DMA) and the fragments come out of ds_read_b64_tr_b16
//                                         (gemm_tn_tr_kernel); f32 / ragged shapes: tiles are transposed in registers (ExE
//                                         blocks) between the coalesced global loads and the ds_write_b128.
#include "gemm_common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int NT_THREADS = 256;
constexpr int ROW_BYTES = 128;              // 8 chunks x 16 B
constexpr int OPER_BYTES = BM * ROW_BYTES;  // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * OPER_BYTES; // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;  // double buffered: 64 KiB -> 2 blocks / CU

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

// out[i] = sum_s part[s][i]   (float4 granules; deterministic split-K reduction)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n4, int split) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 s = *reinterpret_cast<const float4*>(part + 4 * i);
        for (int z = 1; z < split; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(part + 4 * (i + (int64_t)z * n4));
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(out + 4 * i) = s;
    }
}

// ---- one 128x128 tile's worth of MFMAs out of one LDS stage ------------------------------------------------------
template <typename T>
__device__ __forceinline__ void compute_stage(const char* sA, const char* sB, f32x4_t (&acc)[4][4], int wm, int wn, int lane) {
    const int fr = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = *reinterpret_cast<const uint4*>(sA + lds_off(wm * 64 + i * 16 + fr, ks * 4 + g));
            b[i] = *reinterpret_cast<const uint4*>(sB + lds_off(wn * 64 + i * 16 + fr, ks * 4 + g));
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) Mma<T>::run(acc[ni][mi], b[ni], a[mi]);
    }
}

// ---- NT staging, direct to LDS -----------------------------------------------------------------------------------
// wave w fills rows [32w, 32w+32) of both operand tiles with 4 + 4 global_load_lds_dwordx4 (1 KiB each):
// lane l -> LDS slot (row = base + (l>>3), slot = l&7) receives global chunk (slot ^ (row&7)) of that row.
template <typename T>
__device__ __forceinline__ void stage_nt_glds(const KArgs& p, char* sA, char* sB, int m0, int n0, int kt, int wave, int lane) {
    constexpr int E = Elem<T>::kPerChunk;
    const int k0 = kt * 8 * E;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rbase = wave * 32 + i * 8;
        const int row = rbase + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;
        int rb = n0 + row; rb = rb < p.N ? rb : p.N - 1;
        const char* ga = p.A + ((int64_t)ra * p.lda + k0 + chunk * E) * (int64_t)sizeof(T);
        const char* gb = p.B + ((int64_t)rb * p.ldb + k0 + chunk * E) * (int64_t)sizeof(T);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                         (__attribute__((address_space(3))) void*)(sA + rbase * ROW_BYTES), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gb,
                                         (__attribute__((address_space(3))) void*)(sB + rbase * ROW_BYTES), 16, 0, 0);
    }
}

// ---- NT staging through registers (variant 1; also handles ragged K) --------------------------------------------------
template <typename T>
struct NtRegs {
    uint4 a[4], b[4];
    uint32_t ok;    // bits 0-3: a[i] in range, bits 4-7: b[i] in range
};
template <typename T>
__device__ __forceinline__ void load_nt_regs(const KArgs& p, NtRegs<T>& r, int m0, int n0, int kt, int tid) {
    constexpr int E = Elem<T>::kPerChunk;
    const int c = tid & 7;
    const int k = kt * 8 * E + c * E;
    const bool kok = k < p.K;
    uint32_t okm = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const bool oka = kok && (m0 + row < p.M), okb = kok && (n0 + row < p.N);
        okm |= (oka ? (1u << i) : 0u) | (okb ? (16u << i) : 0u);
        r.a[i] = ldg16(p.A + (oka ? ((int64_t)(m0 + row) * p.lda + k) * (int64_t)sizeof(T) : 0));
        r.b[i] = ldg16(p.B + (okb ? ((int64_t)(n0 + row) * p.ldb + k) * (int64_t)sizeof(T) : 0));
    }
    r.ok = okm;
}
__device__ __forceinline__ uint4 mask4(const uint4& v, uint32_t okm, int e) {
    const uint32_t m = (okm >> e) & 1u ? 0xffffffffu : 0u;
    return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m);
}
template <typename T>
__device__ __forceinline__ void store_nt_regs(const NtRegs<T>& r, char* sA, char* sB, int tid) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        *reinterpret_cast<uint4*>(sA + lds_off(row, c)) = mask4(r.a[i], r.ok, i);
        *reinterpret_cast<uint4*>(sB + lds_off(row, c)) = mask4(r.b[i], r.ok, 4 + i);
    }
}

template <typename T, typename Tout, int EPI, bool GLDS>
__global__ __launch_bounds__(NT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_kernel(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = (p.order & 1) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_m_ = (int)gridDim.x / p.tiles_n;
    const int m0 = ((p.order & 2) ? (tile % tiles_m_) : (tile / p.tiles_n)) * BM, n0 = ((p.order & 2) ? (tile / tiles_m_) : (tile % p.tiles_n)) * BN;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = p.k_tiles;
    if (GLDS) {
        stage_nt_glds<T>(p, smem, smem + OPER_BYTES, m0, n0, 0, wave, lane);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            char* cur = smem + (kt & 1) * STAGE_BYTES;
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            if (kt + 1 < nk) stage_nt_glds<T>(p, nxt, nxt + OPER_BYTES, m0, n0, kt + 1, wave, lane);
            compute_stage<T>(cur, cur + OPER_BYTES, acc, wm, wn, lane);
        }
    } else {
        NtRegs<T> r;
        load_nt_regs<T>(p, r, m0, n0, 0, tid);
        store_nt_regs<T>(r, smem, smem + OPER_BYTES, tid);
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();
            char* cur = smem + (kt & 1) * STAGE_BYTES;
            char* nxt = smem + ((kt + 1) & 1) * STAGE_BYTES;
            load_nt_regs<T>(p, r, m0, n0, kt + 1 < nk ? kt + 1 : kt, tid);   // unconditional prefetch (see gemm_tn_kernel)
            __builtin_amdgcn_sched_barrier(0);
            compute_stage<T>(cur, cur + OPER_BYTES, acc, wm, wn, lane);
            __builtin_amdgcn_sched_barrier(0);
            store_nt_regs<T>(r, nxt, nxt + OPER_BYTES, tid);
        }
    }
    epilogue<Tout, EPI, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// Single-LDS-stage variant (32 KiB -> 4 workgroups / 16 waves per CU): thread-level parallelism across resident workgroups
// covers each workgroup's load latency instead of a deeper software pipeline.
template <typename T, typename Tout, int EPI>
__global__ __launch_bounds__(NT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nt_sb_kernel(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = (p.order & 1) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    int m0, n0;
    if (p.order & 2) {   // grouped order: panels of 8 tile rows, rows fastest inside a panel (an XCD's ~128 resident tiles cover
                         // 8 rows x 16 columns instead of 5 rows x all columns: the B tiles are shared by 8 instead of ~5 tiles)
        const int tiles_m = (int)gridDim.x / p.tiles_n, per = 8 * p.tiles_n;
        const int grp = tile / per, r = tile - grp * per;
        const int gm = (tiles_m - grp * 8) < 8 ? (tiles_m - grp * 8) : 8;
        m0 = (grp * 8 + r % gm) * BM;
        n0 = (r / gm) * BN;
    } else {
        m0 = (tile / p.tiles_n) * BM;
        n0 = (tile % p.tiles_n) * BN;
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < p.k_tiles; ++kt) {
        stage_nt_glds<T>(p, smem, smem + OPER_BYTES, m0, n0, kt, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        compute_stage<T>(smem, smem + OPER_BYTES, acc, wm, wn, lane);
        __syncthreads();
    }
    epilogue<Tout, EPI, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// 256 x 128 tile, 8 waves (4 x 2, 64 x 64 each), one 48 KiB stage, 2 workgroups per CU: same occupancy and per-wave work as
// the kernel above with 25 % fewer L2->LDS bytes per flop (A tile shared by 2 x more columns).  Selected for wide N only
// (the tile count of N = 1024 problems would leave CUs idle).
constexpr int NT8_THREADS = 512;
constexpr int NT8_BM = 256;
constexpr int NT8_STAGE_BYTES = NT8_BM * ROW_BYTES + OPER_BYTES;
template <typename T, typename Tout, int EPI>
__global__ __launch_bounds__(NT8_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_nt_sb8_kernel(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = Elem<T>::kPerChunk;
    char* sA = smem;
    char* sB = smem + NT8_BM * ROW_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = (p.order & 1) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    int m0, n0;
    if (p.order & 2) {
        const int tiles_m = (int)gridDim.x / p.tiles_n, per = 8 * p.tiles_n;
        const int grp = tile / per, r = tile - grp * per;
        const int gm = (tiles_m - grp * 8) < 8 ? (tiles_m - grp * 8) : 8;
        m0 = (grp * 8 + r % gm) * NT8_BM;
        n0 = (r / gm) * BN;
    } else {
        m0 = (tile / p.tiles_n) * NT8_BM;
        n0 = (tile % p.tiles_n) * BN;
    }
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < p.k_tiles; ++kt) {
        const int k0 = kt * 8 * E;
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // A: wave w fills rows [32w, 32w + 32)
            const int rbase = wave * 32 + i * 8, row = rbase + (lane >> 3), chunk = (lane & 7) ^ (row & 7);
            int ra = m0 + row; ra = ra < p.M ? ra : p.M - 1;
            const char* ga = p.A + ((int64_t)ra * p.lda + k0 + chunk * E) * (int64_t)sizeof(T);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                             (__attribute__((address_space(3))) void*)(sA + rbase * ROW_BYTES), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // B: wave w fills rows [16w, 16w + 16)
            const int rbase = wave * 16 + i * 8, row = rbase + (lane >> 3), chunk = (lane & 7) ^ (row & 7);
            int rb = n0 + row; rb = rb < p.N ? rb : p.N - 1;
            const char* gb = p.B + ((int64_t)rb * p.ldb + k0 + chunk * E) * (int64_t)sizeof(T);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gb,
                                             (__attribute__((address_space(3))) void*)(sB + rbase * ROW_BYTES), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        compute_stage<T>(sA, sB, acc, wm, wn, lane);
        __syncthreads();
    }
    epilogue<Tout, EPI, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// ---- TN staging: ExE register transposes ------------------------------------------------------------------------------
// Operand source is (Kc, X) row-major, X = M or N contiguous.  Item = one ExE block (E k-rows x E x-columns):
// kb = blk & 7 (LDS chunk column), xb = blk >> 3 (E-wide column group); lanes with consecutive xb read consecutive
// 16-B pieces of a source row, lanes with consecutive kb write the 8 chunk slots of one LDS row (conflict-free).
template <typename T>
struct TnItems;
template <>
struct TnItems<bf16_t> {
    static constexpr int kItems = 1;   // 2 operands x 128 blocks / 256 threads
};
template <>
struct TnItems<float> {
    static constexpr int kItems = 2;   // 2 operands x 256 blocks / 256 threads
};

template <typename T>
struct TnRegs {
    uint4 v[TnItems<T>::kItems][Elem<T>::kPerChunk];
    uint32_t ok[TnItems<T>::kItems];   // bit e: row e of the block is in range (else it is zeroed in the store phase)
};

template <typename T, bool FULL>
__device__ __forceinline__ void load_tn_regs(const KArgs& p, TnRegs<T>& r, int m0, int n0, int kt, int tid) {
    constexpr int E = Elem<T>::kPerChunk;
    constexpr int NB = 8 * (BM / E);   // blocks per operand
#pragma unroll
    for (int it = 0; it < TnItems<T>::kItems; ++it) {
        const int idx = tid + NT_THREADS * it;
        const int oper = idx / NB, blk = idx % NB;
        const int kb = blk & 7, xb = blk >> 3;
        const char* src = oper ? p.B : p.A;
        const int64_t ld = oper ? p.ldb : p.lda;
        const int x = (oper ? n0 : m0) + xb * E;
        const int xlim = oper ? p.N : p.M;
        // Unconditional loads from CLAMPED OFFSETS; out-of-range rows are zeroed later, in the store phase.  (`cond ? load : 0`
        // makes hipcc branch around every load with a vmcnt(0) inside; a select between two POINTERS is lowered to exec-masked
        // blocks as well; a select on the loaded VALUE here would pull the vmcnt wait in front of the MFMAs.)
        // FULL (every tile complete: the training shapes) has no predicates at all -- hipcc keeps re-introducing exec-masked
        // blocks + early vmcnt waits around predicated loads, whichever way the predicate is written.
        uint32_t okm = 0;
        if (FULL) {
            const char* base = src + ((int64_t)(kt * 8 * E + kb * E) * ld + x) * (int64_t)sizeof(T);
#pragma unroll
            for (int e = 0; e < E; ++e) r.v[it][e] = ldg16(base + (int64_t)e * ld * (int64_t)sizeof(T));
            okm = 0xffu;
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int k = kt * 8 * E + kb * E + e;
                const bool ok = (k < p.K) && (x < xlim);
                okm |= ok ? (1u << e) : 0u;
                const int64_t off = ok ? ((int64_t)k * ld + x) * (int64_t)sizeof(T) : 0;
                r.v[it][e] = ldg16(src + off);
            }
        }
        r.ok[it] = okm;
    }
}


__device__ __forceinline__ uint32_t dw(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

template <typename T>
__device__ __forceinline__ void store_tn_regs(const TnRegs<T>& r, char* smem_stage, int tid);

template <>
__device__ __forceinline__ void store_tn_regs<float>(const TnRegs<float>& r, char* st, int tid) {
    constexpr int E = 4, NB = 8 * (BM / E);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = tid + NT_THREADS * it;
        const int oper = idx / NB, blk = idx % NB;
        const int kb = blk & 7, xb = blk >> 3;
        char* s = st + oper * OPER_BYTES;
#pragma unroll
        for (int f = 0; f < 4; ++f) {   // out row f = column f of the 4x4 block
            const uint32_t okm = r.ok[it];
            uint4 o = make_uint4((okm & 1u) ? dw(r.v[it][0], f) : 0u, (okm & 2u) ? dw(r.v[it][1], f) : 0u,
                                 (okm & 4u) ? dw(r.v[it][2], f) : 0u, (okm & 8u) ? dw(r.v[it][3], f) : 0u);
            *reinterpret_cast<uint4*>(s + lds_off(xb * E + f, kb)) = o;
        }
    }
}

template <>
__device__ __forceinline__ void store_tn_regs<bf16_t>(const TnRegs<bf16_t>& r, char* st, int tid) {
    constexpr int E = 8, NB = 8 * (BM / E);
    const int oper = tid / NB, blk = tid % NB;
    const int kb = blk & 7, xb = blk >> 3;
    char* s = st + oper * OPER_BYTES;
    uint4 v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = mask4(r.v[0][e], r.ok[0], e);
#pragma unroll
    for (int d = 0; d < 4; ++d) {       // source dword d holds columns f = 2d (lo half), 2d+1 (hi half)
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // output dword q holds k = 2q (lo), 2q+1 (hi)
            const uint32_t e0 = dw(v[2 * q], d), e1 = dw(v[2 * q + 1], d);
            lo[q] = (e0 & 0xffffu) | (e1 << 16);
            hi[q] = (e0 >> 16) | (e1 & 0xffff0000u);
        }
        *reinterpret_cast<uint4*>(s + lds_off(xb * E + 2 * d, kb)) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint4*>(s + lds_off(xb * E + 2 * d + 1, kb)) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    }
}

template <typename T, bool FULL>
__global__ __launch_bounds__(NT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn_kernel(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = (p.order & 1) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_m_ = (int)gridDim.x / p.tiles_n;
    const int m0 = ((p.order & 2) ? (tile % tiles_m_) : (tile / p.tiles_n)) * BM, n0 = ((p.order & 2) ? (tile / tiles_m_) : (tile % p.tiles_n)) * BN;
    const int kt0 = blockIdx.y * p.k_tiles_per_split;
    int kt1 = kt0 + p.k_tiles_per_split;
    kt1 = kt1 < p.k_tiles ? kt1 : p.k_tiles;
    if (kt0 >= kt1) return;
    p.C += (int64_t)blockIdx.y * p.split_stride * (int64_t)sizeof(float);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    TnRegs<T> r;
    load_tn_regs<T, FULL>(p, r, m0, n0, kt0, tid);
    store_tn_regs<T>(r, smem, tid);
    for (int kt = kt0; kt < kt1; ++kt) {
        __syncthreads();
        char* cur = smem + ((kt - kt0) & 1) * STAGE_BYTES;
        char* nxt = smem + ((kt - kt0 + 1) & 1) * STAGE_BYTES;
        // The prefetch is UNCONDITIONAL (the last iteration re-loads its own tile into the idle buffer): an `if` around it
        // turns the registers into loop phis and hipcc copies them -- i.e. waits for the loads -- before the MFMAs.
        const int ktn = kt + 1 < kt1 ? kt + 1 : kt;
        load_tn_regs<T, FULL>(p, r, m0, n0, ktn, tid);
        __builtin_amdgcn_sched_barrier(0);   // keep the 8 global loads ahead of the MFMAs (hipcc otherwise sinks them to their use)
        compute_stage<T>(cur, cur + OPER_BYTES, acc, wm, wn, lane);
        __builtin_amdgcn_sched_barrier(0);
        store_tn_regs<T>(r, nxt, tid);
    }
    epilogue<float, MTP_EPI_BIAS, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// Single-LDS-stage TN variant (32 KiB, 3 workgroups per CU): the prefetched tile waits in registers during the MFMAs and is
// written to the one stage between two barriers; the other resident workgroups cover the bubbles.
template <typename T, bool FULL>
__global__ __launch_bounds__(NT_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void gemm_tn_sb_kernel(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = (p.order & 1) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int kt0 = blockIdx.y * p.k_tiles_per_split;
    int kt1 = kt0 + p.k_tiles_per_split;
    kt1 = kt1 < p.k_tiles ? kt1 : p.k_tiles;
    if (kt0 >= kt1) return;
    p.C += (int64_t)blockIdx.y * p.split_stride * (int64_t)sizeof(float);
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    TnRegs<T> r;
    load_tn_regs<T, FULL>(p, r, m0, n0, kt0, tid);
    store_tn_regs<T>(r, smem, tid);
    for (int kt = kt0; kt < kt1; ++kt) {
        __syncthreads();
        const int ktn = kt + 1 < kt1 ? kt + 1 : kt;
        load_tn_regs<T, FULL>(p, r, m0, n0, ktn, tid);
        __builtin_amdgcn_sched_barrier(0);
        compute_stage<T>(smem, smem + OPER_BYTES, acc, wm, wn, lane);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        store_tn_regs<T>(r, smem, tid);
    }
    epilogue<float, MTP_EPI_BIAS, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// ---- TN, bf16, complete tiles: LDS-DMA staging of the UNtransposed tiles + hardware transpose reads ----------------------
// The register-transposing kernels above spend more LDS-pipe cycles on their 8 ds_write_b128 per thread and k-tile (13 cycles
// each) plus the fragment reads than the SIMDs spend on MFMAs.  Here both operand tiles go HBM -> LDS as they lie in memory
// ([64 t rows][128 x] bf16 = 256-B rows, global_load_lds x16B, no VGPR/ds_write traffic) and the MFMA fragments come out of
// ds_read_b64_tr_b16: a 16-lane group hands in the addresses of a [4 t][16 x] block (lane i: row i>>2, columns 4(i&3)..+3)
// and lane i gets column i, rows 0..3 -- measured on gfx950 with tools/probes/tr_probe.hip.  Two reads = the 8 consecutive k of
// one MFMA operand.  Swizzle: 32-B slot pair ^= f(t), f = (t&3) | (t>>3 & 1)<<2, so that the 8 rows a 32-lane group touches
// ({0..3, 8..11} + const) fall on 8 different bank octets; applied on the per-lane SOURCE address of the DMA.
typedef short tr_v4s __attribute__((ext_vector_type(4)));
constexpr int TR_ROW_BYTES = 256;

__device__ __forceinline__ void stage_tn_glds(const KArgs& p, char* sA, char* sB, int m0, int n0, int kt, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rbase = wave * 16 + i * 4;
        const int row = rbase + (lane >> 4);
        const int f = (row & 3) | ((row >> 1) & 4);
        const int chunk = (lane & 15) ^ (f << 1);
        const int64_t t = (int64_t)kt * 64 + row;
        // edge tiles (M or N not a multiple of 128, e.g. InternImage's 192 channels): columns past the matrix edge re-read the last
        // complete 16-byte chunk of the row; their products land in output columns the epilogue never stores
        int ca = m0 + chunk * 8, cb = n0 + chunk * 8;
        ca = ca < p.M - 8 ? ca : p.M - 8;
        cb = cb < p.N - 8 ? cb : p.N - 8;
        const char* ga = p.A + (t * p.lda + ca) * 2;
        const char* gb = p.B + (t * p.ldb + cb) * 2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                         (__attribute__((address_space(3))) void*)(sA + rbase * TR_ROW_BYTES), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gb,
                                         (__attribute__((address_space(3))) void*)(sB + rbase * TR_ROW_BYTES), 16, 0, 0);
    }
}

__device__ __forceinline__ uint4 tr_frag(const char* s, int off) {
    const tr_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(s + off));
    const tr_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4s*)(s + off + 4 * TR_ROW_BYTES));
    const uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, h.x, h.y);
}

__device__ __forceinline__ void compute_stage_tr(const char* sA, const char* sB, f32x4_t (&acc)[4][4], int wm, int wn, int lane) {
    const int i = lane & 15, g = lane >> 4;
    const int f = (i >> 2) | ((g & 1) << 2);
    const int rowoff = (8 * g + (i >> 2)) * TR_ROW_BYTES + ((i >> 1) & 1) * 16 + (i & 1) * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint4 a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = tr_frag(sA, rowoff + (((q | (wm << 2)) ^ f) << 5) + ks * 32 * TR_ROW_BYTES);
            b[q] = tr_frag(sB, rowoff + (((q | (wn << 2)) ^ f) << 5) + ks * 32 * TR_ROW_BYTES);
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) Mma<bf16_t>::run(acc[ni][mi], b[ni], a[mi]);
    }
}

__global__ __launch_bounds__(NT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_tn_tr_kernel(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = (p.order & 1) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    // an XCD owns a contiguous tile range and streams, per split, its share of one operand but ALL panels of the other one:
    // walk the tiles so that the replicated operand is the smaller one (order bit 1: M fastest, for N > M, e.g. dW of fc2)
    const int tiles_m = (int)gridDim.x / p.tiles_n;
    const int m0 = ((p.order & 2) ? tile % tiles_m : tile / p.tiles_n) * BM, n0 = ((p.order & 2) ? tile / tiles_m : tile % p.tiles_n) * BN;
    const int kt0 = blockIdx.y * p.k_tiles_per_split;
    int kt1 = kt0 + p.k_tiles_per_split;
    kt1 = kt1 < p.k_tiles ? kt1 : p.k_tiles;
    if (kt0 >= kt1) return;
    p.C += (int64_t)blockIdx.y * p.split_stride * (int64_t)sizeof(float);
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // bias-gradient by-product: the workgroups of tile column 0 also add up the rows of their A (= dY) tiles out of LDS
    // (thread = 16-B slot tid&15 of rows (tid>>4) + 16j; all four rows share one swizzle, i.e. one 8-column chunk)
    const bool do_colsum = p.colsum != nullptr && n0 == 0;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int kt = kt0; kt < kt1; ++kt) {
        stage_tn_glds(p, smem, smem + OPER_BYTES, m0, n0, kt, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        compute_stage_tr(smem, smem + OPER_BYTES, acc, wm, wn, lane);
        if (do_colsum) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem + ((tid >> 4) + 16 * j) * TR_ROW_BYTES + (tid & 15) * 16);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cs[2 * e] += __uint_as_float(w[e] << 16);
                    cs[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                }
            }
        }
        __syncthreads();
    }
    if (do_colsum) {   // (after the loop's last barrier nobody reads the stage any more)
        float* red = reinterpret_cast<float*>(smem);
        const int rg = tid >> 4, f = (rg & 3) | ((rg >> 1) & 4), chunk = (tid & 15) ^ (f << 1);
        *reinterpret_cast<float4*>(red + rg * BM + chunk * 8) = make_float4(cs[0], cs[1], cs[2], cs[3]);
        *reinterpret_cast<float4*>(red + rg * BM + chunk * 8 + 4) = make_float4(cs[4], cs[5], cs[6], cs[7]);
        __syncthreads();
        if (tid < BM) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r * BM + tid];
            if (m0 + tid < p.M) atomicAdd(p.colsum + m0 + tid, s);
        }
    }
    epilogue<float, MTP_EPI_BIAS, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

template <typename T>
int fill_common(const mtp_gemm_args* a, KArgs& k) {
    constexpr int E = Elem<T>::kPerChunk;
    if (!a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return MTP_ERR_ARG;
    if (a->M > INT32_MAX || a->N > INT32_MAX || a->K > INT32_MAX) return MTP_ERR_ARG;
    if ((a->lda % E) || (a->ldb % E) || (a->ldc % 4) || (a->N % 4)) return MTP_ERR_ARG;
    if (((uintptr_t)a->A | (uintptr_t)a->B | (uintptr_t)a->C) & 15) return MTP_ERR_ARG;
    k.A = (const char*)a->A; k.B = (const char*)a->B; k.C = (char*)a->C;
    k.M = (int)a->M; k.N = (int)a->N; k.K = (int)a->K;
    k.lda = a->lda; k.ldb = a->ldb; k.ldc = a->ldc;
    k.bias = a->bias; k.bias_mod = (int)a->bias_mod;
    k.res = a->res; k.res_ld = a->res_ld; k.res_mod = (int)a->res_mod;
    k.rowscale = a->rowscale; k.rows_per_sample = (int)(a->rows_per_sample > 0 ? a->rows_per_sample : 1);
    k.aux = (char*)a->aux; k.aux_ld = a->aux_ld;
    k.tiles_n = (int)((a->N + BN - 1) / BN);
    k.k_tiles = (int)((a->K + 8 * E - 1) / (8 * E));
    k.k_tiles_per_split = k.k_tiles;
    k.atomic_out = 0;
    k.order = (a->variant >> 1) & 3;
    k.split_stride = 0;
    k.colsum = nullptr;
    return 0;
}

// which NT kernel family runs a bf16 problem: 0 = the 128-wide kernels of this file, else gemm_p8.hip: 1 = tile height picked per
// problem (256 or 224 rows), 2 = 224-row tiles, 3 = 256-row tiles.  variant bits 8-9 force 1 / 2 / 3 (when the problem fits), bit 10
// forbids the kernel; bits 11-14 select an ablation build (tools/ab_gemm.py).
int nt_p8_mode(const mtp_gemm_args* a, const KArgs& k, int cus) {
    if (a->in_dtype != MTP_BF16 || (a->variant & 1024) || !mtp_nt_p8_fits(k, a->out_dtype, a->epilogue)) return 0;
    const int forced = (a->variant >> 8) & 3;
    if (forced) return forced;
    // default: the pipelined kernel once its 256-wide tiles occupy a good part of the 256 CUs (one workgroup per CU); below that the
    // 128-wide kernels with 4 workgroups per CU spread a small problem better.  Measured on the ViT-L shapes (tools/ab_gemm.py):
    // +15 % (N = 3072 / 4096, K = 1024) ... +25 % (N = 1024, K = 3072 / 4096), +23 % on the FPN GEMM; on the mid-size shapes of
    // InternImage-XL's 768- / 1536-channel levels and of ViT-B at batch 32 (tools/probes/ab_gemm_mid.py, round 3): 96 tiles +6 % (K = 768) /
    // +17 % (K = 3072), 75 tiles +3 % / +16 %, 48 tiles +5 % (K = 1536) / +14 % (K = 6144), but 64 tiles of which half are mostly edge
    // (N = 432, K = 768) -19 %: from 72 tiles on, or from 40 when the contraction is long.
    // (cus: the CUs of the stream, 256 unless it is CU-masked -- the thresholds are fractions of a round)
    const int64_t tiles = ((a->M + 255) / 256) * ((a->N + 255) / 256) * 256 / cus;
    return (tiles >= 72 || (tiles >= 40 && a->K >= 1536)) ? 1 : 0;
}

// the strip kernel of gemm_s8.hip (two accumulator sets, the epilogue of a strip under the next strip's K loop): variant bit 17 forces
// it (when the problem fits), bit 18 forbids it.
int nt_s8_mode(const mtp_gemm_args* a, const KArgs& k, int cus) {
    if (a->in_dtype != MTP_BF16 || (a->variant & (1024 | (1 << 18))) || ((a->variant >> 8) & 3) || !mtp_nt_s8_fits(k, a->out_dtype, a->epilogue)) return 0;
    if (a->variant & (1 << 17)) return 1;
    // default: problems of less than half a round of 256 x 256 tiles on the 256 CUs -- the 768- / 1536-channel levels of InternImage-XL, ViT-B at
    // batch 32 with N = C.  Measured (tools/ab_gemm.py, MTP_AB_SHAPES=mid; profiles/r05_ab_gemm_s8_strip_mid_shapes.txt): 6272 x 768 x 768
    // 17.1 -> 15.6 us, 6272 x 768 x 3072 45.4 -> 38.3, 8192 x 768 x 768 16.8 -> 15.4, 8192 x 768 x 3072 45.5 -> 38.8, 2048 x 1536 x 1536 26.0 -> 22.6,
    // 2048 x 1536 x 6144 83.3 -> 67.9 (twice as many work units, each half as long, the epilogue of all but the last hidden); from 225 tiles on the
    // 8-wave kernel wins (its loop moves 2/3 of the L2 -> LDS bytes per flop): 6272 x 2304 x 768 22.5 vs 29.0, every ViT-L shape 10-38 %.
    // Below 40 tiles (not measured with this kernel) the 128-wide kernels keep the problem: 4 x as many, smaller workgroups.
    const int64_t tiles = ((a->M + 255) / 256) * ((a->N + 255) / 256) * 256 / cus;
    return tiles >= 40 && tiles <= 128;
}

template <typename T, typename Tout, int EPI>
int launch_nt(const mtp_gemm_args* a, hipStream_t stream) {
    constexpr int E = Elem<T>::kPerChunk;
    KArgs k;
    int rc = fill_common<T>(a, k);
    if (rc) return rc;
    if (a->K % E) return MTP_ERR_ARG;
    if (EPI == MTP_EPI_BIAS_RES && (!a->res || (a->res_ld % 4))) return MTP_ERR_ARG;
    if ((EPI == MTP_EPI_BIAS_GELU || EPI == MTP_EPI_DGELU || EPI == MTP_EPI_BIAS_GELU_DG || EPI == MTP_EPI_MUL) && (!a->aux || (a->aux_ld % 4))) return MTP_ERR_ARG;
    if (a->bias && a->bias_mod > 0 && (a->bias_mod % 4)) return MTP_ERR_ARG;
    // 8-wave pipelined kernel (gemm_p8.hip; bf16, whole K-tile pairs): variant bits 8-9 pick the tile height (1 auto, 2 = 224 rows,
    // 3 = 256 rows), bits 11-14 an ablation build, bits 15 / 16 force / forbid persistent tiles; falls through to the 128-wide kernels
    // when the problem does not fit it
    if constexpr (sizeof(T) == 2) {
        const int cus = mtp_stream_cus(stream);
        if (nt_s8_mode(a, k, cus)) return mtp_nt_s8_launch(k, a->out_dtype, EPI, ((((a->variant >> 1) & 3) == 1) ? 2 : 0), stream);
        const int p8 = nt_p8_mode(a, k, cus);
        if (p8) return mtp_nt_p8_launch(k, a->out_dtype, EPI, (p8 == 2 ? 1 : p8 == 3 ? 4 : 0) | ((((a->variant >> 1) & 3) == 1) ? 2 : 0) | (((a->variant >> 15) & 3) << 8) | (((a->variant >> 20) & 3) << 13), stream);
    }
    const int tiles_m = (k.M + BM - 1) / BM;
    dim3 grid(tiles_m * k.tiles_n), block(NT_THREADS);
    const bool glds = ((a->variant & 1) == 0) && (a->K % (8 * E) == 0);
    // single-stage / 4 workgroups per CU (round 1: +20 % over a double-buffered 2-per-CU LDS-DMA kernel on every ViT-L shape; that kernel
    // is gone since round 4 -- the register-staged form below stays for ragged K and as variant bit 0)
    // tile order of the single-stage kernel: variant bits 1-2 = 0 auto, 1 plain blockIdx, 2 grouped, 3 row-major with XCD remap.
    // Grouped (panels of 8 tile rows) measured +2..7 % at N = 3072 and +8..11 % at N = 4096, -2..3 % at N = 1024.
    const int ord = (a->variant >> 1) & 3;
    k.order = ord == 0 ? (k.tiles_n > 8 ? 2 : 0) : ord == 1 ? 1 : ord == 2 ? 2 : 0;
    if (!glds) k.order = ord;   // the register-staged kernel keeps its own meaning of the bits
    // 256 x 128 tile / 8 waves when the whole problem is ONE round of such workgroups on the 256 CUs (2 per CU) and every CU
    // gets at least one: measured on M = 12544, N = 1024 (392 workgroups): +7.5 % (K = 1024), +17..18 % (K = 3072, 4096; up to
    // 1095 TF/s); with several rounds (N = 3072: equal, N = 4096: -5 %) the coarser tiles lose to the tail.  Variant bit 5
    // forces it, bit 6 forbids it.
    const int tiles_m8 = (k.M + NT8_BM - 1) / NT8_BM;
    const bool one_round = tiles_m8 * k.tiles_n >= 256 && tiles_m8 * k.tiles_n <= 512;
    if (glds && ((a->variant & 32) || (one_round && !(a->variant & 64)))) {
        if (ord == 0) k.order = k.tiles_n > 8 ? 2 : 0;
        hipLaunchKernelGGL((gemm_nt_sb8_kernel<T, Tout, EPI>), dim3(tiles_m8 * k.tiles_n), dim3(NT8_THREADS), NT8_STAGE_BYTES, stream, k);
        return mtp_launch_status();
    }
    if (glds)
        hipLaunchKernelGGL((gemm_nt_sb_kernel<T, Tout, EPI>), grid, block, STAGE_BYTES, stream, k);
    else
        hipLaunchKernelGGL((gemm_nt_kernel<T, Tout, EPI, false>), grid, block, LDS_BYTES, stream, k);
    return mtp_launch_status();
}

template <typename T>
int launch_tn(const mtp_gemm_args* a, hipStream_t stream) {
    constexpr int E = Elem<T>::kPerChunk;
    KArgs k;
    int rc = fill_common<T>(a, k);
    if (rc) return rc;
    if (a->out_dtype != MTP_F32 || (a->M % E) || (a->N % E)) return MTP_ERR_ARG;
    k.bias = nullptr;
    int split = a->split_k > 1 ? a->split_k : 1;
    if (split > k.k_tiles) split = k.k_tiles;
    k.k_tiles_per_split = (k.k_tiles + split - 1) / split;
    split = (k.k_tiles + k.k_tiles_per_split - 1) / k.k_tiles_per_split;
    // split-K: with a caller workspace (args.aux, split*M*N f32) every split stores its partial tile and a second kernel
    // sums them (deterministic; f32 atomics run at only ~80 G lane-ops/s and were slower than the GEMM itself);
    // without a workspace fall back to atomics into the zeroed output.
    const bool use_ws = split > 1 && a->aux != nullptr;
    k.atomic_out = split > 1 && !use_ws;
    k.split_stride = 0;
    if (split > 1) {
        if (a->ldc != a->N) return MTP_ERR_ARG;
        if (use_ws) {
            if ((uintptr_t)a->aux & 15) return MTP_ERR_ARG;
            k.C = (char*)a->aux;
            k.split_stride = (int64_t)a->M * a->N;
        } else {
            hipError_t e = hipMemsetAsync(a->C, 0, sizeof(float) * (size_t)a->M * (size_t)a->N, stream);
            if (e != hipSuccess) return (int)e;
        }
    }
    const int tiles_m = (k.M + BM - 1) / BM;
    dim3 grid(tiles_m * k.tiles_n, split), block(NT_THREADS);
    const bool full = (a->K % (8 * E) == 0) && (a->M % BM == 0) && (a->N % BN == 0);
    // bf16 complete tiles: LDS-DMA + transpose-read kernel; variant bit 4 falls back to the register-transposing kernels
    // (the transpose-read kernel also takes edge tiles as long as the rows split into whole 16-byte chunks and K into whole 64-row stages)
    const bool tr = sizeof(T) == 2 && !(a->variant & 16) && (full || ((a->K % 64 == 0) && (a->M % 8 == 0) && (a->N % 8 == 0) && a->M >= 8 && a->N >= 8 && !(a->variant & 32768)));
    if (a->colsum && !tr) {   // the other kernels do not produce the column sums: separate streaming pass over A
        rc = mtp_colsum_acc(a->A, a->in_dtype, a->lda, a->colsum, a->K, a->M, stream);
        if (rc) return rc;
    }
    if (tr) {
        k.colsum = a->colsum;
        const int ord = (a->variant >> 1) & 3;   // 0 auto, 1 plain blockIdx, 2 M-fastest, 3 N-fastest
        k.order = ord == 0 ? (a->N > a->M ? 2 : 0) : ord == 1 ? 1 : ord == 2 ? 2 : 0;
        hipLaunchKernelGGL(gemm_tn_tr_kernel, grid, block, STAGE_BYTES, stream, k);
    } else if (full && (a->variant & 8))
        hipLaunchKernelGGL((gemm_tn_sb_kernel<T, true>), grid, block, STAGE_BYTES, stream, k);
    else if (full)
        hipLaunchKernelGGL((gemm_tn_kernel<T, true>), grid, block, LDS_BYTES, stream, k);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<T, false>), grid, block, LDS_BYTES, stream, k);
    if (use_ws && !a->defer_sum) {   // defer_sum: the caller sums the partial tiles later (mtp_sum_partials_batch, several GEMMs per launch)
        const int64_t n4 = (int64_t)a->M * a->N / 4;
        int64_t nb = (n4 + 255) / 256;
        hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, stream, (const float*)a->aux, (float*)a->C, n4, split);
    }
    return mtp_launch_status();
}

// several split-K reductions in one launch (blockIdx.y = job); the table travels in the kernel arguments
struct SumJobs {
    const float* part[MTP_MAX_SEGMENTS];
    float* out[MTP_MAX_SEGMENTS];
    int64_t n4[MTP_MAX_SEGMENTS];
    int split[MTP_MAX_SEGMENTS];
};
__global__ __launch_bounds__(256) void sum_partials_batch_kernel(SumJobs t) {
    const int j = blockIdx.y;
    const float* __restrict__ part = t.part[j];
    float* __restrict__ out = t.out[j];
    const int64_t n4 = t.n4[j];
    const int split = t.split[j];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 s = *reinterpret_cast<const float4*>(part + 4 * i);
        for (int z = 1; z < split; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(part + 4 * (i + (int64_t)z * n4));
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(out + 4 * i) = s;
    }
}

template <typename T, typename Tout>
int dispatch_epi(const mtp_gemm_args* a, hipStream_t s) {
    switch (a->epilogue) {
        case MTP_EPI_BIAS: return launch_nt<T, Tout, MTP_EPI_BIAS>(a, s);
        case MTP_EPI_BIAS_GELU: return launch_nt<T, Tout, MTP_EPI_BIAS_GELU>(a, s);
        case MTP_EPI_DGELU: return launch_nt<T, Tout, MTP_EPI_DGELU>(a, s);
        case MTP_EPI_BIAS_GELU_DG: return launch_nt<T, Tout, MTP_EPI_BIAS_GELU_DG>(a, s);
        case MTP_EPI_MUL: return launch_nt<T, Tout, MTP_EPI_MUL>(a, s);
        default: return MTP_ERR_UNSUPPORTED;
    }
}

}  // namespace

extern "C" int mtp_gemm_nt(const mtp_gemm_args* a, mtp_stream_t stream) {
    if (!a) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (a->epilogue == MTP_EPI_BIAS_RES) {
        if (a->out_dtype != MTP_F32) return MTP_ERR_ARG;
        return a->in_dtype == MTP_BF16 ? launch_nt<bf16_t, float, MTP_EPI_BIAS_RES>(a, s)
                                      : launch_nt<float, float, MTP_EPI_BIAS_RES>(a, s);
    }
    if (a->in_dtype == MTP_BF16 && a->out_dtype == MTP_BF16) return dispatch_epi<bf16_t, bf16_t>(a, s);
    if (a->in_dtype == MTP_F32 && a->out_dtype == MTP_F32) return dispatch_epi<float, float>(a, s);
    if (a->in_dtype == MTP_BF16 && a->out_dtype == MTP_F32 && a->epilogue == MTP_EPI_BIAS) return launch_nt<bf16_t, float, MTP_EPI_BIAS>(a, s);
    return MTP_ERR_UNSUPPORTED;
}

// (round 4: the stream-K form that used `workspace` was removed -- measured slower; the query stays in the ABI and answers 0)
extern "C" int64_t mtp_gemm_nt_workspace_bytes(void) { return 0; }

extern "C" int mtp_gemm_nt_tile(const mtp_gemm_args* a) {
    if (!a || !a->A || !a->B || !a->C || a->M <= 0 || a->N <= 0 || a->K <= 0) return MTP_ERR_ARG;
    if (a->in_dtype != MTP_BF16) return 128;
    KArgs k;
    if (fill_common<bf16_t>(a, k)) return MTP_ERR_ARG;
    const int cus = mtp_stream_cus(nullptr);
    if (nt_s8_mode(a, k, cus)) return 64;
    return nt_p8_mode(a, k, cus) ? 256 : 128;
}

extern "C" int mtp_sum_partials_batch(const float* const* parts, float* const* outs, const int64_t* numel, const int* splits, int count, mtp_stream_t stream) {
    if (!parts || !outs || !numel || !splits || count <= 0 || count > MTP_MAX_SEGMENTS) return MTP_ERR_ARG;
    SumJobs t;
    int64_t mx = 0;
    for (int i = 0; i < count; ++i) {
        if (!parts[i] || !outs[i] || numel[i] <= 0 || (numel[i] % 4) || splits[i] < 1) return MTP_ERR_ARG;
        t.part[i] = parts[i]; t.out[i] = outs[i]; t.n4[i] = numel[i] / 4; t.split[i] = splits[i];
        mx = t.n4[i] > mx ? t.n4[i] : mx;
    }
    int64_t nb = (mx + 255) / 256;
    hipLaunchKernelGGL(sum_partials_batch_kernel, dim3((unsigned)(nb > 2048 ? 2048 : nb), (unsigned)count), dim3(256), 0, (hipStream_t)stream, t);
    return mtp_launch_status();
}

extern "C" int mtp_gemm_tn(const mtp_gemm_args* a, mtp_stream_t stream) {
    if (!a) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    return a->in_dtype == MTP_BF16 ? launch_tn<bf16_t>(a, s) : launch_tn<float>(a, s);
}
