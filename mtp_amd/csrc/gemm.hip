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
// TN  C[m][n] = sum_k A[k][m] B[k][n]   : weight gradients; tiles are transposed in registers (ExE blocks) between the
//                                         coalesced global loads and the ds_write_b128, split-K over gridDim.y.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int NT_THREADS = 256;
constexpr int ROW_BYTES = 128;              // 8 chunks x 16 B
constexpr int OPER_BYTES = BM * ROW_BYTES;  // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * OPER_BYTES; // A + B
constexpr int LDS_BYTES = 2 * STAGE_BYTES;  // double buffered: 64 KiB -> 2 blocks / CU

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

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ (row & 7)) << 4); }

// bijective XCD-aware remap: hardware places block b on XCD b % 8; give each XCD a contiguous range of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    int xcd = bid & 7, idx = bid >> 3, q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

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
};

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

// ---- epilogue: lane holds C[m0 + wm*64 + mi*16 + (lane&15)][n0 + wn*64 + ni*16 + 4*(lane>>4) + 0..3] -------------
template <typename Tout, int EPI>
__device__ __forceinline__ void epilogue(const KArgs& p, f32x4_t (&acc)[4][4], int m0, int n0, int wm, int wn, int lane) {
    const int fr = lane & 15, g = lane >> 4;
    Tout* C = reinterpret_cast<Tout*>(p.C);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + fr;
        if (m >= p.M) continue;
        float rs = 1.0f;
        const float* resrow = nullptr;
        if (EPI == MTP_EPI_BIAS_RES) {
            if (p.rowscale) rs = p.rowscale[m / p.rows_per_sample];
            resrow = p.res + (int64_t)(p.res_mod > 0 ? m % p.res_mod : m) * p.res_ld;
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + g * 4;
            if (n >= p.N) continue;
            float4 v = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
            if (p.atomic_out) {   // split-K weight gradient: f32 atomics into a zeroed buffer
                float* c = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
                atomicAdd(c + 0, v.x); atomicAdd(c + 1, v.y); atomicAdd(c + 2, v.z); atomicAdd(c + 3, v.w);
                continue;
            }
            if (EPI != MTP_EPI_DGELU && p.bias) {
                const int bn = p.bias_mod > 0 ? n % p.bias_mod : n;
                float4 b = *reinterpret_cast<const float4*>(p.bias + bn);
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
            }
            if (EPI == MTP_EPI_BIAS_GELU) {
                store4(reinterpret_cast<Tout*>(p.aux) + (int64_t)m * p.aux_ld + n, v);
                v = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
            } else if (EPI == MTP_EPI_DGELU) {
                float4 u = load4(reinterpret_cast<const Tout*>(p.aux) + (int64_t)m * p.aux_ld + n);
                v = make_float4(v.x * dgelu_f(u.x), v.y * dgelu_f(u.y), v.z * dgelu_f(u.z), v.w * dgelu_f(u.w));
            } else if (EPI == MTP_EPI_BIAS_RES) {
                float4 r = *reinterpret_cast<const float4*>(resrow + n);
                v = make_float4(r.x + rs * v.x, r.y + rs * v.y, r.z + rs * v.z, r.w + rs * v.w);
            }
            store4(C + (int64_t)m * p.ldc + n, v);
        }
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
};
template <typename T>
__device__ __forceinline__ void load_nt_regs(const KArgs& p, NtRegs<T>& r, int m0, int n0, int kt, int tid) {
    constexpr int E = Elem<T>::kPerChunk;
    const int c = tid & 7;
    const int k = kt * 8 * E + c * E;
    const bool kok = k < p.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const uint4 z = make_uint4(0, 0, 0, 0);
        r.a[i] = (kok && m0 + row < p.M) ? *reinterpret_cast<const uint4*>(p.A + ((int64_t)(m0 + row) * p.lda + k) * (int64_t)sizeof(T)) : z;
        r.b[i] = (kok && n0 + row < p.N) ? *reinterpret_cast<const uint4*>(p.B + ((int64_t)(n0 + row) * p.ldb + k) * (int64_t)sizeof(T)) : z;
    }
}
template <typename T>
__device__ __forceinline__ void store_nt_regs(const NtRegs<T>& r, char* sA, char* sB, int tid) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        *reinterpret_cast<uint4*>(sA + lds_off(row, c)) = r.a[i];
        *reinterpret_cast<uint4*>(sB + lds_off(row, c)) = r.b[i];
    }
}

template <typename T, typename Tout, int EPI, bool GLDS>
__global__ __launch_bounds__(NT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_nt_kernel(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;

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
            if (kt + 1 < nk) load_nt_regs<T>(p, r, m0, n0, kt + 1, tid);
            compute_stage<T>(cur, cur + OPER_BYTES, acc, wm, wn, lane);
            if (kt + 1 < nk) store_nt_regs<T>(r, nxt, nxt + OPER_BYTES, tid);
        }
    }
    epilogue<Tout, EPI>(p, acc, m0, n0, wm, wn, lane);
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
};

template <typename T>
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
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = kt * 8 * E + kb * E + e;
            r.v[it][e] = (k < p.K && x < xlim) ? *reinterpret_cast<const uint4*>(src + ((int64_t)k * ld + x) * (int64_t)sizeof(T))
                                              : make_uint4(0, 0, 0, 0);
        }
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
            uint4 o = make_uint4(dw(r.v[it][0], f), dw(r.v[it][1], f), dw(r.v[it][2], f), dw(r.v[it][3], f));
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
#pragma unroll
    for (int d = 0; d < 4; ++d) {       // source dword d holds columns f = 2d (lo half), 2d+1 (hi half)
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // output dword q holds k = 2q (lo), 2q+1 (hi)
            const uint32_t e0 = dw(r.v[0][2 * q], d), e1 = dw(r.v[0][2 * q + 1], d);
            lo[q] = (e0 & 0xffffu) | (e1 << 16);
            hi[q] = (e0 >> 16) | (e1 & 0xffff0000u);
        }
        *reinterpret_cast<uint4*>(s + lds_off(xb * E + 2 * d, kb)) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint4*>(s + lds_off(xb * E + 2 * d + 1, kb)) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    }
}

template <typename T>
__global__ __launch_bounds__(NT_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn_kernel(KArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int kt0 = blockIdx.y * p.k_tiles_per_split;
    int kt1 = kt0 + p.k_tiles_per_split;
    kt1 = kt1 < p.k_tiles ? kt1 : p.k_tiles;
    if (kt0 >= kt1) return;

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    TnRegs<T> r;
    load_tn_regs<T>(p, r, m0, n0, kt0, tid);
    store_tn_regs<T>(r, smem, tid);
    for (int kt = kt0; kt < kt1; ++kt) {
        __syncthreads();
        char* cur = smem + ((kt - kt0) & 1) * STAGE_BYTES;
        char* nxt = smem + ((kt - kt0 + 1) & 1) * STAGE_BYTES;
        if (kt + 1 < kt1) load_tn_regs<T>(p, r, m0, n0, kt + 1, tid);
        compute_stage<T>(cur, cur + OPER_BYTES, acc, wm, wn, lane);
        if (kt + 1 < kt1) store_tn_regs<T>(r, nxt, tid);
    }
    epilogue<float, MTP_EPI_BIAS>(p, acc, m0, n0, wm, wn, lane);
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
    return 0;
}

template <typename T, typename Tout, int EPI>
int launch_nt(const mtp_gemm_args* a, hipStream_t stream) {
    constexpr int E = Elem<T>::kPerChunk;
    KArgs k;
    int rc = fill_common<T>(a, k);
    if (rc) return rc;
    if (a->K % E) return MTP_ERR_ARG;
    if (EPI == MTP_EPI_BIAS_RES && (!a->res || (a->res_ld % 4))) return MTP_ERR_ARG;
    if ((EPI == MTP_EPI_BIAS_GELU || EPI == MTP_EPI_DGELU) && (!a->aux || (a->aux_ld % 4))) return MTP_ERR_ARG;
    if (a->bias && a->bias_mod > 0 && (a->bias_mod % 4)) return MTP_ERR_ARG;
    const int tiles_m = (k.M + BM - 1) / BM;
    dim3 grid(tiles_m * k.tiles_n), block(NT_THREADS);
    const bool glds = (a->variant == 0) && (a->K % (8 * E) == 0);
    if (glds)
        hipLaunchKernelGGL((gemm_nt_kernel<T, Tout, EPI, true>), grid, block, LDS_BYTES, stream, k);
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
    k.atomic_out = split > 1;
    if (split > 1) {
        if (a->ldc != a->N) return MTP_ERR_ARG;
        hipError_t e = hipMemsetAsync(a->C, 0, sizeof(float) * (size_t)a->M * (size_t)a->N, stream);
        if (e != hipSuccess) return (int)e;
    }
    const int tiles_m = (k.M + BM - 1) / BM;
    dim3 grid(tiles_m * k.tiles_n, split), block(NT_THREADS);
    hipLaunchKernelGGL((gemm_tn_kernel<T>), grid, block, LDS_BYTES, stream, k);
    return mtp_launch_status();
}

template <typename T, typename Tout>
int dispatch_epi(const mtp_gemm_args* a, hipStream_t s) {
    switch (a->epilogue) {
        case MTP_EPI_BIAS: return launch_nt<T, Tout, MTP_EPI_BIAS>(a, s);
        case MTP_EPI_BIAS_GELU: return launch_nt<T, Tout, MTP_EPI_BIAS_GELU>(a, s);
        case MTP_EPI_DGELU: return launch_nt<T, Tout, MTP_EPI_DGELU>(a, s);
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

extern "C" int mtp_gemm_tn(const mtp_gemm_args* a, mtp_stream_t stream) {
    if (!a) return MTP_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    return a->in_dtype == MTP_BF16 ? launch_tn<bf16_t>(a, s) : launch_tn<float>(a, s);
}
