// Full (global) attention of the MTP backbone on token grids of at most 16 x 16 (the 224^2 configurations: 14 x 14 = 196 tokens) --
// Attention.forward VIT:90-111 with calc_rel_pos_spatial VIT:142-193 -- forward and backward, bf16 MFMA, gfx950.  Round 3 rewrite of
// the <= 256-token kernels of attn_full_mfma.hip, which spent their time on three LDS look-ups per (query, key) element (two table
// terms + the key's grid position), on fragments re-read from global memory inside the inner loops and on 16-token tiles that
// straddle image rows.  What is different here:
//
//  * ROW-ALIGNED TILES.  Tokens are laid out with a row pitch of 16: padded index 16 y + x (x < Wp valid, the rest zero rows), so an
//    MFMA tile of 16 tokens is exactly one image row: tile index = y, lane-in-tile = x.
//  * THE RELATIVE-POSITION LOGITS COME OUT OF THE MATRIX CORES.  For a query q = (yq, xq) the term q.Rh[yq - yk + Hp - 1] depends on
//    the key only through its row yk, q.Rw[xq - xk + Wp - 1] only through its column xk.  Per query that is one "bias row" of
//    16 + 16 numbers  A[q][0..15] = q.Rh[yq - s + Hp - 1],  A[q][16 + s] = q.Rw[xq - s + Wp - 1]  (and -30000 in the slots of the
//    padding columns xk >= Wp, which masks the padding keys for free).  With the one-hot key codes E[k][s] = [s == yk] + [s - 16 == xk]
//    the whole logit is ONE contraction over 64 + 32 + 32 slots:
//        S[q][k] = [ Q | A_hi | A_lo ][q] . [ K | E | E ][k]        (A split into two bf16 halves: 16 bits of mantissa)
//    i.e. two more MFMAs per 16 x 16 tile instead of 3 x 4 LDS look-ups per lane.  The bias rows of a tile are two small MFMAs against
//    the tables, re-arranged through a 3-KiB per-wave LDS tile once per 16 queries.
//  * The backward needs d(bias row) = dS . E: the per-row / per-column sums of dS are again MFMAs against one-hot operands, and both
//    the dq contribution and the table gradients follow from them as before (delta-indexed re-arrangement + MFMAs against R^T / Q^T).
//  * Every operand fragment comes from LDS images staged once per (image, head): K, V row-major XOR-swizzled (V^T / K^T / Q^T / dO^T
//    fragments through ds_read_b64_tr_b16); nothing is re-read from global memory inside a tile loop.
//
// Orientation tricks are the ones of attn_full_mfma.hip: S^T = K.Q^T so that a query's softmax is in-lane + two shuffles and P^T is
// directly the B operand of O^T = V^T.P^T; the backward uses both orientations (kernel A: lane = query, dQ and the table gradients;
// kernel B: lane = key, dK and dV).  Scale convention: logits = scale * S (VIT:100 scales q before both products).
#include "attn_mfma.h"
#include "attn_full_common.h"

namespace {

struct V3Geom {
    int Hp, Wp, N, heads, NPR, KK, RH, RW;   // NPR: rows of the LDS images (16 Hp rounded up to 32, + 16 zero rows), KK = pairs of row tiles
};

constexpr float V3_MASK = -30000.0f;   // logit (before the scale) of a padding key column
constexpr float V3_LSE_PAD = 1e30f;    // "lse" of a padding query: exp(s - lse) = 0

typedef short v3tr4_t __attribute__((ext_vector_type(4)));
// transposed fragment out of a row-major swizzled image: MFMA operand lane (fr = column 16 dt + fr of the image, gq) with the 8 k-slots
// = image rows row0 .. row0+3 and row0+16 .. row0+19  (row0 = 32 kk + 4 gq: the key / query order of the packed P^T / dS^T operands)
__device__ __forceinline__ uint4 v3_frag_tr(const char* img, int row0, int dt, int fr) {
    const int c = 16 * dt + 4 * (fr & 3);
    const int ra = row0 + (fr >> 2), rb = ra + 16;
    const int oa = ra * 128 + ((((c >> 3) ^ (ra & 7))) << 4) + (c & 7) * 2, ob = rb * 128 + ((((c >> 3) ^ (rb & 7))) << 4) + (c & 7) * 2;
    const v3tr4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v3tr4_t*)(img + oa));
    const v3tr4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v3tr4_t*)(img + ob));
    const uint2 l = __builtin_bit_cast(uint2, lo), h = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, h.x, h.y);
}

// 64-wide rows of a token-major tensor -> padded swizzled image: image row 16 y + x = token y Wp + x, everything else zero
__device__ __forceinline__ void v3_stage(const bf16_t* __restrict__ src, int64_t ld, const V3Geom& g, char* img, int tid, int nthreads, int rows = 0) {
    for (int idx = tid; idx < (rows ? rows : g.NPR) * 8; idx += nthreads) {
        const int row = idx >> 3, c = idx & 7, y = row >> 4, x = row & 15;
        const bool ok = y < g.Hp && x < g.Wp;
        *reinterpret_cast<uint4*>(img + swz(row, c)) = row_frag(src, ld, ok ? y * g.Wp + x : 0, ok, 8 * c);
    }
}

// The same in two halves: ALL loads of a workgroup's images are issued before the first LDS store (v3_stage alone is a loop of
// load -> wait -> store round trips: 7 serialized global latencies per image with 256 threads, a third of the first version's time).
template <int NTH, int IT>
__device__ __forceinline__ void v3_stage_load(const bf16_t* __restrict__ src, int64_t ld, const V3Geom& g, int rows, int tid, uint4 (&v)[IT]) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NTH, row = idx >> 3, c = idx & 7, y = row >> 4, x = row & 15;
        const bool ok = idx < rows * 8 && y < g.Hp && x < g.Wp;
        v[i] = row_frag(src, ld, ok ? y * g.Wp + x : 0, ok, 8 * c);
    }
}
template <int NTH, int IT>
__device__ __forceinline__ void v3_stage_store(char* img, int rows, int tid, const uint4 (&v)[IT]) {
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NTH, row = idx >> 3, c = idx & 7;
        if (idx < rows * 8) *reinterpret_cast<uint4*>(img + swz(row, c)) = v[i];
    }
}

// a rel-pos table (rows x 64 f32) -> 32-row bf16 swizzled LDS image (4 KiB), rows >= `rows` zero
__device__ __forceinline__ void v3_stage_table(const float* __restrict__ tab, int rows, char* img, int tid, int nthreads) {
    for (int idx = tid; idx < 32 * 8; idx += nthreads) {
        const int row = idx >> 3, c = idx & 7;
        *reinterpret_cast<uint4*>(img + swz(row, c)) = table_frag(tab, row, rows, 8 * c);
    }
}

// one-hot key codes of key tile (image row) kt as an MFMA operand: lane (fr = key column, gq), slots 8 gq .. 8 gq + 7 of
// [s == kt] (s < 16) | [s - 16 == fr]
__device__ __forceinline__ uint4 v3_ecode(int kt, int fr, int gq) {
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    const int e = gq < 2 ? kt - 8 * gq : fr - 8 * (gq - 2);
    const uint32_t one = 0x3f80u << ((e & 1) * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (e >= 0 && e < 8 && (e >> 1) == i) ? one : 0u;
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// bias rows of query tile y (see the header): qf = the tile's Q fragments (lane (fr = query column, gq): 16-B chunk ks * 4 + gq of the row),
// tw = rel_w table fragments [row tile][k step]; xr = this wave's 3-KiB exchange tile.  Returns the two bf16 halves as MFMA operands
// (lane (fr, gq): slots 8 gq .. 8 gq + 7).
// (RhI / RwI: the two tables as 32-row bf16 LDS images, rows beyond the table zero -- v3_stage_table)
__device__ __forceinline__ void v3_bias_rows(const V3Geom& g, const char* RhI, const char* RwI, const uint4 (&qf)[2], float* xr, int y, int fr,
                                             int gq, uint4& hi, uint4& lo) {
    float* Hx = xr;              // [16 slots hk][16 queries]
    float* Wx = xr + 256;        // [32 delta][16 queries]
    f32x4_t ah = {0.f, 0.f, 0.f, 0.f};
    const int hrow = fr < g.Hp ? y - fr + g.Hp - 1 : 31;      // table row of slot hk = fr (lane fr of the A operand = slot); row 31 is zero
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) ah = mma(ld16(RhI + swz(hrow, ks * 4 + gq)), qf[ks], ah);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Hx[(4 * gq + rr) * 16 + fr] = ah[rr];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f32x4_t aw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) aw = mma(ld16(RwI + swz(16 * rt + fr, ks * 4 + gq)), qf[ks], aw);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Wx[(16 * rt + 4 * gq + rr) * 16 + fr] = aw[rr];
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int s = 8 * (gq & 1) + e;                     // slot inside its half: hk (gq < 2) or wk (gq >= 2)
        const int dlt = fr + g.Wp - 1 - s;                  // rel_w row of (query column fr, key column s)
        const bool wok = s < g.Wp && fr < g.Wp;
        const float t = gq < 2 ? Hx[s * 16 + fr] : Wx[(wok ? dlt : 0) * 16 + fr];
        v[e] = gq < 2 ? t : (s < g.Wp ? (wok ? t : 0.f) : V3_MASK);
    }
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = v[e] - bf16_bits_to_f32(f32_to_bf16_bits(v[e]));
    hi = pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    lo = pack_bf16x8(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
}

// ===================================================================================================================
// forward: one workgroup (NW waves) per (image, head); wave = query tiles (image rows) w, w + NW, ...
// dynamic LDS: Ks | Vs (NPR x 128 each) | RhI | RwI (4 KiB each) | xr[NW][768] f32
// ===================================================================================================================
// HPT = the number of image rows when known at compile time (14: the 224^2 configurations), 0 = read it from the geometry.  With a runtime
// row count every key-tile step sits in its own branch, which pins its LDS reads and its chain of four dependent MFMAs between two waits;
// with a constant the compiler batches the reads of all tiles and interleaves their independent accumulation chains.
template <int NW, int HPT>
__global__ __launch_bounds__(64 * NW, 2) void v3_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                         const float* __restrict__ rel_h, const float* __restrict__ rel_w, V3Geom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int NR = g.NPR - 16;      // (the forward's transposed V fragments stay inside 32 KK rows)
    char* Ks = sm;
    char* Vs = Ks + NR * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, gq = lane >> 4;
    char* RhI = Vs + NR * 128;
    char* RwI = RhI + 4096;
    float* xr = reinterpret_cast<float*>(RwI + 4096) + wave * 768;
    const int bh = blockIdx.x, b = bh / g.heads, h = bh % g.heads;
    const int C = g.heads * HD, N = g.N;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;

    // this wave's first Q tile: issued before the staging loads and the barrier, the next tile's at the top of each iteration (the
    // fragments of a tile are one dependent global round trip otherwise: 47 % of the wave cycles of the first version were waits)
    const bool nv = fr < g.Wp;
    uint4 qn[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qn[ks] = row_frag(base, ld, wave * g.Wp + (nv ? fr : 0), nv && wave < g.Hp, ks * 32 + gq * 8);
    {
        constexpr int IT = (256 * 8 + 64 * NW - 1) / (64 * NW);     // NR <= 256 rows
        uint4 vk[IT], vv[IT];
        v3_stage_load<64 * NW, IT>(base + C, ld, g, NR, tid, vk);
        v3_stage_load<64 * NW, IT>(base + 2 * C, ld, g, NR, tid, vv);
        v3_stage_table(rel_h, g.RH, RhI, tid, 64 * NW);
        v3_stage_table(rel_w, g.RW, RwI, tid, 64 * NW);
        v3_stage_store<64 * NW, IT>(Ks, NR, tid, vk);
        v3_stage_store<64 * NW, IT>(Vs, NR, tid, vv);
    }
    __syncthreads();

    constexpr int KTMAX = HPT ? HPT : 16, KKMAX = (KTMAX + 1) / 2;
    for (int y = wave; y < g.Hp; y += NW) {
        const int tok = y * g.Wp + (nv ? fr : 0);
        uint4 qf[2], ahi, alo;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = qn[ks];
            qn[ks] = row_frag(base, ld, (y + NW) * g.Wp + (nv ? fr : 0), nv && y + NW < g.Hp, ks * 32 + gq * 8);
        }
        v3_bias_rows(g, RhI, RwI, qf, xr, y, fr, gq, ahi, alo);
        f32x4_t s[2 * KKMAX];
        float m = -INFINITY;
        if constexpr (KTMAX & 1) s[KTMAX] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KTMAX; ++kt) {
            s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (HPT || kt < g.Hp) {
                const uint4 ec = v3_ecode(kt, fr, gq);
                s[kt] = mma(ld16(Ks + swz(16 * kt + fr, gq)), qf[0], s[kt]);
                s[kt] = mma(ld16(Ks + swz(16 * kt + fr, 4 + gq)), qf[1], s[kt]);
                s[kt] = mma(ec, ahi, s[kt]);
                s[kt] = mma(ec, alo, s[kt]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s[kt][r] *= scale;
                    m = fmaxf(m, s[kt][r]);
                }
            }
        }
        m = xor16_max(m);
        m = xor32_max(m);
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < KTMAX; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (HPT || kt < g.Hp) ? __expf(s[kt][r] - m) : 0.f;     // (padding key columns: scale * -30000 -> 0)
                s[kt][r] = p;
                l += p;
            }
        l = xor16_sum(l);
        l = xor32_sum(l);
        f32x4_t oa[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oa[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KKMAX; ++kk) {
            if (HPT || kk < g.KK) {
                const uint4 pf = pack_bf16x8(s[2 * kk][0], s[2 * kk][1], s[2 * kk][2], s[2 * kk][3], s[2 * kk + 1][0], s[2 * kk + 1][1], s[2 * kk + 1][2], s[2 * kk + 1][3]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oa[dt] = mma(v3_frag_tr(Vs, 32 * kk + 4 * gq, dt, fr), pf, oa[dt]);
            }
        }
        if (nv) {
            const float inv = 1.0f / l;
            bf16_t* op = o + ((int64_t)b * N + tok) * C + h * HD + 4 * gq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store4(op + 16 * dt, make_float4(oa[dt][0] * inv, oa[dt][1] * inv, oa[dt][2] * inv, oa[dt][3] * inv));
            if (gq == 0) lse[(int64_t)bh * N + tok] = m + __logf(l);
        }
    }
}

// ===================================================================================================================
// backward A: dQ and the rel-pos table gradients.  wave = query tiles.
// dynamic LDS: Ks | Vs | Qs (NPR x 128 each) | RhI | RwI (4 KiB each) | xr[NW][768 + 512] f32
// Register budget (2 waves per SIMD = 256): the dS^T of a pair of key tiles is consumed as soon as it exists (no softmax pass: the
// forward's lse is known), the table fragments come out of the LDS images (R^T through the transpose read), so what stays live across
// a query tile is dq (16) + the table-gradient accumulators (64).
// ===================================================================================================================
template <int NW, int HPT>
__global__ __launch_bounds__(64 * NW) void v3_bwd_a_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ lse, bf16_t* __restrict__ dqkv, const float* __restrict__ rel_h,
                                                           const float* __restrict__ rel_w, float* __restrict__ drel_part, V3Geom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* Ks = sm;
    char* Vs = Ks + g.NPR * 128;
    char* Qs = Vs + g.NPR * 128;
    char* RhI = Qs + g.NPR * 128;
    char* RwI = RhI + 4096;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, gq = lane >> 4;
    float* xr = reinterpret_cast<float*>(RwI + 4096) + wave * 1280;
    float* dHx = xr + 768;       // [16 hk][16 queries]
    float* dWx = xr + 1024;      // [16 wk][16 queries]
    const int bh = blockIdx.x, b = bh / g.heads, h = bh % g.heads;
    const int C = g.heads * HD, N = g.N;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* dob = dout + (int64_t)b * N * C + h * HD;
    const bf16_t* ob = o + (int64_t)b * N * C + h * HD;

    // this wave's first dO / O tile and lse: in flight under the staging loads and the barrier; later tiles one iteration ahead
    const bool nv = fr < g.Wp;
    uint4 don[2], on[2];
    float lsn;
    {
        const bool ok0 = nv && wave < g.Hp;
        const int tok0 = wave * g.Wp + (nv ? fr : 0);
        lsn = lse[(int64_t)bh * N + (ok0 ? tok0 : 0)];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            don[ks] = row_frag(dob, C, tok0, ok0, ks * 32 + gq * 8);
            on[ks] = row_frag(ob, C, tok0, ok0, ks * 32 + gq * 8);
        }
    }
    {
        constexpr int IT = (272 * 8 + 64 * NW - 1) / (64 * NW);     // NPR <= 272 rows
        uint4 vk[IT], vv[IT], vq[IT];
        v3_stage_load<64 * NW, IT>(base + C, ld, g, g.NPR, tid, vk);
        v3_stage_load<64 * NW, IT>(base + 2 * C, ld, g, g.NPR, tid, vv);
        v3_stage_load<64 * NW, IT>(base, ld, g, g.NPR, tid, vq);
        v3_stage_table(rel_h, g.RH, RhI, tid, 64 * NW);
        v3_stage_table(rel_w, g.RW, RwI, tid, 64 * NW);
        v3_stage_store<64 * NW, IT>(Ks, g.NPR, tid, vk);
        v3_stage_store<64 * NW, IT>(Vs, g.NPR, tid, vv);
        v3_stage_store<64 * NW, IT>(Qs, g.NPR, tid, vq);
    }
    f32x4_t tacc[2][2][4];   // [table][row tile][d tile]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) tacc[t][rt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // one-hot column selector of the packed dS^T operand (slots 0-3 = key tile 2 kk, rows 4 gq + e; slots 4-7 = key tile 2 kk + 1):
    // lane (fr = wk, gq) has ones where the slot's key column 4 gq + (e & 3) equals fr
    uint4 ewT;
    {
        const int j = fr & 3;
        const uint32_t one = 0x3f80u << ((j & 1) * 16), hit = (fr >> 2) == gq ? one : 0u;
        ewT = make_uint4((j >> 1) == 0 ? hit : 0u, (j >> 1) == 1 ? hit : 0u, (j >> 1) == 0 ? hit : 0u, (j >> 1) == 1 ? hit : 0u);
    }
    __syncthreads();

    for (int y = wave; y < g.Hp; y += NW) {
        const int tok = y * g.Wp + (nv ? fr : 0);
        uint4 qf[2], dof[2], ahi, alo;
        float dl = 0.f;
        const float ls = nv ? lsn : V3_LSE_PAD;
        {   // take the prefetched dO / O fragments, issue the next tile's
            const bool more = nv && y + NW < g.Hp;
            const int tokn = (y + NW) * g.Wp + (nv ? fr : 0);
            lsn = lse[(int64_t)bh * N + (more ? tokn : 0)];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                qf[ks] = ld16(Qs + swz(16 * y + fr, ks * 4 + gq));
                dof[ks] = don[ks];
                const uint4 of = on[ks];
                don[ks] = row_frag(dob, C, tokn, more, ks * 32 + gq * 8);
                on[ks] = row_frag(ob, C, tokn, more, ks * 32 + gq * 8);
                const uint32_t a[4] = {dof[ks].x, dof[ks].y, dof[ks].z, dof[ks].w}, c[4] = {of.x, of.y, of.z, of.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dl += bf16_bits_to_f32(a[e] & 0xffffu) * bf16_bits_to_f32(c[e] & 0xffffu) + bf16_bits_to_f32(a[e] >> 16) * bf16_bits_to_f32(c[e] >> 16);
            }
        }
        dl = xor16_sum(dl);
        dl = xor32_sum(dl);
        v3_bias_rows(g, RhI, RwI, qf, xr, y, fr, gq, ahi, alo);

        // d(qs)^T = K^T.dS^T (+ table terms below); row / column sums of dS^T = gradients of the bias rows
        f32x4_t dq[4], dh = {0.f, 0.f, 0.f, 0.f}, dw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // two key-tile pairs per loop iteration (the independent chains of one pair cover the latencies of the other); a fully unrolled
        // loop makes hipcc hoist every LDS read of the tile and spill
        auto pair_step = [&](int kk) __attribute__((always_inline)) {
            float ds[2][4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kt = 2 * kk + half;              // (a key tile beyond the grid is a zero image: s = bias only, but its E code has no
                const uint4 ec = v3_ecode(kt, fr, gq);     //  row slot < Hp ... so mask it explicitly below)
                f32x4_t sT = {0.f, 0.f, 0.f, 0.f}, dpT = {0.f, 0.f, 0.f, 0.f};
                sT = mma(ld16(Ks + swz(16 * kt + fr, gq)), qf[0], sT);
                sT = mma(ld16(Ks + swz(16 * kt + fr, 4 + gq)), qf[1], sT);
                sT = mma(ec, ahi, sT);
                sT = mma(ec, alo, sT);
                dpT = mma(ld16(Vs + swz(16 * kt + fr, gq)), dof[0], dpT);
                dpT = mma(ld16(Vs + swz(16 * kt + fr, 4 + gq)), dof[1], dpT);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(fminf(scale * sT[r] - ls, 30.f));
                    ds[half][r] = ((HPT && kt < HPT) || (!HPT && kt < g.Hp)) ? p * (dpT[r] - dl) : 0.f;
                }
            }
            const uint4 dsf = pack_bf16x8(ds[0][0], ds[0][1], ds[0][2], ds[0][3], ds[1][0], ds[1][1], ds[1][2], ds[1][3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = mma(v3_frag_tr(Ks, 32 * kk + 4 * gq, dt, fr), dsf, dq[dt]);
            const uint32_t o2 = 0x3f803f80u;
            const uint4 ehT = make_uint4(fr == 2 * kk ? o2 : 0u, fr == 2 * kk ? o2 : 0u, fr == 2 * kk + 1 ? o2 : 0u, fr == 2 * kk + 1 ? o2 : 0u);
            dh = mma(ehT, dsf, dh);      // lane (fr = query, gq): d(bias row)[hk = 4 gq + r]
            dw = mma(ewT, dsf, dw);      //                        d(bias row)[16 + wk], wk = 4 gq + r
        };
        if constexpr (HPT != 0) {
            constexpr int nkk = (HPT + 1) / 2;
#pragma unroll 1
            for (int k2 = 0; k2 + 1 < nkk; k2 += 2) {
                pair_step(k2);
                pair_step(k2 + 1);
            }
            if (nkk & 1) pair_step(nkk - 1);
        } else {
#pragma unroll 1
            for (int kk = 0; kk < g.KK; ++kk) pair_step(kk);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dHx[(4 * gq + r) * 16 + fr] = dh[r];
            dWx[(4 * gq + r) * 16 + fr] = dw[r];
        }
        // delta-indexed view: dQR_h[dlt][q] = dH[y + Hp - 1 - dlt][q],  dQR_w[dlt][q] = dW[xq + Wp - 1 - dlt][q].  Slot order of the
        // operands = what the transpose read of the table images delivers: lane group gq holds dlt = 4 gq .. 4 gq + 3 and 16 + 4 gq .. + 3
        {
            float e[8], f[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int dlt = 4 * gq + (x & 3) + (x >> 2) * 16;
                const int hk = y + g.Hp - 1 - dlt, wk = fr + g.Wp - 1 - dlt;
                const bool hok = hk >= 0 && hk < g.Hp, wok = wk >= 0 && wk < g.Wp && nv;
                const float a = dHx[(hok ? hk : 0) * 16 + fr], c = dWx[(wok ? wk : 0) * 16 + fr];
                e[x] = hok ? a : 0.f;
                f[x] = wok ? c : 0.f;
            }
            const uint4 eh = pack_bf16x8(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
            const uint4 ew = pack_bf16x8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dq[dt] = mma(v3_frag_tr(RhI, 4 * gq, dt, fr), eh, dq[dt]);      // A[d][dlt] = Rh[dlt][d]
                dq[dt] = mma(v3_frag_tr(RwI, 4 * gq, dt, fr), ew, dq[dt]);
            }
        }
        if (nv) {
            bf16_t* dp = dqkv + ((int64_t)b * N + tok) * ld + h * HD + 4 * gq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store4(dp + 16 * dt, make_float4(dq[dt][0] * scale, dq[dt][1] * scale, dq[dt][2] * scale, dq[dt][3] * scale));
        }
        // table gradients: tacc[t][rt][dt] += dQR_t[dlt = 16 rt + fr][q] . Q[q][d] over the 16 queries of this tile.  Operand slots: lane group
        // gq holds queries 4 gq .. 4 gq + 3 in slots 0-3 (slots 4-7 = the next image row in the Q^T fragment: zero in the A operand)
        {
            uint4 bq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) bq[dt] = v3_frag_tr(Qs, 16 * y + 4 * gq, dt, fr);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float vh[4], vw[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int q = 4 * gq + x, dlt = 16 * rt + fr;
                    const int hk = y + g.Hp - 1 - dlt, wk = q + g.Wp - 1 - dlt;
                    const bool hok = hk >= 0 && hk < g.Hp, wok = wk >= 0 && wk < g.Wp && q < g.Wp;
                    const float a = dHx[(hok ? hk : 0) * 16 + q], c = dWx[(wok ? wk : 0) * 16 + q];
                    vh[x] = hok ? a : 0.f;
                    vw[x] = wok ? c : 0.f;
                }
                const uint4 ah = make_uint4(pack_bf16x2(vh[0], vh[1]), pack_bf16x2(vh[2], vh[3]), 0u, 0u);
                const uint4 aw = make_uint4(pack_bf16x2(vw[0], vw[1]), pack_bf16x2(vw[2], vw[3]), 0u, 0u);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    tacc[0][rt][dt] = mma(ah, bq[dt], tacc[0][rt][dt]);
                    tacc[1][rt][dt] = mma(aw, bq[dt], tacc[1][rt][dt]);
                }
            }
        }
    }
    // table partials of the NW waves -> ONE image per (image, head): tree reduction through LDS (the Q / K / V images are dead by now; the exchange buffer holds
    // register images -- 16 float4 per lane, lane-contiguous 16-byte accesses, no bank conflicts), then plain stores by wave 0.  Round 5 had every wave add its
    // partials to global memory with f32 atomics: 8 x 14 MB of read-modify-write per launch where 14 MB of stores are owed (profiles/r05_hbm_fractions.txt: 1.74 x the
    // algorithmic bytes), plus the clearing pass in front of it.
    __syncthreads();
    float4* xb = reinterpret_cast<float4*>(sm);
#pragma unroll
    for (int half = NW / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            float4* slot = xb + (wave - half) * 1024 + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        slot[((t * 2 + rt) * 4 + dt) * 64] = make_float4(tacc[t][rt][dt][0], tacc[t][rt][dt][1], tacc[t][rt][dt][2], tacc[t][rt][dt][3]);
        }
        __syncthreads();
        if (wave < half) {
            const float4* slot = xb + wave * 1024 + lane;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const float4 v = slot[((t * 2 + rt) * 4 + dt) * 64];
                        tacc[t][rt][dt][0] += v.x; tacc[t][rt][dt][1] += v.y; tacc[t][rt][dt][2] += v.z; tacc[t][rt][dt][3] += v.w;
                    }
        }
        if (half > 1) __syncthreads();
    }
    if (wave == 0) {
        float* dp = drel_part + (int64_t)bh * (g.RH + g.RW) * HD;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int r = 16 * rt + 4 * gq + rr;
                        if (r < (t ? g.RW : g.RH)) dp[((t ? g.RH : 0) + r) * HD + 16 * dt + fr] = tacc[t][rt][dt][rr] * scale;
                    }
    }
}

// ===================================================================================================================
// backward B: dK, dV.  wave = key tiles.
// dynamic LDS: Qs | dOs | QA (NPR x 128 each; QA row = [16 slots H | 16 slots W] hi, then lo) | RhI | RwI | lses[NPR] | delta[NPR] | xr[NW][768] f32
// ===================================================================================================================
template <int NW, int HPT>
__global__ __launch_bounds__(64 * NW) void v3_bwd_b_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                           const float* __restrict__ lse, bf16_t* __restrict__ dqkv, const float* __restrict__ rel_h,
                                                           const float* __restrict__ rel_w, V3Geom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* Qs = sm;
    char* dOs = Qs + g.NPR * 128;
    char* QA = dOs + g.NPR * 128;
    char* RhI = QA + g.NPR * 128;
    char* RwI = RhI + 4096;
    float* lses = reinterpret_cast<float*>(RwI + 4096);
    float* delta = lses + g.NPR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, gq = lane >> 4;
    float* xr = delta + g.NPR + wave * 768;
    const int bh = blockIdx.x, b = bh / g.heads, h = bh % g.heads;
    const int C = g.heads * HD, N = g.N;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* dob = dout + (int64_t)b * N * C + h * HD;
    const bf16_t* ob = o + (int64_t)b * N * C + h * HD;

    // this wave's first K / V tile: in flight under the staging; later tiles one iteration ahead
    const bool kv = fr < g.Wp;
    uint4 kn[2], vn[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kn[ks] = row_frag(base + C, ld, wave * g.Wp + (kv ? fr : 0), kv && wave < g.Hp, ks * 32 + gq * 8);
        vn[ks] = row_frag(base + 2 * C, ld, wave * g.Wp + (kv ? fr : 0), kv && wave < g.Hp, ks * 32 + gq * 8);
    }
    {
        constexpr int IT = (272 * 8 + 64 * NW - 1) / (64 * NW);
        uint4 vq[IT], vd[IT];
        v3_stage_load<64 * NW, IT>(base, ld, g, g.NPR, tid, vq);
        v3_stage_load<64 * NW, IT>(dob, C, g, g.NPR, tid, vd);
        v3_stage_table(rel_h, g.RH, RhI, tid, 64 * NW);
        v3_stage_table(rel_w, g.RW, RwI, tid, 64 * NW);
        v3_stage_store<64 * NW, IT>(Qs, g.NPR, tid, vq);
        v3_stage_store<64 * NW, IT>(dOs, g.NPR, tid, vd);
    }
    for (int row = tid; row < g.NPR; row += 64 * NW) {
        const int y = row >> 4, x = row & 15;
        const bool ok = y < g.Hp && x < g.Wp;
        const int tok = ok ? y * g.Wp + x : 0;
        float dl = 0.f;
#pragma unroll
        for (int i = 0; i < HD / 8; ++i) {
            float a[8], c[8];
            load8(dob + (int64_t)tok * C + 8 * i, a);
            load8(ob + (int64_t)tok * C + 8 * i, c);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += a[e] * c[e];
        }
        const float ls = lse[(int64_t)bh * N + tok];
        delta[row] = ok ? dl : 0.f;
        lses[row] = ok ? ls : V3_LSE_PAD;
    }
    // rows of QA beyond the grid (y >= Hp) are zero: their queries have lse = V3_LSE_PAD anyway
    for (int idx = tid; idx < (g.NPR - 16 * g.Hp) * 8; idx += 64 * NW) *reinterpret_cast<uint4*>(QA + 16 * g.Hp * 128 + idx * 16) = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    for (int y = wave; y < g.Hp; y += NW) {       // bias rows of every query tile -> QA image (hi: chunks 0-3, lo: chunks 4-7)
        uint4 qf[2], ahi, alo;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = ld16(Qs + swz(16 * y + fr, ks * 4 + gq));
        v3_bias_rows(g, RhI, RwI, qf, xr, y, fr, gq, ahi, alo);
        *reinterpret_cast<uint4*>(QA + swz(16 * y + fr, gq)) = ahi;
        *reinterpret_cast<uint4*>(QA + swz(16 * y + fr, 4 + gq)) = alo;
    }
    __syncthreads();

    for (int kt = wave; kt < g.Hp; kt += NW) {
        const int tok = kt * g.Wp + (kv ? fr : 0);
        uint4 kfb[2], vfb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bool more = kv && kt + NW < g.Hp;
            const int tokn = (kt + NW) * g.Wp + (kv ? fr : 0);
            kfb[ks] = kn[ks];
            vfb[ks] = vn[ks];
            kn[ks] = row_frag(base + C, ld, tokn, more, ks * 32 + gq * 8);
            vn[ks] = row_frag(base + 2 * C, ld, tokn, more, ks * 32 + gq * 8);
        }
        const uint4 ec = v3_ecode(kt, fr, gq);
        f32x4_t dks[4], dvs[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dks[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dvs[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        auto pair_step = [&](int kk) __attribute__((always_inline)) {
            float pv[2][4], dv[2][4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qrow = 16 * (2 * kk + half) + fr;       // image row of the A fragments (lane fr = query column)
                f32x4_t sB = {0.f, 0.f, 0.f, 0.f}, dpB = {0.f, 0.f, 0.f, 0.f};
                sB = mma(ld16(Qs + swz(qrow, gq)), kfb[0], sB);            // D[query 4 gq + r of the tile][key fr]
                sB = mma(ld16(Qs + swz(qrow, 4 + gq)), kfb[1], sB);
                sB = mma(ld16(QA + swz(qrow, gq)), ec, sB);
                sB = mma(ld16(QA + swz(qrow, 4 + gq)), ec, sB);
                dpB = mma(ld16(dOs + swz(qrow, gq)), vfb[0], dpB);
                dpB = mma(ld16(dOs + swz(qrow, 4 + gq)), vfb[1], dpB);
                const float4 l4 = *reinterpret_cast<const float4*>(lses + 16 * (2 * kk + half) + 4 * gq);
                const float4 d4 = *reinterpret_cast<const float4*>(delta + 16 * (2 * kk + half) + 4 * gq);
                const float lsv[4] = {l4.x, l4.y, l4.z, l4.w}, dlv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(fminf(scale * sB[r] - lsv[r], 30.f));
                    pv[half][r] = p;
                    dv[half][r] = p * (dpB[r] - dlv[r]);
                }
            }
            const uint4 pfb = pack_bf16x8(pv[0][0], pv[0][1], pv[0][2], pv[0][3], pv[1][0], pv[1][1], pv[1][2], pv[1][3]);
            const uint4 dsfb = pack_bf16x8(dv[0][0], dv[0][1], dv[0][2], dv[0][3], dv[1][0], dv[1][1], dv[1][2], dv[1][3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dks[dt] = mma(v3_frag_tr(Qs, 32 * kk + 4 * gq, dt, fr), dsfb, dks[dt]);   // D[d][key fr]
                dvs[dt] = mma(v3_frag_tr(dOs, 32 * kk + 4 * gq, dt, fr), pfb, dvs[dt]);
            }
        };
        if constexpr (HPT != 0) {
            constexpr int nkk = (HPT + 1) / 2;
#pragma unroll 1
            for (int k2 = 0; k2 + 1 < nkk; k2 += 2) {
                pair_step(k2);
                pair_step(k2 + 1);
            }
            if (nkk & 1) pair_step(nkk - 1);
        } else {
#pragma unroll 1
            for (int kk = 0; kk < g.KK; ++kk) pair_step(kk);
        }
        if (kv) {
            bf16_t* dk = dqkv + ((int64_t)b * N + tok) * ld + C + h * HD + 4 * gq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                store4(dk + 16 * dt, make_float4(dks[dt][0] * scale, dks[dt][1] * scale, dks[dt][2] * scale, dks[dt][3] * scale));
                store4(dk + C + 16 * dt, make_float4(dvs[dt][0], dvs[dt][1], dvs[dt][2], dvs[dt][3]));
            }
        }
    }
}

bool v3_geom(int64_t Hp, int64_t Wp, int64_t heads, V3Geom& g) {
    if (Hp < 1 || Wp < 1 || Hp > 16 || Wp > 16) return false;
    g.Hp = (int)Hp; g.Wp = (int)Wp; g.N = (int)(Hp * Wp); g.heads = (int)heads;
    g.KK = (g.Hp + 1) / 2;
    g.NPR = 32 * g.KK + 16;       // (+16 zero rows: the transposed fragments of the last tile reach one row tile further)
    g.RH = 2 * g.Hp - 1;
    g.RW = 2 * g.Wp - 1;
    return true;
}

}  // namespace

bool mtp_full_v3_fits(int64_t Hp, int64_t Wp) {
    return Hp >= 1 && Wp >= 1 && Hp <= 16 && Wp <= 16;
}

int mtp_full_v3_fwd_launch(const void* qkv, void* o, float* lse, const float* rel_h, const float* rel_w, int64_t B, int64_t Hp, int64_t Wp, int64_t heads,
                           float scale, hipStream_t s) {
    V3Geom g;
    if (!v3_geom(Hp, Wp, heads, g)) return MTP_ERR_UNSUPPORTED;
    constexpr int NW = 4;        // 2 x 28 KiB images + tables + 12 KiB: two workgroups (8 waves) per CU at 14 x 14
    const size_t lds = 2 * (size_t)(g.NPR - 16) * 128 + 8192 + (size_t)NW * 768 * 4;      // 76 KiB at 14 x 14
    const dim3 grid((unsigned)(B * heads)), block(64 * NW);
    if (Hp == 14) {
        (void)hipFuncSetAttribute((const void*)v3_fwd_kernel<NW, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((v3_fwd_kernel<NW, 14>), grid, block, lds, s, (const bf16_t*)qkv, (bf16_t*)o, lse, rel_h, rel_w, g, scale);
    } else {
        (void)hipFuncSetAttribute((const void*)v3_fwd_kernel<NW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((v3_fwd_kernel<NW, 0>), grid, block, lds, s, (const bf16_t*)qkv, (bf16_t*)o, lse, rel_h, rel_w, g, scale);
    }
    return mtp_launch_status();
}

int mtp_full_v3_bwd_launch(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, const float* rel_h, const float* rel_w,
                           float* drel_part, int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s) {
    V3Geom g;
    if (!v3_geom(Hp, Wp, heads, g)) return MTP_ERR_UNSUPPORTED;
    // (drel_part needs no clearing: v3_bwd_a_kernel stores every row of every (image, head) image once)
    constexpr int NW = 8;
    const size_t lds_a = 3 * (size_t)g.NPR * 128 + 8192 + (size_t)NW * 1280 * 4;      // >= 64 KiB for every grid: the table-partial exchange buffer (NW / 2 x 16 KiB) fits
    static_assert(NW == 8, "the exchange buffer of v3_bwd_a_kernel's table reduction is sized for eight waves");
    const size_t lds_b = 3 * (size_t)g.NPR * 128 + 8192 + 2 * (size_t)g.NPR * 4 + (size_t)NW * 768 * 4;
    const dim3 grid((unsigned)(B * heads)), block(64 * NW);
    if (Hp == 14) {
        (void)hipFuncSetAttribute((const void*)v3_bwd_a_kernel<NW, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
        (void)hipFuncSetAttribute((const void*)v3_bwd_b_kernel<NW, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        hipLaunchKernelGGL((v3_bwd_a_kernel<NW, 14>), grid, block, lds_a, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse, (bf16_t*)dqkv, rel_h,
                           rel_w, drel_part, g, scale);
        hipLaunchKernelGGL((v3_bwd_b_kernel<NW, 14>), grid, block, lds_b, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse, (bf16_t*)dqkv, rel_h,
                           rel_w, g, scale);
    } else {
        (void)hipFuncSetAttribute((const void*)v3_bwd_a_kernel<NW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
        (void)hipFuncSetAttribute((const void*)v3_bwd_b_kernel<NW, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        hipLaunchKernelGGL((v3_bwd_a_kernel<NW, 0>), grid, block, lds_a, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse, (bf16_t*)dqkv, rel_h,
                           rel_w, drel_part, g, scale);
        hipLaunchKernelGGL((v3_bwd_b_kernel<NW, 0>), grid, block, lds_b, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse, (bf16_t*)dqkv, rel_h,
                           rel_w, g, scale);
    }
    return mtp_launch_status();
}
