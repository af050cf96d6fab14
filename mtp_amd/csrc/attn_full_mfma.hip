// bf16 MFMA kernels for the full (global) attention blocks of the MTP backbone (Attention.forward, VIT:90-111, with the
// decomposed relative-position terms of calc_rel_pos_spatial, VIT:142-193), gfx950, head_dim 64, N = Hp*Wp <= 256 tokens.
//
// One workgroup per (image, head) (forward: 4 waves; backward A / B: 8 waves where the LDS allows); K (row-major, XOR-swizzled) and
// V^T (transposed image) live in LDS; each wave owns 16-query (or 16-key) tiles.  Same tricks as the RVSA kernels (attn_mfma.hip): S^T = K.Q^T so a query's softmax is
// in-lane + two shuffles and P is directly the B operand of O^T = V^T.P^T; the q.Rh / q.Rw terms are one MFMA against the
// tables, exchanged through a per-wave LDS tile; the backward uses both MFMA orientations instead of transposing P/dS:
//   kernel A (lane: query, 4 keys)  : dQ^T = K^T.dS^T + Rh^T.dQRh + Rw^T.dQRw,  d(rel_pos_h/w) partials
//   kernel B (lane: key, 4 queries) : dK^T = Q^T.dS,  dV^T = dO^T.P
#include "attn_mfma.h"
#include "attn_full_common.h"

namespace {

// ===================================================================================================================
// forward.  dynamic LDS: Ks[16NT*128] | Vt[64*TPV] | QR[4 waves][64][16] f32 | kpos[NP2] u32
// ===================================================================================================================
__global__ __launch_bounds__(256) void full_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                           const float* __restrict__ rel_h, const float* __restrict__ rel_w, FGeom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* Ks = sm;
    char* Vt = Ks + g.NT * 16 * 128;
    float* QRall = reinterpret_cast<float*>(Vt + 64 * g.TPV);
    uint32_t* kpos = reinterpret_cast<uint32_t*>(QRall + 4 * 64 * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, gq = lane >> 4;
    const int bh = blockIdx.x, b = bh / g.heads, h = bh % g.heads;
    const int C = g.heads * HD, N = g.N;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    float* QR = QRall + wave * 64 * 16;

    stage_rows_swz(base + C, ld, N, g.NT * 16, Ks, tid);
    stage_rows_t(base + 2 * C, ld, N, g.NP2, g.TPV, Vt, tid);
    for (int i = tid; i < g.NP2; i += 256) {
        const int n = i < N ? i : N - 1;
        kpos[i] = (uint32_t)(n / g.Wp) | ((uint32_t)(n % g.Wp) << 8);
    }
    uint4 th[2][2], tw[2][2];   // table fragments [row tile][k step]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            th[rt][ks] = table_frag(rel_h, 16 * rt + fr, g.RH, ks * 32 + gq * 8);
            tw[rt][ks] = table_frag(rel_w, 16 * rt + fr, g.RW, ks * 32 + gq * 8);
        }
    __syncthreads();

    const int iters = (g.NT + 3) / 4;
    for (int it = 0; it < iters; ++it) {
        const int qt = wave + 4 * it;
        const bool tile_ok = qt < g.NT;
        const int n = 16 * qt + fr;
        const bool nv = tile_ok && n < N;
        const int nc = nv ? n : 0;
        uint4 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = row_frag(base, ld, nc, nv, ks * 32 + gq * 8);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x4_t ah = {0.f, 0.f, 0.f, 0.f}, aw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ah = mma(th[rt][ks], qf[ks], ah);
                aw = mma(tw[rt][ks], qf[ks], aw);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                QR[(16 * rt + 4 * gq + rr) * 16 + fr] = ah[rr];
                QR[(32 + 16 * rt + 4 * gq + rr) * 16 + fr] = aw[rr];
            }
        }
        __syncthreads();
        f32x4_t s[16];
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
            s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (kt < g.NT) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) s[kt] = mma(ld16(Ks + swz(16 * kt + fr, ks * 4 + gq)), qf[ks], s[kt]);
            }
        }
        const uint32_t qp = kpos[nc];
        const int hq = (int)(qp & 0xffu) + g.Hp - 1, wq = (int)(qp >> 8) + g.Wp - 1;
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 16; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * gq + r, kc = key < g.NP2 ? key : 0;
                const uint32_t kp = kpos[kc];
                float v = scale * (s[kt][r] + QR[(hq - (int)(kp & 0xffu)) * 16 + fr] + QR[(32 + wq - (int)(kp >> 8)) * 16 + fr]);
                v = key < N ? v : -INFINITY;
                s[kt][r] = v;
                m = fmaxf(m, v);
            }
        m = xor16_max(m);
        m = xor32_max(m);
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 16; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[kt][r] - m);
                s[kt][r] = p;
                l += p;
            }
        l = xor16_sum(l);
        l = xor32_sum(l);
        f32x4_t oa[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oa[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk < g.KK) {
                const uint4 pf = pack_bf16x8(s[2 * kk][0], s[2 * kk][1], s[2 * kk][2], s[2 * kk][3], s[2 * kk + 1][0], s[2 * kk + 1][1], s[2 * kk + 1][2], s[2 * kk + 1][3]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const char* row = Vt + (16 * dt + fr) * g.TPV;
                    oa[dt] = mma(ld8x2(row + (32 * kk + 4 * gq) * 2, row + (32 * kk + 16 + 4 * gq) * 2), pf, oa[dt]);
                }
            }
        }
        if (nv) {
            const float inv = 1.0f / l;
            bf16_t* op = o + ((int64_t)b * N + n) * C + h * HD + 4 * gq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store4(op + 16 * dt, make_float4(oa[dt][0] * inv, oa[dt][1] * inv, oa[dt][2] * inv, oa[dt][3] * inv));
            if (gq == 0) lse[(int64_t)bh * N + n] = m + __logf(l);
        }
        __syncthreads();   // QR tile is rewritten by the next iteration
    }
}

// ===================================================================================================================
// forward beyond 256 tokens (448^2 pretraining inputs: 784): flash-style.  Workgroup = 64 queries of one (image, head)
// (wave = one 16-query tile), loop over blocks of 256 keys with an online softmax; per block the same S^T = K.Q^T /
// in-lane softmax / O^T = V^T.P^T scheme as above.  RT = 16-row tiles per table: 4 (Hp, Wp <= 32) or 8 (<= 64: the 1024^2
// detection fine-tunes, 4096 tokens).
// dynamic LDS: Ks[FKB*128] | Vs[FKB*128] | QR[4 waves][32 RT][16] f32 | kpos[FKB] u32
// ===================================================================================================================

template <int RT, int FKB>
__global__ __launch_bounds__(256) void full_fwd_flash_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                                 const float* __restrict__ rel_h, const float* __restrict__ rel_w,
                                                                 int N, int Hp, int Wp, int heads, float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* Ks = sm;
    char* Vs = Ks + FKB * 128;      // V rows like the K rows (round 6): the V^T operand of O^T = V^T.P^T comes out of ds_read_b64_tr_b16, not out of a transposed image written
    float* QRall = reinterpret_cast<float*>(Vs + FKB * 128);      // in 2-byte pieces (32 ds_write_b16 per thread and key block)
    uint32_t* kpos = reinterpret_cast<uint32_t*>(QRall + 4 * 32 * RT * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, gq = lane >> 4;
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int C = heads * HD, RH = 2 * Hp - 1, RW = 2 * Wp - 1;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    float* QR = QRall + wave * 32 * RT * 16;
    const int n = 16 * (blockIdx.y * 4 + wave) + fr;
    const bool nv = n < N;
    const int nc = nv ? n : N - 1;
    uint4 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = row_frag(base, ld, nc, nv, ks * 32 + gq * 8);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {   // q.Rh / q.Rw for every table row: one MFMA tile row each, exchanged through LDS
        f32x4_t ah = {0.f, 0.f, 0.f, 0.f}, aw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            ah = mma(table_frag(rel_h, 16 * rt + fr, RH, ks * 32 + gq * 8), qf[ks], ah);
            aw = mma(table_frag(rel_w, 16 * rt + fr, RW, ks * 32 + gq * 8), qf[ks], aw);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            QR[(16 * rt + 4 * gq + rr) * 16 + fr] = ah[rr];
            QR[(16 * RT + 16 * rt + 4 * gq + rr) * 16 + fr] = aw[rr];
        }
    }
    const int hq = nc / Wp + Hp - 1, wq = nc % Wp + Wp - 1;
    float m = -INFINITY, l = 0.f;
    f32x4_t oa[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oa[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nblk = (N + FKB - 1) / FKB;
    uint4 kpre[FKB * 8 / 256], vpre[FKB * 8 / 256];      // the next key block's rows, in flight while this one is worked on (round 6)
    prefetch_rows<FKB>(base + C, ld, N, tid, kpre);
    prefetch_rows<FKB>(base + 2 * C, ld, N, tid, vpre);
    for (int jb = 0; jb < nblk; ++jb) {
        const int kb0 = jb * FKB, rem = N - kb0;
        __syncthreads();   // the previous block's K / V reads are done (first pass: the QR tiles are visible)
        commit_rows<FKB>(Ks, tid, kpre);
        commit_rows<FKB>(Vs, tid, vpre);
        if (tid < FKB) {
            const int key = kb0 + tid < N ? kb0 + tid : N - 1;
            kpos[tid] = (uint32_t)(key / Wp) | ((uint32_t)(key % Wp) << 8);
        }
        if (jb + 1 < nblk) {
            prefetch_rows<FKB>(base + C + (int64_t)(kb0 + FKB) * ld, ld, rem - FKB, tid, kpre);
            prefetch_rows<FKB>(base + 2 * C + (int64_t)(kb0 + FKB) * ld, ld, rem - FKB, tid, vpre);
        }
        __syncthreads();
        const int keys = rem < FKB ? rem : FKB, tiles = (keys + 15) / 16, kkb = (keys + 31) / 32;
        f32x4_t s[FKB / 16];
#pragma unroll
        for (int kt = 0; kt < FKB / 16; ++kt) {
            s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (kt < tiles) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) s[kt] = mma(ld16(Ks + swz(16 * kt + fr, ks * 4 + gq)), qf[ks], s[kt]);
            }
        }
        float bm = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < FKB / 16; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kl = 16 * kt + 4 * gq + r;
                const uint32_t kp = kpos[kl];
                float v = scale * (s[kt][r] + QR[(hq - (int)(kp & 0xffu)) * 16 + fr] + QR[(16 * RT + wq - (int)(kp >> 8)) * 16 + fr]);
                v = kb0 + kl < N ? v : -INFINITY;
                s[kt][r] = v;
                bm = fmaxf(bm, v);
            }
        bm = xor16_max(bm);
        bm = xor32_max(bm);
        const float mnew = fmaxf(m, bm);
        const float alpha = __expf(m - mnew);   // first block: exp(-inf) = 0
        float lb = 0.f;
#pragma unroll
        for (int kt = 0; kt < FKB / 16; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[kt][r] - mnew);
                s[kt][r] = p;
                lb += p;
            }
        lb = xor16_sum(lb);
        lb = xor32_sum(lb);
        l = l * alpha + lb;
        m = mnew;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) oa[dt] = f32x4_t{oa[dt][0] * alpha, oa[dt][1] * alpha, oa[dt][2] * alpha, oa[dt][3] * alpha};
#pragma unroll
        for (int kk = 0; kk < FKB / 32; ++kk) {
            if (kk < kkb) {
                const uint4 pf = pack_bf16x8(s[2 * kk][0], s[2 * kk][1], s[2 * kk][2], s[2 * kk][3], s[2 * kk + 1][0], s[2 * kk + 1][1], s[2 * kk + 1][2], s[2 * kk + 1][3]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oa[dt] = mma(kt_frag_tr(Vs, 32 * kk + 4 * gq, dt, fr), pf, oa[dt]);
            }
        }
    }
    if (nv) {
        const float inv = 1.0f / l;
        bf16_t* op = o + ((int64_t)b * N + n) * C + h * HD + 4 * gq;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(op + 16 * dt, make_float4(oa[dt][0] * inv, oa[dt][1] * inv, oa[dt][2] * inv, oa[dt][3] * inv));
        if (gq == 0) lse[(int64_t)bh * N + n] = m + __logf(l);
    }
}

// ===================================================================================================================
// backward A: dQ and the rel-pos table gradients.
// dynamic LDS: Ks | Vs (NP2 x 128 each) | QR[NW][64][16] f32 | dQR[NW][64][16] f32 | Qtt[NW][64*40 B] | kpos[NP2] | E[2][KK][64] x16 B
// ===================================================================================================================
constexpr int QTP = 40;   // byte pitch of the per-wave transposed 16-query tile [d][16 q] (32 + 8)

// NW waves per workgroup: 8 (two per SIMD, the second hides the first one's LDS / exp latency) when the per-wave tiles fit next to
// the K / V images, else 4.  K^T fragments for dQ^T = K^T.dS^T come out of the row-major K image with the hardware transpose read
// (ds_read_b64_tr_b16; layout probed in tools/probes/tr_probe.hip) -- the separate K^T image of round 1 (29 KiB) is gone.
template <int NW>
__global__ __launch_bounds__(64 * NW) void full_bwd_a_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                             const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                             const float* __restrict__ rel_h, const float* __restrict__ rel_w, float* __restrict__ drel_part,
                                                             FGeom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* Ks = sm;
    constexpr int NTH = 64 * NW;
    char* Vs = Ks + g.NP2 * 128;
    float* QRall = reinterpret_cast<float*>(Vs + g.NP2 * 128);
    float* dQRall = QRall + NW * 64 * 16;
    char* Qttall = reinterpret_cast<char*>(dQRall + NW * 64 * 16);
    uint32_t* kpos = reinterpret_cast<uint32_t*>(Qttall + NW * 64 * QTP);
    char* Eimg = reinterpret_cast<char*>(kpos + g.NP2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, gq = lane >> 4;
    const int bh = blockIdx.x, b = bh / g.heads, h = bh % g.heads;
    const int C = g.heads * HD, N = g.N;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* dob = dout + (int64_t)b * N * C + h * HD;
    const bf16_t* ob = o + (int64_t)b * N * C + h * HD;
    float* QR = QRall + wave * 64 * 16;
    float* dQR = dQRall + wave * 64 * 16;
    char* Qtt = Qttall + wave * 64 * QTP;

    // 0/1 indicator operands E_h[a][key] = (row(key) == a), E_w[a][key] = (col(key) == a) in the MFMA A layout with the key
    // order of the dS^T fragments below: sum_k E[a][k] dS^T[k][q] is d(q.Rh)[q][hq - a] -- the segmented row / column sums
    // of dS come out of the matrix cores instead of LDS float atomics (measured ~190 LDS cycles per ds_add_f32 instruction).
    for (int idx = tid; idx < 2 * g.KK * 64; idx += NTH) {
        const int t = idx / (g.KK * 64), rem = idx % (g.KK * 64), kk = rem >> 6, l = rem & 63, a = l & 15, gl = l >> 4;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key = 32 * kk + 4 * gl + (j & 3) + (j >> 2) * 16;
            const bool hit = key < N && (t ? key % g.Wp : key / g.Wp) == a;
            w[j >> 1] |= hit ? (0x3f80u << ((j & 1) * 16)) : 0u;
        }
        *reinterpret_cast<uint4*>(Eimg + (size_t)idx * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }

    stage_rows_swz(base + C, ld, N, g.NP2, Ks, tid, NTH);
    stage_rows_swz(base + 2 * C, ld, N, g.NP2, Vs, tid, NTH);
    for (int i = tid; i < g.NP2; i += NTH) {
        const int n = i < N ? i : N - 1;
        kpos[i] = (uint32_t)(n / g.Wp) | ((uint32_t)(n % g.Wp) << 8);
    }
    uint4 th[2][2], tw[2][2], rhT[4], rwT[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            th[rt][ks] = table_frag(rel_h, 16 * rt + fr, g.RH, ks * 32 + gq * 8);
            tw[rt][ks] = table_frag(rel_w, 16 * rt + fr, g.RW, ks * 32 + gq * 8);
        }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        rhT[dt] = table_frag_t(rel_h, 16 * dt + fr, g.RH, 8 * gq);
        rwT[dt] = table_frag_t(rel_w, 16 * dt + fr, g.RW, 8 * gq);
    }
    f32x4_t tacc[2][2][4];   // [table][row tile][d tile]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) tacc[t][rt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    const int iters = (g.NT + NW - 1) / NW;
    for (int it = 0; it < iters; ++it) {
        const int qt = wave + NW * it;
        const bool tile_ok = qt < g.NT;
        const int n = 16 * qt + fr;
        const bool nv = tile_ok && n < N;
        const int nc = nv ? n : 0;
        uint4 qf[2], dof[2];
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[ks] = row_frag(base, ld, nc, nv, ks * 32 + gq * 8);
            dof[ks] = row_frag(dob, C, nc, nv, ks * 32 + gq * 8);
            const uint4 of = row_frag(ob, C, nc, nv, ks * 32 + gq * 8);
            const uint32_t a[4] = {dof[ks].x, dof[ks].y, dof[ks].z, dof[ks].w}, c[4] = {of.x, of.y, of.z, of.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                dl += bf16_bits_to_f32(a[e] & 0xffffu) * bf16_bits_to_f32(c[e] & 0xffffu) + bf16_bits_to_f32(a[e] >> 16) * bf16_bits_to_f32(c[e] >> 16);
        }
        dl = xor16_sum(dl);
        dl = xor32_sum(dl);
        const float ls = nv ? lse[(int64_t)bh * N + nc] : 0.f;
        // transposed copy of this query tile for the table-gradient MFMA: Qtt[d][fr]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint32_t w[4] = {qf[ks].x, qf[ks].y, qf[ks].z, qf[ks].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<uint16_t*>(Qtt + (ks * 32 + gq * 8 + 2 * e) * QTP + fr * 2) = (uint16_t)(w[e] & 0xffffu);
                *reinterpret_cast<uint16_t*>(Qtt + (ks * 32 + gq * 8 + 2 * e + 1) * QTP + fr * 2) = (uint16_t)(w[e] >> 16);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            f32x4_t ah = {0.f, 0.f, 0.f, 0.f}, aw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ah = mma(th[rt][ks], qf[ks], ah);
                aw = mma(tw[rt][ks], qf[ks], aw);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = 16 * rt + 4 * gq + rr;
                QR[r * 16 + fr] = ah[rr];
                QR[(32 + r) * 16 + fr] = aw[rr];
                dQR[r * 16 + fr] = 0.f;
                dQR[(32 + r) * 16 + fr] = 0.f;
            }
        }
        __syncthreads();
        const uint32_t qp = kpos[nc];
        const int hq = (int)(qp & 0xffu) + g.Hp - 1, wq = (int)(qp >> 8) + g.Wp - 1;
        f32x4_t dsT[16];
#pragma unroll
        for (int kt = 0; kt < 16; ++kt) {
            dsT[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (kt < g.NT) {
                f32x4_t sT = {0.f, 0.f, 0.f, 0.f}, dpT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    sT = mma(ld16(Ks + swz(16 * kt + fr, ks * 4 + gq)), qf[ks], sT);
                    dpT = mma(ld16(Vs + swz(16 * kt + fr, ks * 4 + gq)), dof[ks], dpT);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kt + 4 * gq + r;
                    const uint32_t kp = kpos[key];
                    const int ih = (hq - (int)(kp & 0xffu)) * 16 + fr, iw = (32 + wq - (int)(kp >> 8)) * 16 + fr;
                    const float v = scale * (sT[r] + QR[ih] + QR[iw]);
                    float ds = __expf(fminf(v - ls, 30.f)) * (dpT[r] - dl);
                    ds = (nv && key < N) ? ds : 0.f;
                    dsT[kt][r] = ds;
                }
            }
        }
        // d(qs)^T = K^T.dS^T + Rh^T.dQRh + Rw^T.dQRw ; dq = scale * d(qs)
        f32x4_t dq[4], dqh = {0.f, 0.f, 0.f, 0.f}, dqw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (kk < g.KK) {
                const uint4 dsf = pack_bf16x8(dsT[2 * kk][0], dsT[2 * kk][1], dsT[2 * kk][2], dsT[2 * kk][3],
                                              dsT[2 * kk + 1][0], dsT[2 * kk + 1][1], dsT[2 * kk + 1][2], dsT[2 * kk + 1][3]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) dq[dt] = mma(kt_frag_tr(Ks, 32 * kk + 4 * gq, dt, fr), dsf, dq[dt]);
                dqh = mma(ld16(Eimg + (kk * 64 + lane) * 16), dsf, dqh);             // lane: d(q.Rh) of query fr for key rows 4gq + r
                dqw = mma(ld16(Eimg + ((g.KK + kk) * 64 + lane) * 16), dsf, dqw);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int a = 4 * gq + r;
            if (a < g.Hp) dQR[(hq - a) * 16 + fr] = dqh[r];
            if (a < g.Wp) dQR[(32 + wq - a) * 16 + fr] = dqw[r];
        }
        __syncthreads();
        float e[8], f[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            e[x] = dQR[(8 * gq + x) * 16 + fr];
            f[x] = dQR[(32 + 8 * gq + x) * 16 + fr];
        }
        const uint4 eh = pack_bf16x8(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
        const uint4 ew = pack_bf16x8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dq[dt] = mma(rhT[dt], eh, dq[dt]);
            dq[dt] = mma(rwT[dt], ew, dq[dt]);
        }
        if (nv) {
            bf16_t* dp = dqkv + ((int64_t)b * N + n) * ld + h * HD + 4 * gq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) store4(dp + 16 * dt, make_float4(dq[dt][0] * scale, dq[dt][1] * scale, dq[dt][2] * scale, dq[dt][3] * scale));
        }
        // table gradients: tacc[t][rt][dt] += dQR_t[rows][16 queries] . Q[16 queries][d]   (k = 16 of the 32 slots used)
        {
            uint4 bq[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const char* row = Qtt + (16 * dt + fr) * QTP + (gq & 1) * 16;
                bq[dt] = gq < 2 ? ld8x2(row, row + 8) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    float v[8];
#pragma unroll
                    for (int x = 0; x < 8; ++x) v[x] = gq < 2 ? dQR[(32 * t + 16 * rt + fr) * 16 + (gq & 1) * 8 + x] : 0.f;
                    const uint4 af = pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) tacc[t][rt][dt] = mma(af, bq[dt], tacc[t][rt][dt]);
                }
        }
        __syncthreads();
    }
    // per-wave partial sums -> drel_part[bh][RH + RW rows][64] (zeroed by the launcher)
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
                    if (r < (t ? g.RW : g.RH)) atomicAdd(dp + ((t ? g.RH : 0) + r) * HD + 16 * dt + fr, tacc[t][rt][dt][rr] * scale);
                }
}

// ===================================================================================================================
// backward B: dK, dV.  8 waves per workgroup (two per SIMD: the second hides the first one's LDS / exp latency; the LDS images are
// shared, so the workgroup's footprint does not grow).   dynamic LDS: Qt[64*TPV] | dOt[64*TPV] | QRf[64][NP] f32 | lses[NP] | delta[NP] | kpos[NP2]
// ===================================================================================================================
__global__ __launch_bounds__(512) void full_bwd_b_mfma_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                             const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                             const float* __restrict__ rel_h, const float* __restrict__ rel_w, FGeom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* Qt = sm;
    char* dOt = Qt + 64 * g.TPV;
    float* QRf = reinterpret_cast<float*>(dOt + 64 * g.TPV);
    float* lses = QRf + 64 * g.NP;
    float* delta = lses + g.NP;
    uint32_t* kpos = reinterpret_cast<uint32_t*>(delta + g.NP);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, gq = lane >> 4;
    const int bh = blockIdx.x, b = bh / g.heads, h = bh % g.heads;
    const int C = g.heads * HD, N = g.N, NP = g.NP;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* dob = dout + (int64_t)b * N * C + h * HD;
    const bf16_t* ob = o + (int64_t)b * N * C + h * HD;

    stage_rows_t(base, ld, N, g.NP2, g.TPV, Qt, tid, 512);
    stage_rows_t(dob, C, N, g.NP2, g.TPV, dOt, tid, 512);
    for (int i = tid; i < g.NP2; i += 512) {
        const int n = i < N ? i : N - 1;
        kpos[i] = (uint32_t)(n / g.Wp) | ((uint32_t)(n % g.Wp) << 8);
    }
    for (int n = tid; n < NP; n += 512) {
        float dl = 0.f, ls = 0.f;
        const bool nv = n < N;
        const int nc = nv ? n : 0;
#pragma unroll
        for (int i = 0; i < HD / 8; ++i) {
            float a[8], c[8];
            load8(dob + (int64_t)nc * C + 8 * i, a);
            load8(ob + (int64_t)nc * C + 8 * i, c);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl += a[e] * c[e];
        }
        ls = lse[(int64_t)bh * N + nc];
        delta[n] = nv ? dl : 0.f;
        lses[n] = nv ? ls : 0.f;
    }
    {   // QRf[t*32 + r][n] = q_n . rel_t[r] for every query
        uint4 th[2][2], tw[2][2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                th[rt][ks] = table_frag(rel_h, 16 * rt + fr, g.RH, ks * 32 + gq * 8);
                tw[rt][ks] = table_frag(rel_w, 16 * rt + fr, g.RW, ks * 32 + gq * 8);
            }
        for (int qt = wave; qt < g.NT; qt += 8) {
            const int n = 16 * qt + fr;
            const bool nv = n < N;
            uint4 qf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[ks] = row_frag(base, ld, nv ? n : 0, nv, ks * 32 + gq * 8);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                f32x4_t ah = {0.f, 0.f, 0.f, 0.f}, aw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    ah = mma(th[rt][ks], qf[ks], ah);
                    aw = mma(tw[rt][ks], qf[ks], aw);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    QRf[(16 * rt + 4 * gq + rr) * NP + n] = ah[rr];
                    QRf[(32 + 16 * rt + 4 * gq + rr) * NP + n] = aw[rr];
                }
            }
        }
    }
    __syncthreads();

    for (int kt = wave; kt < g.NT; kt += 8) {
        const int key = 16 * kt + fr;
        const bool kv = key < N;
        const int kc = kv ? key : 0;
        uint4 kfb[2], vfb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kfb[ks] = row_frag(base + C, ld, kc, kv, ks * 32 + gq * 8);
            vfb[ks] = row_frag(base + 2 * C, ld, kc, kv, ks * 32 + gq * 8);
        }
        const uint32_t kp = kpos[kc];
        const int hk = (int)(kp & 0xffu) - (g.Hp - 1), wk = (int)(kp >> 8) - (g.Wp - 1);
        f32x4_t dks[4], dvs[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dks[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dvs[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int kk = 0; kk < g.KK; ++kk) {
            float pv[2][4], dv[2][4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * kk + half;
                const int nq = 16 * qt + fr;               // query row of the A fragments (lane fr)
                const bool qv = nq < N;
                f32x4_t sB = {0.f, 0.f, 0.f, 0.f}, dpB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    sB = mma(row_frag(base, ld, qv ? nq : 0, qv, ks * 32 + gq * 8), kfb[ks], sB);      // D[query 16qt+4gq+r][key fr]
                    dpB = mma(row_frag(dob, C, qv ? nq : 0, qv, ks * 32 + gq * 8), vfb[ks], dpB);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 16 * qt + 4 * gq + r, nn = n < NP ? n : 0;
                    const uint32_t qp = kpos[n < g.NP2 ? n : 0];
                    const float v = scale * (sB[r] + QRf[((int)(qp & 0xffu) - hk) * NP + nn] + QRf[(32 + (int)(qp >> 8) - wk) * NP + nn]);
                    float p = __expf(fminf(v - lses[nn], 30.f));
                    p = (n < N && kv) ? p : 0.f;
                    pv[half][r] = p;
                    dv[half][r] = p * (dpB[r] - delta[nn]);
                }
            }
            const uint4 pfb = pack_bf16x8(pv[0][0], pv[0][1], pv[0][2], pv[0][3], pv[1][0], pv[1][1], pv[1][2], pv[1][3]);
            const uint4 dsfb = pack_bf16x8(dv[0][0], dv[0][1], dv[0][2], dv[0][3], dv[1][0], dv[1][1], dv[1][2], dv[1][3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const char* rq = Qt + (16 * dt + fr) * g.TPV;
                const char* rd = dOt + (16 * dt + fr) * g.TPV;
                dks[dt] = mma(ld8x2(rq + (32 * kk + 4 * gq) * 2, rq + (32 * kk + 16 + 4 * gq) * 2), dsfb, dks[dt]);   // D[d][key fr]
                dvs[dt] = mma(ld8x2(rd + (32 * kk + 4 * gq) * 2, rd + (32 * kk + 16 + 4 * gq) * 2), pfb, dvs[dt]);
            }
        }
        if (kv) {
            bf16_t* dk = dqkv + ((int64_t)b * N + key) * ld + C + h * HD + 4 * gq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                store4(dk + 16 * dt, make_float4(dks[dt][0] * scale, dks[dt][1] * scale, dks[dt][2] * scale, dks[dt][3] * scale));
                store4(dk + C + 16 * dt, make_float4(dvs[dt][0], dvs[dt][1], dvs[dt][2], dvs[dt][3]));
            }
        }
    }
}

bool make_fgeom(int64_t Hp, int64_t Wp, int64_t heads, FGeom& g) {
    g.N = (int)(Hp * Wp); g.Hp = (int)Hp; g.Wp = (int)Wp; g.heads = (int)heads;
    g.NT = (g.N + 15) / 16;
    g.NP = g.NT * 16;
    g.NP2 = (g.N + 31) / 32 * 32;
    g.KK = g.NP2 / 32;
    g.TPV = g.NP2 * 2 + 8;
    g.RH = 2 * g.Hp - 1;
    g.RW = 2 * g.Wp - 1;
    return g.N <= 256 && g.RH <= 32 && g.RW <= 32;
}

}  // namespace

int mtp_full_fwd_mfma_launch(const void* qkv, void* o, float* lse, const float* rel_h, const float* rel_w,
                             int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s) {
    if (mtp_full_v3_fits(Hp, Wp)) return mtp_full_v3_fwd_launch(qkv, o, lse, rel_h, rel_w, B, Hp, Wp, heads, scale, s);
    FGeom g;
    if (!make_fgeom(Hp, Wp, heads, g)) {
        const int64_t N = Hp * Wp;
        if (N <= 256 || Hp > 64 || Wp > 64) return MTP_ERR_UNSUPPORTED;
        const bool big = Hp > 32 || Wp > 32;          // tables of up to 127 rows: 8 row tiles each
        const dim3 grid((unsigned)(B * heads), (unsigned)((N + 63) / 64));
        if (big) {
            constexpr int KBLK = 256;
            const size_t lds = 2 * (size_t)KBLK * 128 + (size_t)4 * 32 * 8 * 16 * 4 + KBLK * 4;
            (void)hipFuncSetAttribute((const void*)full_fwd_flash_mfma_kernel<8, KBLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((full_fwd_flash_mfma_kernel<8, KBLK>), grid, dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)o, lse, rel_h, rel_w, (int)N, (int)Hp, (int)Wp, (int)heads, scale);
        } else {
            // 128-key blocks: 65 KiB of LDS, two workgroups (8 waves) per CU instead of one
            constexpr int KBLK = 128;
            const size_t lds = 2 * (size_t)KBLK * 128 + (size_t)4 * 32 * 4 * 16 * 4 + KBLK * 4;
            (void)hipFuncSetAttribute((const void*)full_fwd_flash_mfma_kernel<4, KBLK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((full_fwd_flash_mfma_kernel<4, KBLK>), grid, dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)o, lse, rel_h, rel_w, (int)N, (int)Hp, (int)Wp, (int)heads, scale);
        }
        return mtp_launch_status();
    }
    const size_t lds = (size_t)g.NT * 16 * 128 + (size_t)64 * g.TPV + 4 * 64 * 16 * 4 + (size_t)g.NP2 * 4;
    (void)hipFuncSetAttribute((const void*)full_fwd_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(full_fwd_mfma_kernel, dim3((unsigned)(B * heads)), dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)o, lse, rel_h, rel_w, g, scale);
    return mtp_launch_status();
}

int mtp_full_bwd_mfma_launch(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, const float* rel_h, const float* rel_w,
                             float* drel_part, int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s) {
    if (mtp_full_v3_fits(Hp, Wp)) return mtp_full_v3_bwd_launch(qkv, o, dout, lse, dqkv, rel_h, rel_w, drel_part, B, Hp, Wp, heads, scale, s);
    FGeom g;
    if (!make_fgeom(Hp, Wp, heads, g)) return MTP_ERR_UNSUPPORTED;
    hipError_t e = hipMemsetAsync(drel_part, 0, sizeof(float) * (size_t)(B * heads) * (size_t)(g.RH + g.RW) * HD, s);
    if (e != hipSuccess) return (int)e;
    const auto lds_a = [&](int nw) { return 2 * (size_t)g.NP2 * 128 + 2 * (size_t)nw * 64 * 16 * 4 + (size_t)nw * 64 * QTP + (size_t)g.NP2 * 4 + 2 * (size_t)g.KK * 64 * 16; };
    const size_t lds_b = 2 * (size_t)64 * g.TPV + (size_t)64 * g.NP * 4 + 2 * (size_t)g.NP * 4 + (size_t)g.NP2 * 4;
    if (lds_a(4) > 160 * 1024 || lds_b > 160 * 1024) return MTP_ERR_UNSUPPORTED;
    (void)hipFuncSetAttribute((const void*)full_bwd_a_mfma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_a(8) <= 160 * 1024 ? lds_a(8) : lds_a(4)));
    (void)hipFuncSetAttribute((const void*)full_bwd_a_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a(4));
    (void)hipFuncSetAttribute((const void*)full_bwd_b_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
    if (lds_a(8) <= 160 * 1024)
        hipLaunchKernelGGL(full_bwd_a_mfma_kernel<8>, dim3((unsigned)(B * heads)), dim3(512), lds_a(8), s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse,
                           (bf16_t*)dqkv, rel_h, rel_w, drel_part, g, scale);
    else
        hipLaunchKernelGGL(full_bwd_a_mfma_kernel<4>, dim3((unsigned)(B * heads)), dim3(256), lds_a(4), s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse,
                           (bf16_t*)dqkv, rel_h, rel_w, drel_part, g, scale);
    hipLaunchKernelGGL(full_bwd_b_mfma_kernel, dim3((unsigned)(B * heads)), dim3(512), lds_b, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse,
                       (bf16_t*)dqkv, rel_h, rel_w, g, scale);
    return mtp_launch_status();
}
