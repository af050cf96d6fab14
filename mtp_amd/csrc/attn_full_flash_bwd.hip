// Flash-style bf16 MFMA backward of the full (global) attention blocks beyond 256 tokens (Attention.forward, VIT:90-111, with the
// decomposed relative-position terms of calc_rel_pos_spatial, VIT:142-193): 448^2 pretraining inputs (28 x 28 = 784 tokens),
// 512^2 (1024) and the 1024^2 detection fine-tunes (64 x 64 = 4096).  gfx950, head_dim 64, Hp, Wp <= 64, Wp >= 10.
//
// The relative-position logit of a (query, key) pair is  q.Rh[hq - hk + Hp - 1] + q.Rw[wq - wk + Wp - 1]: for one query it takes
// only Hp values along the key rows and Wp along the key columns.  Those Hp + Wp numbers per query ("bias rows") are what the
// three kernels exchange, in the caller's f32 workspace, instead of per-(query, key) matrices:
//   prep : bias[n][0..Hp) = q_n.Rh[hq - a + Hp - 1],  bias[n][Hp..Hp+Wp) = q_n.Rw[wq - a + Wp - 1]  (MFMA against the tables),
//          delta[n] = dO_n . O_n
//   dq   : workgroup = 64 queries, loop over blocks of 128 keys: S^T = K.Q^T, dP^T = V.dO^T, dS = P (dP - delta),
//          dQ^T += K^T.dS^T; the gradient of the bias rows is a segmented sum of dS over the keys of one grid row / column,
//          done on the matrix cores against 0/1 indicator operands (as in attn_full_mfma.hip); at the end
//          dQ^T += Rh^T.dQRh + Rw^T.dQRw and the table gradients d(rel_pos_h/w) += dQR . Q (atomics into the per-(image, head) partials)
//   dkv  : workgroup = 64 keys, loop over blocks of 64 queries: dV^T += dO^T.P, dK^T += Q^T.dS; the bias comes from the
//          workspace rows (only the <= 8 grid rows the workgroup's keys touch, and all Wp columns)
#include "attn_mfma.h"
#include "attn_full_common.h"

namespace {

constexpr int QTP = 40;             // byte pitch of the per-wave transposed 16-query tile [d][16 q]

struct FlashGeom {
    int N, Hp, Wp, heads, HW, HWP, RH, RW, WT;
};

// ===================================================================================================================
// prep: bias rows and delta.  grid (B*heads, ceil(N / 64)); wave = one 16-query tile
// ===================================================================================================================
__global__ __launch_bounds__(256) void flash_prep_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                        const float* __restrict__ rel_h, const float* __restrict__ rel_w,
                                                        float* __restrict__ bias, float* __restrict__ delta, FlashGeom g) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, gq = lane >> 4;
    const int bh = blockIdx.x, b = bh / g.heads, h = bh % g.heads;
    const int C = g.heads * HD, N = g.N;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* dob = dout + (int64_t)b * N * C + h * HD;
    const bf16_t* ob = o + (int64_t)b * N * C + h * HD;
    const int n = 64 * blockIdx.y + 16 * wave + fr;
    const bool nv = n < N;
    const int nc = nv ? n : N - 1;
    uint4 qf[2];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = row_frag(base, ld, nc, nv, ks * 32 + gq * 8);
        const uint4 df = row_frag(dob, C, nc, nv, ks * 32 + gq * 8), of = row_frag(ob, C, nc, nv, ks * 32 + gq * 8);
        const uint32_t a[4] = {df.x, df.y, df.z, df.w}, c[4] = {of.x, of.y, of.z, of.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            dl += bf16_bits_to_f32(a[e] & 0xffffu) * bf16_bits_to_f32(c[e] & 0xffffu) + bf16_bits_to_f32(a[e] >> 16) * bf16_bits_to_f32(c[e] >> 16);
    }
    dl = xor16_sum(dl);
    dl = xor32_sum(dl);
    if (nv && gq == 0) delta[(int64_t)bh * N + n] = dl;
    const int hq = nc / g.Wp + g.Hp - 1, wq = nc % g.Wp + g.Wp - 1;
    float* row = bias + ((int64_t)bh * N + nc) * g.HW;
#pragma unroll 1
    for (int rt = 0; rt < (g.RH + 15) / 16; ++rt) {   // D[table row 16rt + 4gq + rr][query fr]
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc = mma(table_frag(rel_h, 16 * rt + fr, g.RH, ks * 32 + gq * 8), qf[ks], acc);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int a = hq - (16 * rt + 4 * gq + rr);
            if (nv && a >= 0 && a < g.Hp) row[a] = acc[rr];
        }
    }
#pragma unroll 1
    for (int rt = 0; rt < (g.RW + 15) / 16; ++rt) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc = mma(table_frag(rel_w, 16 * rt + fr, g.RW, ks * 32 + gq * 8), qf[ks], acc);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int a = wq - (16 * rt + 4 * gq + rr);
            if (nv && a >= 0 && a < g.Wp) row[g.Hp + a] = acc[rr];
        }
    }
}

// ===================================================================================================================
// dQ and the table gradients.  grid (B*heads, ceil(N / 64)); wave = one 16-query tile.
// dynamic LDS: Ks | Vs (KB x 128 B) | E[(1 + WT)][KB / 32][64] x 16 B | bias[64][HWP] f32 | dbias[64][HWP] f32 |
//              Qtt[4 waves][64][QTP] | kpos[KB] u32 | qpos[64] u32
// ===================================================================================================================
template <int KB>      // keys per block: 128, or 64 when that lets two workgroups share a CU
__global__ __launch_bounds__(256) void flash_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dqkv, const float* __restrict__ rel_h, const float* __restrict__ rel_w,
                                                          const float* __restrict__ bias_g, const float* __restrict__ delta_g, float* __restrict__ drel_part,
                                                          FlashGeom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* Ks = sm;
    char* Vs = Ks + KB * 128;
    char* Eimg = Vs + KB * 128;      // (no K^T image since round 6: the K^T fragments of dQ^T += K^T.dS^T come out of the K rows with ds_read_b64_tr_b16 -- the image cost a
                                     //  second global read of the key block and 16 two-byte LDS stores per thread and block)
    float* bias = reinterpret_cast<float*>(Eimg + (1 + g.WT) * (KB / 32) * 64 * 16);
    float* dbias = bias + 64 * g.HWP;
    char* Qttall = reinterpret_cast<char*>(dbias + 64 * g.HWP);
    uint32_t* kpos = reinterpret_cast<uint32_t*>(Qttall + 4 * 64 * QTP);
    uint32_t* qpos = kpos + KB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, gq = lane >> 4;
    const int bh = blockIdx.x, b = bh / g.heads, h = bh % g.heads;
    const int C = g.heads * HD, N = g.N, Hp = g.Hp, Wp = g.Wp, HWP = g.HWP;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* dob = dout + (int64_t)b * N * C + h * HD;
    char* Qtt = Qttall + wave * 64 * QTP;
    const int q0 = 64 * blockIdx.y;
    const int n = q0 + 16 * wave + fr;
    const bool nv = n < N;
    const int nc = nv ? n : N - 1;
    const int qrow = (16 * wave + fr) * HWP;

    uint4 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = row_frag(base, ld, nc, nv, ks * 32 + gq * 8);
        dof[ks] = row_frag(dob, C, nc, nv, ks * 32 + gq * 8);
        const uint32_t w[4] = {qf[ks].x, qf[ks].y, qf[ks].z, qf[ks].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // transposed copy of the query tile for the table-gradient MFMA: Qtt[d][fr]
            *reinterpret_cast<uint16_t*>(Qtt + (ks * 32 + gq * 8 + 2 * e) * QTP + fr * 2) = (uint16_t)(w[e] & 0xffffu);
            *reinterpret_cast<uint16_t*>(Qtt + (ks * 32 + gq * 8 + 2 * e + 1) * QTP + fr * 2) = (uint16_t)(w[e] >> 16);
        }
    }
    const float ls = nv ? lse[(int64_t)bh * N + nc] : 0.f;
    const float dl = nv ? delta_g[(int64_t)bh * N + nc] : 0.f;
    for (int idx = tid; idx < 64 * g.HW; idx += 256) {
        const int q = idx / g.HW, a = idx - q * g.HW;
        bias[q * HWP + a] = q0 + q < N ? bias_g[((int64_t)bh * N + q0 + q) * g.HW + a] : 0.f;
        dbias[q * HWP + a] = 0.f;
    }
    if (tid < 64) {
        const int nn = q0 + tid < N ? q0 + tid : N - 1;
        qpos[tid] = (uint32_t)(nn / Wp) | ((uint32_t)(nn % Wp) << 8);
    }
    f32x4_t dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nblk = (N + KB - 1) / KB;
    uint4 kpre[KB * 8 / 256], vpre[KB * 8 / 256];      // the next key block's rows, in flight while this one is worked on (round 6)
    prefetch_rows<KB>(base + C, ld, N, tid, kpre);
    prefetch_rows<KB>(base + 2 * C, ld, N, tid, vpre);
    for (int jb = 0; jb < nblk; ++jb) {
        const int kb0 = jb * KB, rem = N - kb0, kh0 = kb0 / Wp;
        __syncthreads();   // the previous block's K / V / E reads are done
        commit_rows<KB>(Ks, tid, kpre);
        commit_rows<KB>(Vs, tid, vpre);
        if (jb + 1 < nblk) {
            prefetch_rows<KB>(base + C + (int64_t)(kb0 + KB) * ld, ld, rem - KB, tid, kpre);
            prefetch_rows<KB>(base + 2 * C + (int64_t)(kb0 + KB) * ld, ld, rem - KB, tid, vpre);
        }
        if (tid < KB) {   // (grid row relative to the block's first row) | column << 8 | valid << 16
            const bool ok = kb0 + tid < N;
            const int key = ok ? kb0 + tid : N - 1;
            kpos[tid] = (uint32_t)(key / Wp - kh0) | ((uint32_t)(key % Wp) << 8) | (ok ? 0x10000u : 0u);
        }
        __syncthreads();
        // 0/1 indicator operands in the MFMA A layout, key order of the dS^T fragments: E[0][a][key] = (row(key) - kh0 == a),
        // E[1 + wt][a][key] = (col(key) == 16 wt + a).  sum_k E[a][k] dS^T[k][q] = the bias-row gradient of query q.
        for (int idx = tid; idx < (1 + g.WT) * (KB / 32) * 64; idx += 256) {
            const int t = idx / ((KB / 32) * 64), kk = (idx >> 6) % (KB / 32), l = idx & 63, a = l & 15, gl = l >> 4;
            const uint4 k0 = *reinterpret_cast<const uint4*>(kpos + 32 * kk + 4 * gl), k1 = *reinterpret_cast<const uint4*>(kpos + 32 * kk + 16 + 4 * gl);
            const uint32_t kp[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
            const uint32_t want = t == 0 ? (uint32_t)a : (uint32_t)(16 * (t - 1) + a);
            uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t pos = t == 0 ? (kp[j] & 0xffu) : ((kp[j] >> 8) & 0xffu);
                const bool hit = (kp[j] & 0x10000u) && pos == want;
                w[j >> 1] |= hit ? (0x3f80u << ((j & 1) * 16)) : 0u;
            }
            *reinterpret_cast<uint4*>(Eimg + (size_t)idx * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        }
        __syncthreads();

        f32x4_t dsT[KB / 16];
#pragma unroll
        for (int kt = 0; kt < KB / 16; ++kt) {
            f32x4_t sT = {0.f, 0.f, 0.f, 0.f}, dpT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                sT = mma(ld16(Ks + swz(16 * kt + fr, ks * 4 + gq)), qf[ks], sT);        // D[key 16kt + 4gq + r][query fr]
                dpT = mma(ld16(Vs + swz(16 * kt + fr, ks * 4 + gq)), dof[ks], dpT);
            }
            const uint4 kq = *reinterpret_cast<const uint4*>(kpos + 16 * kt + 4 * gq);
            const uint32_t kp[4] = {kq.x, kq.y, kq.z, kq.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = scale * (sT[r] + bias[qrow + kh0 + (int)(kp[r] & 0xffu)] + bias[qrow + Hp + (int)((kp[r] >> 8) & 0xffu)]);
                const float ds = __expf(fminf(v - ls, 30.f)) * (dpT[r] - dl);
                dsT[kt][r] = (nv && (kp[r] & 0x10000u)) ? ds : 0.f;
            }
        }
        f32x4_t dqh = {0.f, 0.f, 0.f, 0.f}, dqw[4];
#pragma unroll
        for (int wt = 0; wt < 4; ++wt) dqw[wt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KB / 32; ++kk) {
            const uint4 dsf = pack_bf16x8(dsT[2 * kk][0], dsT[2 * kk][1], dsT[2 * kk][2], dsT[2 * kk][3],
                                          dsT[2 * kk + 1][0], dsT[2 * kk + 1][1], dsT[2 * kk + 1][2], dsT[2 * kk + 1][3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = mma(kt_frag_tr(Ks, 32 * kk + 4 * gq, dt, fr), dsf, dq[dt]);
            dqh = mma(ld16(Eimg + (kk * 64 + lane) * 16), dsf, dqh);      // D[relative key row 4gq + r][query fr]
#pragma unroll
            for (int wt = 0; wt < 4; ++wt)
                if (wt < g.WT) dqw[wt] = mma(ld16(Eimg + (((1 + wt) * (KB / 32) + kk) * 64 + lane) * 16), dsf, dqw[wt]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // this wave's rows of dbias: nobody else touches them
            const int a = kh0 + 4 * gq + r;
            if (a < Hp) dbias[qrow + a] += dqh[r];
#pragma unroll
            for (int wt = 0; wt < 4; ++wt) {
                const int c = 16 * wt + 4 * gq + r;
                if (wt < g.WT && c < Wp) dbias[qrow + Hp + c] += dqw[wt][r];
            }
        }
    }

    // ---- dQ^T += Rh^T . dQRh + Rw^T . dQRw with dQRh[r][q] = dbias[q][hq + Hp - 1 - r]
    const uint32_t qp = qpos[16 * wave + fr];
    const int hq = (int)(qp & 0xffu) + Hp - 1, wq = (int)(qp >> 8) + Wp - 1;
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
        const float* tab = t ? rel_w : rel_h;
        const int R = t ? g.RW : g.RH, lim = t ? Wp : Hp, off = t ? Hp : 0, pq = t ? wq : hq;
#pragma unroll 1
        for (int c = 0; c < (R + 31) / 32; ++c) {
            float e[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int a = pq - (32 * c + 8 * gq + x);
                e[x] = (a >= 0 && a < lim) ? dbias[qrow + off + a] : 0.f;
            }
            const uint4 ef = pack_bf16x8(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = mma(table_frag_t(tab, 16 * dt + fr, R, 32 * c + 8 * gq), ef, dq[dt]);
        }
    }
    if (nv) {
        bf16_t* dp = dqkv + ((int64_t)b * N + n) * ld + h * HD + 4 * gq;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(dp + 16 * dt, make_float4(dq[dt][0] * scale, dq[dt][1] * scale, dq[dt][2] * scale, dq[dt][3] * scale));
    }
    // ---- table gradients: d(rel)[r][d] += scale * sum_q dQR[r][q] Q[q][d]   (k = 16 of the 32 slots used)
    {
        uint4 bq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const char* row = Qtt + (16 * dt + fr) * QTP + (gq & 1) * 16;
            bq[dt] = gq < 2 ? ld8x2(row, row + 8) : make_uint4(0, 0, 0, 0);
        }
        uint32_t qps[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) qps[x] = qpos[16 * wave + 8 * (gq & 1) + x];
        float* dp = drel_part + (int64_t)bh * (g.RH + g.RW) * HD;
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            const int R = t ? g.RW : g.RH, lim = t ? Wp : Hp, off = t ? Hp : 0;
#pragma unroll 1
            for (int rt = 0; rt < (R + 15) / 16; ++rt) {
                float v[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const int pq = t ? (int)(qps[x] >> 8) + Wp - 1 : (int)(qps[x] & 0xffu) + Hp - 1;
                    const int a = pq - (16 * rt + fr);
                    v[x] = (gq < 2 && a >= 0 && a < lim) ? dbias[(16 * wave + 8 * (gq & 1) + x) * HWP + off + a] : 0.f;
                }
                const uint4 af = pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const f32x4_t acc = mma(af, bq[dt], f32x4_t{0.f, 0.f, 0.f, 0.f});     // D[table row 16rt + 4gq + rr][d = 16dt + fr]
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int r = 16 * rt + 4 * gq + rr;
                        if (r < R && acc[rr] != 0.f) atomicAdd(dp + ((t ? g.RH : 0) + r) * HD + 16 * dt + fr, acc[rr] * scale);
                    }
                }
            }
        }
    }
}

// ===================================================================================================================
// dK, dV.  grid (B*heads, ceil(N / 64)); wave = one 16-key tile; loop over blocks of 64 queries.
// dynamic LDS: Qs | dOs (64 x 128 B, swizzled rows) | bw[64][WPP] f32 | bhs[64][9] f32 | lses[64] | dels[64]
// ===================================================================================================================
__global__ __launch_bounds__(256) void flash_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dqkv, const float* __restrict__ bias_g, const float* __restrict__ delta_g,
                                                           FlashGeom g, float scale) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* Qs = sm;
    char* dOs = Qs + 64 * 128;
    const int WPP = g.Wp | 1;
    float* bw = reinterpret_cast<float*>(dOs + 64 * 128);      // (no Q^T / dO^T images since round 6: transposed reads of the row images)
    float* bhs = bw + 64 * WPP;
    float* lses = bhs + 64 * 9;
    float* dels = lses + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, gq = lane >> 4;
    const int bh = blockIdx.x, b = bh / g.heads, h = bh % g.heads;
    const int C = g.heads * HD, N = g.N, Hp = g.Hp, Wp = g.Wp;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* dob = dout + (int64_t)b * N * C + h * HD;
    const int k0 = 64 * blockIdx.y, khlo = k0 / Wp;
    const int key = k0 + 16 * wave + fr;
    const bool kv = key < N;
    const int kc = kv ? key : N - 1;
    const int ka = kc / Wp - khlo, kw = kc % Wp;     // ka <= 63 / Wp + 1 <= 7 (Wp >= 10)
    uint4 kfb[2], vfb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kfb[ks] = row_frag(base + C, ld, kc, kv, ks * 32 + gq * 8);
        vfb[ks] = row_frag(base + 2 * C, ld, kc, kv, ks * 32 + gq * 8);
    }
    f32x4_t dks[4], dvs[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        dks[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dvs[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    const int nblk = (N + 63) / 64;
    uint4 qpre[2], dpre[2];      // the next query block's Q / dO rows, in flight while this one is worked on (round 6)
    prefetch_rows<64>(base, ld, N, tid, qpre);
    prefetch_rows<64>(dob, C, N, tid, dpre);
    for (int qb = 0; qb < nblk; ++qb) {
        const int qb0 = 64 * qb, rem = N - qb0;
        __syncthreads();
        commit_rows<64>(Qs, tid, qpre);
        commit_rows<64>(dOs, tid, dpre);
        if (qb + 1 < nblk) {
            prefetch_rows<64>(base + (int64_t)(qb0 + 64) * ld, ld, rem - 64, tid, qpre);
            prefetch_rows<64>(dob + (int64_t)(qb0 + 64) * C, C, rem - 64, tid, dpre);
        }
        for (int idx = tid; idx < 64 * Wp; idx += 256) {
            const int q = idx / Wp, c = idx - q * Wp;
            bw[q * WPP + c] = q < rem ? bias_g[((int64_t)bh * N + qb0 + q) * g.HW + Hp + c] : 0.f;
        }
        for (int idx = tid; idx < 64 * 8; idx += 256) {
            const int q = idx >> 3, a = idx & 7;
            bhs[q * 9 + a] = (q < rem && khlo + a < Hp) ? bias_g[((int64_t)bh * N + qb0 + q) * g.HW + khlo + a] : 0.f;
        }
        if (tid < 64) {
            lses[tid] = tid < rem ? lse[(int64_t)bh * N + qb0 + tid] : 0.f;
            dels[tid] = tid < rem ? delta_g[(int64_t)bh * N + qb0 + tid] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float pv[2][4], dv[2][4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * kk + half;
                f32x4_t sB = {0.f, 0.f, 0.f, 0.f}, dpB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    sB = mma(ld16(Qs + swz(16 * qt + fr, ks * 4 + gq)), kfb[ks], sB);        // D[query 16qt + 4gq + r][key fr]
                    dpB = mma(ld16(dOs + swz(16 * qt + fr, ks * 4 + gq)), vfb[ks], dpB);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 16 * qt + 4 * gq + r;
                    const float v = scale * (sB[r] + bhs[q * 9 + ka] + bw[q * WPP + kw]);
                    float p = __expf(fminf(v - lses[q], 30.f));
                    p = (q < rem && kv) ? p : 0.f;
                    pv[half][r] = p;
                    dv[half][r] = p * (dpB[r] - dels[q]);
                }
            }
            const uint4 pfb = pack_bf16x8(pv[0][0], pv[0][1], pv[0][2], pv[0][3], pv[1][0], pv[1][1], pv[1][2], pv[1][3]);
            const uint4 dsfb = pack_bf16x8(dv[0][0], dv[0][1], dv[0][2], dv[0][3], dv[1][0], dv[1][1], dv[1][2], dv[1][3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dks[dt] = mma(kt_frag_tr(Qs, 32 * kk + 4 * gq, dt, fr), dsfb, dks[dt]);   // D[d 16dt + 4gq + r][key fr]
                dvs[dt] = mma(kt_frag_tr(dOs, 32 * kk + 4 * gq, dt, fr), pfb, dvs[dt]);
            }
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

}  // namespace

// workspace (f32, mtp_full_attn_bwd_workspace_floats): bias rows (B*heads, N, Hp + Wp) | delta (B*heads, N)
int mtp_full_bwd_flash_launch(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, const float* rel_h, const float* rel_w,
                              float* drel_part, float* workspace, int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s) {
    const int64_t N = Hp * Wp;
    // (<= 256 tokens: measured at 14 x 14, B = 64: 579 us vs 590 us for the one-workgroup-per-(image, head) kernels -- 196 = 3 x 64 + 4
    //  wastes a quarter of the query workgroups; not worth a second code path)
    if (N <= 256 || Hp > 64 || Wp > 64 || Wp < 10) return MTP_ERR_UNSUPPORTED;
    if (!workspace) return MTP_ERR_ARG;
    FlashGeom g;
    g.N = (int)N; g.Hp = (int)Hp; g.Wp = (int)Wp; g.heads = (int)heads;
    g.HW = g.Hp + g.Wp; g.HWP = g.HW | 1;
    g.RH = 2 * g.Hp - 1; g.RW = 2 * g.Wp - 1;
    g.WT = (g.Wp + 15) / 16;
    float* bias = workspace;
    float* delta = workspace + B * heads * N * g.HW;
    hipError_t e = hipMemsetAsync(drel_part, 0, sizeof(float) * (size_t)(B * heads) * (size_t)(g.RH + g.RW) * HD, s);
    if (e != hipSuccess) return (int)e;
    const auto lds_q = [&](int kb) { return 2 * (size_t)kb * 128 + (size_t)(1 + g.WT) * (kb / 32) * 64 * 16 + 2 * (size_t)64 * g.HWP * 4 + 4 * 64 * QTP + kb * 4 + 64 * 4; };
    const size_t lds_k = 2 * (size_t)64 * 128 + (size_t)64 * (g.Wp | 1) * 4 + 64 * 9 * 4 + 2 * 64 * 4;
    // key block of the dq kernel: 64 keys when that brings its LDS under half a CU's (two workgroups = 8 waves per CU; 28 x 28: 70 KiB)
    // and the indicator rows still fit one MFMA tile (64 / Wp + 2 <= 16 always holds for Wp >= 10)
    const bool small = lds_q(64) <= 80 * 1024;
    if (!small && lds_q(128) > 160 * 1024) return MTP_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(B * heads), (unsigned)((N + 63) / 64)), block(256);
    (void)hipFuncSetAttribute((const void*)flash_bwd_dq_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q(64));
    (void)hipFuncSetAttribute((const void*)flash_bwd_dq_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q(128));
    (void)hipFuncSetAttribute((const void*)flash_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);
    hipLaunchKernelGGL(flash_prep_kernel, grid, block, 0, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, rel_h, rel_w, bias, delta, g);
    if (small)
        hipLaunchKernelGGL(flash_bwd_dq_kernel<64>, grid, block, lds_q(64), s, (const bf16_t*)qkv, (const bf16_t*)dout, lse, (bf16_t*)dqkv, rel_h, rel_w,
                           (const float*)bias, (const float*)delta, drel_part, g, scale);
    else
        hipLaunchKernelGGL(flash_bwd_dq_kernel<128>, grid, block, lds_q(128), s, (const bf16_t*)qkv, (const bf16_t*)dout, lse, (bf16_t*)dqkv, rel_h, rel_w,
                           (const float*)bias, (const float*)delta, drel_part, g, scale);
    hipLaunchKernelGGL(flash_bwd_dkv_kernel, grid, block, lds_k, s, (const bf16_t*)qkv, (const bf16_t*)dout, lse, (bf16_t*)dqkv,
                       (const float*)bias, (const float*)delta, g, scale);
    return mtp_launch_status();
}
