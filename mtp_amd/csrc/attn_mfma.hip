// bf16 MFMA attention kernels for gfx950 (v_mfma_f32_16x16x32_bf16), head_dim = 64.
//
// RVSA window attention: one 64-lane wavefront per (image, window, head); the 49x49 problem is padded to 64x64.
//   S^T = Ksel . Q^T  -> lane holds (query = lane&15 of a 16-query tile; 4 consecutive keys) so a query's softmax needs only
//   in-lane reductions + two cross-lane shuffles (xor 16, 32), and the probabilities are directly the B-operand of the next
//   MFMA (O^T = V^T . P^T) -- with the k index permuted identically on the A side (two 8-byte LDS reads of the transposed
//   V image), so P never goes through LDS.  The backward computes S and dP in BOTH orientations by swapping the MFMA
//   operands (same registers), which replaces every LDS transpose of P / dS:
//     orientation A (lane: query, 4 keys)  -> dQ^T = K^T . dS^T          (contraction over keys)
//     orientation B (lane: key, 4 queries) -> dK^T = Q^T . dS,  dV^T = dO^T . P   (contraction over queries)
//   dK_sel/dV_sel are scattered through the bilinear footprint with f32 atomics; the sampling-coordinate gradients are
//   reduced in-wave to the 5 scalars of the (window, head).
#include "attn_mfma.h"
#include "common.h"

namespace {

constexpr int HD = 64;
constexpr int TP = 136;   // byte pitch of the transposed [d][key|query] bf16 images (128 + 8: conflict-free 8-byte reads)

struct RvsaGeom {
    int Hp, Wp, He, We, pad_t, pad_l, nh, nw, heads;
    float inv_div_x, inv_div_y;
};
struct Sample {
    float fx, fy;
    int x0, y0;
    float rx, ry, cs, sn, relx, rely;
};

__device__ __forceinline__ Sample make_sample(const RvsaGeom& g, const float* __restrict__ sp, int h, int wi, int wj, int a, int bb) {
    Sample s;
    const int H = g.heads;
    const float offx = sp[2 * h] * g.inv_div_x, offy = sp[2 * h + 1] * g.inv_div_y;
    const float sx = sp[2 * H + 2 * h] + 1.0f, sy = sp[2 * H + 2 * h + 1] + 1.0f;
    const float ang = sp[4 * H + h];
    const float stepx = 2.0f / (float)(g.We - 1), stepy = 2.0f / (float)(g.He - 1);
    const float cenx = -1.0f + stepx * (float)(7 * wj + 3), ceny = -1.0f + stepy * (float)(7 * wi + 3);
    s.relx = (float)(bb - 3) * stepx;
    s.rely = (float)(a - 3) * stepy;
    s.rx = s.relx * sx;
    s.ry = s.rely * sy;
    s.cs = cosf(ang);
    s.sn = sinf(ang);
    const float gx = cenx + (s.rx * s.cs - s.ry * s.sn) + offx;
    const float gy = ceny + (s.ry * s.cs + s.rx * s.sn) + offy;
    float ix = (gx + 1.0f) * 0.5f * (float)(g.We - 1), iy = (gy + 1.0f) * 0.5f * (float)(g.He - 1);
    ix = fminf(fmaxf(ix, -4.0f), (float)g.We + 4.0f);
    iy = fminf(fmaxf(iy, -4.0f), (float)g.He + 4.0f);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    s.x0 = (int)fx0; s.y0 = (int)fy0;
    s.fx = ix - fx0; s.fy = iy - fy0;
    return s;
}
__device__ __forceinline__ int neighbour(const RvsaGeom& g, int x0, int y0, float fx, float fy, int k, float& w) {
    const int dx = k & 1, dy = k >> 1;
    const int xi = x0 + dx, yi = y0 + dy;
    w = (dx ? fx : 1.0f - fx) * (dy ? fy : 1.0f - fy);
    const int tx = xi - g.pad_l, ty = yi - g.pad_t;
    if (xi < 0 || xi > g.We - 1 || yi < 0 || yi > g.He - 1 || tx < 0 || tx >= g.Wp || ty < 0 || ty >= g.Hp) return -1;
    return ty * g.Wp + tx;
}
__device__ __forceinline__ int query_token(const RvsaGeom& g, int n, int wi, int wj) {   // n < 49
    const int a = n / 7, bb = n - 7 * a;
    const int ty = 7 * wi + a - g.pad_t, tx = 7 * wj + bb - g.pad_l;
    return (ty >= 0 && ty < g.Hp && tx >= 0 && tx < g.Wp) ? ty * g.Wp + tx : -1;
}

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }
__device__ __forceinline__ f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint4 ld16(const char* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 ld8x2(const char* p0, const char* p1) {   // two 8-byte LDS reads -> one 8 x bf16 operand
    const uint2 a = *reinterpret_cast<const uint2*>(p0), b = *reinterpret_cast<const uint2*>(p1);
    return make_uint4(a.x, a.y, b.x, b.y);
}
// 8 f32 table values (row r, elements e0..e0+7) -> bf16 operand; zero when the row is out of range
__device__ __forceinline__ uint4 table_frag(const float* __restrict__ tab, int r, int rows, int e0) {
    if (r >= rows) return make_uint4(0, 0, 0, 0);
    const float4 a = *reinterpret_cast<const float4*>(tab + r * HD + e0), b = *reinterpret_cast<const float4*>(tab + r * HD + e0 + 4);
    return pack_bf16x8(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
}
// transposed table operand: lane (d, g) -> tab[8g+e][d], e = 0..7
__device__ __forceinline__ uint4 table_frag_t(const float* __restrict__ tab, int d, int rows, int r0) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (r0 + e) < rows ? tab[(r0 + e) * HD + d] : 0.f;
    return pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
}

// lane = key: bilinear gather of this key's K/V rows (f32 blend of <= 4 bf16 token rows)
__device__ __forceinline__ void gather_kv(const RvsaGeom& g, const Sample& s, const bf16_t* __restrict__ base, int64_t ld, int C, float (&ks)[HD], float (&vs)[HD]) {
    // branch-free: an out-of-map neighbour reads token 0 with weight 0 (a branch around the loads would make hipcc wait for
    // every neighbour separately; this way all 64 row loads are in flight together)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float w;
        const int tok = neighbour(g, s.x0, s.y0, s.fx, s.fy, k, w);
        const int tc = tok >= 0 ? tok : 0;
        w = tok >= 0 ? w : 0.f;
#pragma unroll
        for (int i = 0; i < HD / 8; ++i) {
            float t[8];
            load8(base + C + (int64_t)tc * ld + 8 * i, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) ks[8 * i + e] += w * t[e];
            load8(base + 2 * C + (int64_t)tc * ld + 8 * i, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) vs[8 * i + e] += w * t[e];
        }
    }
}
__device__ __attribute__((aligned(16))) const uint4 g_zero16a = {0u, 0u, 0u, 0u};
// 16-byte fragment of row `tok` (or zeros when tok < 0) without a branch around the load
__device__ __forceinline__ uint4 row_frag(const bf16_t* __restrict__ rows, int64_t ld, int tok, int e0) {
    return ldg16(tok >= 0 ? reinterpret_cast<const char*>(rows + (int64_t)tok * ld + e0) : reinterpret_cast<const char*>(&g_zero16a));
}
__device__ __forceinline__ void put_row_swz(char* img, int row, const float (&v)[HD]) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
        *reinterpret_cast<uint4*>(img + swz(row, c)) = pack_bf16x8(v[8 * c], v[8 * c + 1], v[8 * c + 2], v[8 * c + 3], v[8 * c + 4], v[8 * c + 5], v[8 * c + 6], v[8 * c + 7]);
}
__device__ __forceinline__ void put_col_t(char* img, int col, const float (&v)[HD]) {   // img[d][col] = v[d]
#pragma unroll
    for (int d = 0; d < HD; ++d) *reinterpret_cast<uint16_t*>(img + d * TP + col * 2) = (uint16_t)f32_to_bf16_bits(v[d]);
}
__device__ __forceinline__ void put_col_t_bits(char* img, int col, const uint4 (&rowbits)[8]) {   // 64 bf16 already packed
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint32_t w[4] = {rowbits[c].x, rowbits[c].y, rowbits[c].z, rowbits[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<uint16_t*>(img + (8 * c + 2 * e) * TP + col * 2) = (uint16_t)(w[e] & 0xffffu);
            *reinterpret_cast<uint16_t*>(img + (8 * c + 2 * e + 1) * TP + col * 2) = (uint16_t)(w[e] >> 16);
        }
    }
}

// ===================================================================================================================
// RVSA forward
// ===================================================================================================================
__global__ __launch_bounds__(64) void rvsa_fwd_mfma_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ samp, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                          const float* __restrict__ rel_h, const float* __restrict__ rel_w, const float* __restrict__ bias_table,
                                                          RvsaGeom g, float scale) {
    __shared__ __attribute__((aligned(16))) char Ks[64 * 128];
    __shared__ __attribute__((aligned(16))) char Vt[64 * TP];
    __shared__ float QR[26 * 64];
    __shared__ float tab[176];
    const int lane = threadIdx.x, fr = lane & 15, gq = lane >> 4;
    const int H = g.heads, nW = g.nh * g.nw;
    const int h = blockIdx.x % H, bw = blockIdx.x / H, b = bw / nW, win = bw % nW, wi = win / g.nw, wj = win % g.nw;
    const int C = H * HD, N = g.Hp * g.Wp;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;

    for (int i = lane; i < 169; i += 64) tab[i] = bias_table[i * H + h];
    {   // ---- gather, lane = key
        float ks[HD], vs[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { ks[d] = 0.f; vs[d] = 0.f; }
        if (lane < 49) {
            const Sample s = make_sample(g, samp + (int64_t)bw * 5 * H, h, wi, wj, lane / 7, lane % 7);
            gather_kv(g, s, base, ld, C, ks, vs);
        }
        put_row_swz(Ks, lane, ks);
        put_col_t(Vt, lane, vs);
    }
    // ---- Q fragments (B operand: lane (query fr of tile qt, d chunk gq))
    uint4 qf[4][2];
    int qtok[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int n = 16 * qt + fr;
        qtok[qt] = n < 49 ? query_token(g, n, wi, wj) : -1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[qt][ks] = row_frag(base, ld, qtok[qt], ks * 32 + gq * 8);
    }
    // ---- QR[t*13 + r][query] = q . rel_t[r]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float* tb = t ? rel_w : rel_h;
        const uint4 a0 = table_frag(tb, fr, 13, gq * 8), a1 = table_frag(tb, fr, 13, 32 + gq * 8);
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            acc = mma(a0, qf[qt][0], acc);
            acc = mma(a1, qf[qt][1], acc);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (4 * gq + rr < 13) QR[(t * 13 + 4 * gq + rr) * 64 + 16 * qt + fr] = acc[rr];
        }
    }
    __syncthreads();
    // ---- S^T = Ksel . Q^T
    f32x4_t s[4][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) s[kt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint4 kf[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) kf[kt] = ld16(Ks + swz(16 * kt + fr, ks * 4 + gq));
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) s[kt][qt] = mma(kf[kt], qf[qt][ks], s[kt][qt]);
    }
    // ---- + rel-pos + bias, softmax over keys (in-lane + xor 16/32), P as B operand
    uint4 pf[4][2];
    float inv[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int n = 16 * qt + fr, nq = n < 48 ? n : 48;
        const int aq = (nq * 37) >> 8, bq = nq - 7 * aq;
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * gq + r, kc = key < 48 ? key : 48;   // branch-free: clamped indices + select
                const int ak = (kc * 37) >> 8, bk = kc - 7 * ak, dh = aq - ak + 6, dw = bq - bk + 6;
                float v = scale * s[kt][qt][r] + QR[dh * 64 + n] + QR[(13 + dw) * 64 + n] + tab[dh * 13 + dw];
                v = key < 49 ? v : -INFINITY;
                s[kt][qt][r] = v;
                m = fmaxf(m, v);
            }
        m = xor16_max(m);
        m = xor32_max(m);
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[kt][qt][r] - m);
                s[kt][qt][r] = p;
                l += p;
            }
        l = xor16_sum(l);
        l = xor32_sum(l);
        inv[qt] = 1.0f / l;
        if (gq == 0 && n < 49) lse[(int64_t)blockIdx.x * 49 + n] = m + __logf(l);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            pf[qt][kk] = pack_bf16x8(s[2 * kk][qt][0], s[2 * kk][qt][1], s[2 * kk][qt][2], s[2 * kk][qt][3],
                                     s[2 * kk + 1][qt][0], s[2 * kk + 1][qt][1], s[2 * kk + 1][qt][2], s[2 * kk + 1][qt][3]);
    }
    // ---- O^T = V^T . P^T  (k = key, permuted identically on both operands)
    f32x4_t oa[4][4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) oa[dt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const char* row = Vt + (16 * dt + fr) * TP;
            const uint4 vf = ld8x2(row + (32 * kk + 4 * gq) * 2, row + (32 * kk + 16 + 4 * gq) * 2);
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) oa[dt][qt] = mma(vf, pf[qt][kk], oa[dt][qt]);
        }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt)
        if (qtok[qt] >= 0) {
            bf16_t* op = o + ((int64_t)b * N + qtok[qt]) * C + h * HD + 4 * gq;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                store4(op + 16 * dt, make_float4(oa[dt][qt][0] * inv[qt], oa[dt][qt][1] * inv[qt], oa[dt][qt][2] * inv[qt], oa[dt][qt][3] * inv[qt]));
        }
}

// ===================================================================================================================
// RVSA backward
// LDS: Ks | Vs (row-major, swizzled) | R2 = {Kt} then {Qt | dOt} (transposed images) | QR | dQR | tab | dtab | lses | delta | smp
// ===================================================================================================================
constexpr int SMP_F = 10;   // floats of Sample kept per key

__global__ __launch_bounds__(64) void rvsa_bwd_mfma_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ samp, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                          const float* __restrict__ lse, bf16_t* __restrict__ dqkv, float* __restrict__ dkv, float* __restrict__ dsamp,
                                                          float* __restrict__ rel_part, float* __restrict__ tab_part,
                                                          const float* __restrict__ rel_h, const float* __restrict__ rel_w, const float* __restrict__ bias_table,
                                                          RvsaGeom g, float scale) {
    __shared__ __attribute__((aligned(16))) char Ks[64 * 128];
    __shared__ __attribute__((aligned(16))) char Vs[64 * 128];
    __shared__ __attribute__((aligned(16))) char R2[2 * 64 * TP];
    __shared__ float QR[26 * 64];
    __shared__ float dQR[26 * 64];
    __shared__ float tab[176];
    __shared__ float dtab[176];
    __shared__ float lses[64];
    __shared__ float delta[64];
    __shared__ float smp[SMP_F * 64];
    char* Kt = R2;
    char* Qt = R2;
    char* dOt = R2 + 64 * TP;
    const int lane = threadIdx.x, fr = lane & 15, gq = lane >> 4;
    const int H = g.heads, nW = g.nh * g.nw;
    const int h = blockIdx.x % H, bw = blockIdx.x / H, b = bw / nW, win = bw % nW, wi = win / g.nw, wj = win % g.nw;
    const int C = H * HD, N = g.Hp * g.Wp;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* dob = dout + (int64_t)b * N * C + h * HD;

    for (int i = lane; i < 176; i += 64) {
        tab[i] = i < 169 ? bias_table[i * H + h] : 0.f;
        dtab[i] = 0.f;
    }
    for (int i = lane; i < 26 * 64; i += 64) dQR[i] = 0.f;
    {   // ---- gather, lane = key
        float ks[HD], vs[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { ks[d] = 0.f; vs[d] = 0.f; }
        Sample s;
        s.fx = 0.f; s.fy = 0.f; s.x0 = -100; s.y0 = -100; s.rx = 0.f; s.ry = 0.f; s.cs = 1.f; s.sn = 0.f; s.relx = 0.f; s.rely = 0.f;
        if (lane < 49) {
            s = make_sample(g, samp + (int64_t)bw * 5 * H, h, wi, wj, lane / 7, lane % 7);
            gather_kv(g, s, base, ld, C, ks, vs);
        }
        smp[0 * 64 + lane] = s.fx; smp[1 * 64 + lane] = s.fy; smp[2 * 64 + lane] = __int_as_float(s.x0); smp[3 * 64 + lane] = __int_as_float(s.y0);
        smp[4 * 64 + lane] = s.rx; smp[5 * 64 + lane] = s.ry; smp[6 * 64 + lane] = s.cs; smp[7 * 64 + lane] = s.sn;
        smp[8 * 64 + lane] = s.relx; smp[9 * 64 + lane] = s.rely;
        put_row_swz(Ks, lane, ks);
        put_row_swz(Vs, lane, vs);
        put_col_t(Kt, lane, ks);
    }
    {   // ---- lane = query: delta = dO . O, lse
        float dl = 0.f, ls = 0.f;
        const int tok = lane < 49 ? query_token(g, lane, wi, wj) : -1;
        {
            const int tc = tok >= 0 ? tok : 0;
#pragma unroll
            for (int i = 0; i < HD / 8; ++i) {
                float a[8], c[8];
                load8(dob + (int64_t)tc * C + 8 * i, a);
                load8(o + ((int64_t)b * N + tc) * C + h * HD + 8 * i, c);
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += a[e] * c[e];
            }
            dl = tok >= 0 ? dl : 0.f;
        }
        if (lane < 49) ls = lse[(int64_t)blockIdx.x * 49 + lane];
        delta[lane] = dl;
        lses[lane] = ls;
    }
    // ---- Q / dO fragments (lane (row fr of tile, d chunk gq)); usable as A or B operand
    uint4 qf[4][2], dof[4][2];
    int qtok[4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int n = 16 * qt + fr;
        qtok[qt] = n < 49 ? query_token(g, n, wi, wj) : -1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[qt][ks] = row_frag(base, ld, qtok[qt], ks * 32 + gq * 8);
            dof[qt][ks] = row_frag(dob, C, qtok[qt], ks * 32 + gq * 8);
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float* tb = t ? rel_w : rel_h;
        const uint4 a0 = table_frag(tb, fr, 13, gq * 8), a1 = table_frag(tb, fr, 13, 32 + gq * 8);
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            acc = mma(a0, qf[qt][0], acc);
            acc = mma(a1, qf[qt][1], acc);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (4 * gq + rr < 13) QR[(t * 13 + 4 * gq + rr) * 64 + 16 * qt + fr] = acc[rr];
        }
    }
    __syncthreads();

    // ================= phase A: lane (query; 4 keys) -> dQ, dQR, dtab =====================================================
    {
        uint4 kf[4][2], vf[4][2], rhT[4], rwT[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[kt][ks] = ld16(Ks + swz(16 * kt + fr, ks * 4 + gq));
                vf[kt][ks] = ld16(Vs + swz(16 * kt + fr, ks * 4 + gq));
            }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            rhT[dt] = table_frag_t(rel_h, 16 * dt + fr, 13, 8 * gq);
            rwT[dt] = table_frag_t(rel_w, 16 * dt + fr, 13, 8 * gq);
        }
#pragma unroll 1
        for (int qt = 0; qt < 4; ++qt) {
            const int n = 16 * qt + fr, nq = n < 48 ? n : 48;
            const int aq = (nq * 37) >> 8, bq = nq - 7 * aq;
            const float ls = lses[n], dl = delta[n];
            f32x4_t sT[4], dpT[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                sT[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                dpT[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    sT[kt] = mma(kf[kt][ks], qf[qt][ks], sT[kt]);
                    dpT[kt] = mma(vf[kt][ks], dof[qt][ks], dpT[kt]);
                }
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kt + 4 * gq + r, kc = key < 48 ? key : 48;
                    const int ak = (kc * 37) >> 8, bk = kc - 7 * ak, dh = aq - ak + 6, dw = bq - bk + 6;
                    const float v = scale * sT[kt][r] + QR[dh * 64 + n] + QR[(13 + dw) * 64 + n] + tab[dh * 13 + dw];
                    float p = __expf(fminf(v - ls, 30.f));
                    p = (key < 49 && n < 49) ? p : 0.f;        // masked pairs contribute 0 (branch-free LDS atomics)
                    const float ds = p * (dpT[kt][r] - dl);
                    atomicAdd(&dQR[dh * 64 + n], ds);
                    atomicAdd(&dQR[(13 + dw) * 64 + n], ds);
                    atomicAdd(&dtab[dh * 13 + dw], ds);
                    sT[kt][r] = ds * scale;
                    // P^T / dS^T images for the key-major phase (Ks/Vs are dead: their fragments live in kf/vf): [key][query] bf16,
                    // 16-byte slots XOR-swizzled by the key so the 8-byte operand reads of phase B are conflict-free
                    const int off = key * 128 + ((n * 2) ^ ((key & 7) << 4));
                    *reinterpret_cast<uint16_t*>(Ks + off) = (uint16_t)f32_to_bf16_bits(p);
                    *reinterpret_cast<uint16_t*>(Vs + off) = (uint16_t)f32_to_bf16_bits(ds * scale);
                }
            uint4 dsf[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                dsf[kk] = pack_bf16x8(sT[2 * kk][0], sT[2 * kk][1], sT[2 * kk][2], sT[2 * kk][3], sT[2 * kk + 1][0], sT[2 * kk + 1][1], sT[2 * kk + 1][2], sT[2 * kk + 1][3]);
            __syncthreads();   // dQR rows of these 16 queries are complete
            float e[8], f[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int r = 8 * gq + x;
                e[x] = r < 13 ? dQR[r * 64 + n] : 0.f;
                f[x] = r < 13 ? dQR[(13 + r) * 64 + n] : 0.f;
            }
            const uint4 eh = pack_bf16x8(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
            const uint4 ew = pack_bf16x8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                const char* row = Kt + (16 * dt + fr) * TP;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) acc = mma(ld8x2(row + (32 * kk + 4 * gq) * 2, row + (32 * kk + 16 + 4 * gq) * 2), dsf[kk], acc);
                acc = mma(rhT[dt], eh, acc);
                acc = mma(rwT[dt], ew, acc);
                if (qtok[qt] >= 0) store4(dqkv + ((int64_t)b * N + qtok[qt]) * ld + h * HD + 16 * dt + 4 * gq, make_float4(acc[0], acc[1], acc[2], acc[3]));
            }
        }
    }
    __syncthreads();
    // ================= stage Q^T, dO^T (lane = query); R2 no longer holds K^T =============================================
    {
        const int tok = lane < 49 ? query_token(g, lane, wi, wj) : -1;
        uint4 rq[8], rd[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            rq[c] = row_frag(base, ld, tok, 8 * c);
            rd[c] = row_frag(dob, C, tok, 8 * c);
        }
        put_col_t_bits(Qt, lane, rq);
        put_col_t_bits(dOt, lane, rd);
    }
    __syncthreads();
    // ================= table gradients: rel_part[t*13 + r][d] = sum_query dQR[t*13+r][query] * Q[query][d] ================
    {
        float* rp = rel_part + (int64_t)blockIdx.x * 26 * HD;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            uint4 af[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) v[x] = fr < 13 ? dQR[(t * 13 + fr) * 64 + 32 * ks + 8 * gq + x] : 0.f;
                af[ks] = pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                const char* row = Qt + (16 * dt + fr) * TP;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) acc = mma(af[ks], ld8x2(row + (32 * ks + 8 * gq) * 2, row + (32 * ks + 8 * gq + 4) * 2), acc);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    if (4 * gq + rr < 13) rp[(t * 13 + 4 * gq + rr) * HD + 16 * dt + fr] = acc[rr];
            }
        }
        for (int i = lane; i < 169; i += 64) tab_part[((int64_t)bw * H + h) * 169 + i] = dtab[i];   // (window, head, 169): contiguous per workgroup
    }
    // ================= phase B: lane (key; 4 queries) -> dK_sel^T, dV_sel^T, scatter, coordinate gradients ================
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    {
        uint4 qtf[4][2], dotf[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const char* rq = Qt + (16 * dt + fr) * TP;
                const char* rd = dOt + (16 * dt + fr) * TP;
                qtf[dt][kk] = ld8x2(rq + (32 * kk + 4 * gq) * 2, rq + (32 * kk + 16 + 4 * gq) * 2);
                dotf[dt][kk] = ld8x2(rd + (32 * kk + 4 * gq) * 2, rd + (32 * kk + 16 + 4 * gq) * 2);
            }
#pragma unroll 1
        for (int kt = 0; kt < 4; ++kt) {
            const int key = 16 * kt + fr, kc = key < 48 ? key : 48;
            // P and scale*dS of this key row, written by phase A into the (dead) Ks / Vs regions: operands for queries
            // perm(gq, e) = {32kk + 4gq + e, 32kk + 16 + 4gq + e}  (no recomputation of S / exp in the key-major orientation)
            uint4 pfb[2], dsfb[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int o0 = key * 128 + (((32 * kk + 4 * gq) * 2) ^ ((key & 7) << 4));
                const int o1 = key * 128 + (((32 * kk + 16 + 4 * gq) * 2) ^ ((key & 7) << 4));
                pfb[kk] = ld8x2(Ks + o0, Ks + o1);
                dsfb[kk] = ld8x2(Vs + o0, Vs + o1);
            }
            f32x4_t dks[4], dvs[4];   // [dt]: lane (key = 16kt + fr; d = 16dt + 4gq + r)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dks[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                dvs[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    dks[dt] = mma(qtf[dt][kk], dsfb[kk], dks[dt]);
                    dvs[dt] = mma(dotf[dt][kk], pfb[kk], dvs[dt]);
                }
            }
            // ---- scatter: the SAME products in the other orientation (operands swapped: lane = (d = 16dt + fr; keys 4gq + r)) so that
            //      one atomic instruction covers 16 consecutive channels (64 B) of 4 tokens instead of 4-byte pieces of 16 tokens
            {
                f32x4_t dk2[4], dv2[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dk2[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    dv2[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        dk2[dt] = mma(dsfb[kk], qtf[dt][kk], dk2[dt]);
                        dv2[dt] = mma(pfb[kk], dotf[dt][kk], dv2[dt]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key2 = 16 * kt + 4 * gq + r, k2 = key2 < 48 ? key2 : 48;
                    const float fx2 = smp[0 * 64 + k2], fy2 = smp[1 * 64 + k2];
                    const int x02 = __float_as_int(smp[2 * 64 + k2]), y02 = __float_as_int(smp[3 * 64 + k2]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float w;
                        const int tok = key2 < 49 ? neighbour(g, x02, y02, fx2, fy2, k, w) : -1;
                        if (tok >= 0) {
                            float* drow = dkv + ((int64_t)b * N + tok) * (2 * C) + h * HD + fr;
#pragma unroll
                            for (int dt = 0; dt < 4; ++dt) {
                                atomicAdd(drow + 16 * dt, w * dk2[dt][r]);
                                atomicAdd(drow + C + 16 * dt, w * dv2[dt][r]);
                            }
                        }
                    }
                }
            }
            // ---- coordinate gradients of this lane's key (16kt + fr)
            const float fx = smp[0 * 64 + kc], fy = smp[1 * 64 + kc];
            const int x0 = __float_as_int(smp[2 * 64 + kc]), y0 = __float_as_int(smp[3 * 64 + kc]);
            float dix = 0.f, diy = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float w;
                const int tok = key < 49 ? neighbour(g, x0, y0, fx, fy, k, w) : -1;
                float dot = 0.f;
                {   // loads unconditional on a clamped token (see gather_kv); only the atomics are predicated
                    const int tc = tok >= 0 ? tok : 0;
                    const bf16_t* krow = base + C + (int64_t)tc * ld + 4 * gq;
                    const bf16_t* vrow = base + 2 * C + (int64_t)tc * ld + 4 * gq;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const float4 kv = load4(krow + 16 * dt), vv = load4(vrow + 16 * dt);
                        dot += dks[dt][0] * kv.x + dks[dt][1] * kv.y + dks[dt][2] * kv.z + dks[dt][3] * kv.w
                             + dvs[dt][0] * vv.x + dvs[dt][1] * vv.y + dvs[dt][2] * vv.z + dvs[dt][3] * vv.w;
                    }
                    dot = tok >= 0 ? dot : 0.f;
                }
                dot = xor16_sum(dot);
                dot = xor32_sum(dot);
                const int dx = k & 1, dy = k >> 1;
                dix += dot * (dy ? fy : 1.0f - fy) * (dx ? 1.0f : -1.0f);
                diy += dot * (dx ? fx : 1.0f - fx) * (dy ? 1.0f : -1.0f);
            }
            if (gq == 0 && key < 49) {
                const float rx = smp[4 * 64 + kc], ry = smp[5 * 64 + kc], cs = smp[6 * 64 + kc], sn = smp[7 * 64 + kc];
                const float dgx = dix * 0.5f * (float)(g.We - 1), dgy = diy * 0.5f * (float)(g.He - 1);
                v0 += dgx * g.inv_div_x;
                v1 += dgy * g.inv_div_y;
                v2 += (dgx * cs + dgy * sn) * smp[8 * 64 + kc];
                v3 += (-dgx * sn + dgy * cs) * smp[9 * 64 + kc];
                v4 += dgx * (-rx * sn - ry * cs) + dgy * (-ry * sn + rx * cs);
            }
        }
    }
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2); v3 = wave_sum(v3); v4 = wave_sum(v4);
    if (lane == 0) {
        float* dp = dsamp + (int64_t)bw * 5 * H;
        dp[2 * h] = v0; dp[2 * h + 1] = v1; dp[2 * H + 2 * h] = v2; dp[2 * H + 2 * h + 1] = v3; dp[4 * H + h] = v4;
    }
}

RvsaGeom make_geom(int64_t Hp, int64_t Wp, int64_t heads) {
    RvsaGeom g;
    const int pad_h = (int)((7 - Hp % 7) % 7), pad_w = (int)((7 - Wp % 7) % 7);
    g.Hp = (int)Hp; g.Wp = (int)Wp;
    g.pad_t = pad_h / 2; g.pad_l = pad_w / 2;
    g.He = (int)Hp + pad_h; g.We = (int)Wp + pad_w;
    g.nh = g.He / 7; g.nw = g.We / 7;
    g.heads = (int)heads;
    g.inv_div_x = 1.0f / (float)(Hp / 7);
    g.inv_div_y = 1.0f / (float)(Wp / 7);
    return g;
}

}  // namespace

// single-wave-per-problem forward (kept for A/B; the shipped launcher is the 4-wave kernel in attn_rvsa_fwd4.hip)
int mtp_rvsa_fwd1_mfma_launch(const void* qkv, const float* samp, void* o, float* lse, const float* rel_h, const float* rel_w, const float* bias_table,
                             int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s) {
    const RvsaGeom g = make_geom(Hp, Wp, heads);
    hipLaunchKernelGGL(rvsa_fwd_mfma_kernel, dim3((unsigned)(B * g.nh * g.nw * heads)), dim3(64), 0, s, (const bf16_t*)qkv, samp, (bf16_t*)o, lse,
                       rel_h, rel_w, bias_table, g, scale);
    return mtp_launch_status();
}

// single-wave-per-problem backward (kept for A/B; the shipped launcher is the 4-wave kernel in attn_rvsa_bwd4.hip)
int mtp_rvsa_bwd1_mfma_launch(const void* qkv, const float* samp, const void* o, const void* dout, const float* lse, void* dqkv, float* dkv, float* dsamp,
                             float* rel_part, float* tab_part, const float* rel_h, const float* rel_w, const float* bias_table,
                             int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s) {
    const RvsaGeom g = make_geom(Hp, Wp, heads);
    hipLaunchKernelGGL(rvsa_bwd_mfma_kernel, dim3((unsigned)(B * g.nh * g.nw * heads)), dim3(64), 0, s, (const bf16_t*)qkv, samp, (const bf16_t*)o, (const bf16_t*)dout, lse,
                       (bf16_t*)dqkv, dkv, dsamp, rel_part, tab_part, rel_h, rel_w, bias_table, g, scale);
    return mtp_launch_status();
}

// ---- full-attention MFMA kernels: see attn_full_mfma.hip
