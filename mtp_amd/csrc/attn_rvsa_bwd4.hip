// 4-wave variant of the RVSA backward (see attn_mfma.hip for the algorithm and the single-wave forward).
#include <stdlib.h>

#include "attn_mfma.h"
#include "common.h"

namespace {

constexpr int HD = 64;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v_t;
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
    s.cs = __cosf(ang);      // v_cos / v_sin (abs error ~1e-6 on |ang| < pi): this kernel and the forward share the expression
    s.sn = __sinf(ang);
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
    // unconditional loads on a clamped row, masked afterwards: a branch around the loads makes hipcc wait for them inside it
    const int rc = r < rows ? r : rows - 1;
    const float m = r < rows ? 1.0f : 0.0f;
    const float4 a = *reinterpret_cast<const float4*>(tab + rc * HD + e0), b = *reinterpret_cast<const float4*>(tab + rc * HD + e0 + 4);
    return pack_bf16x8(m * a.x, m * a.y, m * a.z, m * a.w, m * b.x, m * b.y, m * b.z, m * b.w);
}
// transposed table operand: lane (d, g) -> tab[8g+e][d], e = 0..7
__device__ __forceinline__ uint4 table_frag_t(const float* __restrict__ tab, int d, int rows, int r0) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {      // unconditional loads on clamped rows (see table_frag)
        const int r = r0 + e;
        const float t = tab[(r < rows ? r : rows - 1) * HD + d];
        v[e] = r < rows ? t : 0.f;
    }
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


constexpr int SMP_F = 18;      // per key: fx fy x0 y0 rx ry cs sn relx rely | tok[4] | w[4] (the neighbour records, one computation per key)

// ===================================================================================================================
// RVSA backward, 4 waves per (image, window, head): wave w owns query tile w in the query-major phase and key tile w in the
// key-major phase, so the problem's critical path is 4x shorter and 12 waves share a CU (3 workgroups x 4) instead of 3.
// LDS: Ks | Vs (K_sel / V_sel rows, later P^T / dS^T) | R2 = {K^T} then {Q^T | dO^T} | QR | dQR | tab | lses | delta | smp | vsum
// ===================================================================================================================
// MODE: the scatter form, a compile-time choice: 4 = rvsa_scatter_gemm_kernel takes the dK_sel / dV_sel rows (token grids it fits: the default),
// 1 = f32 atomics per token tile inside this kernel (larger grids; MTP_RVSA_SCATTER=dense forces it).  (Round 1's per-(key, corner) atomics and
// the phase-timing ablation switches were removed in round 4; measurements in DESIGN section 4.)
typedef short tr4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 rows_frag_tr(const char* img, int row0, int dt, int fr) {   // (d = 16 dt + fr; rows row0..+3, row0+16..+19)
    const int c = 16 * dt + 4 * (fr & 3);
    const int ra = row0 + (fr >> 2), rb = ra + 16;
    const int oa = ra * 128 + (((c >> 3) ^ (ra & 7)) << 4) + (c & 7) * 2, ob = rb * 128 + (((c >> 3) ^ (rb & 7)) << 4) + (c & 7) * 2;
    const tr4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4s_t*)(img + oa));
    const tr4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4s_t*)(img + ob));
    const uint2 l = __builtin_bit_cast(uint2, lo), hh = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, hh.x, hh.y);
}

// ===================================================================================================================
// Round 6: the same backward WITHOUT transposed LDS images.  rvsa_bwd4_mfma_kernel below writes K^T, Q^T | dO^T and P^T | dS^T in 2-byte units (the bank
// conflicts of profiles/r04_pmc_sq_rvsa.txt) and loads Q / dO from global memory a second time to build Q^T | dO^T.  Here every image is row-major ([row][16-B
// chunk ^ (row & 7)], 16- or 8-byte stores) and every operand that needs the other orientation comes out of ds_read_b64_tr_b16 (rows_frag_tr): K^T from the
// K_sel rows, Q^T | dO^T from row images written out of the fragments the wave loaded at the top, the key-major P / dS fragments of phase B from [query][key]
// images.  Same MFMAs on the same operand values: dq, dK_sel | dV_sel, dsamp and the bias-table partials are bit-identical to the kernel below; the rel-pos table
// partials sum the queries in the transpose read's slot order.  One barrier and one global round trip fewer per workgroup.  Dense-scatter grids keep the old kernel.
// LDS: Ks | Vs (K_sel / V_sel rows, later Q / dO rows) | R2 = P | dS [query][key], later dK_sel | dV_sel rows | QR | dQR | tab | lses | delta | smp | vsum
// ===================================================================================================================
__global__ __launch_bounds__(256, 3) void rvsa_bwd5_mfma_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ samp, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                            const float* __restrict__ lse, bf16_t* __restrict__ dqkv, float* __restrict__ dkv, float* __restrict__ dsamp,
                                                            float* __restrict__ rel_part, float* __restrict__ tab_part,
                                                            const float* __restrict__ rel_h, const float* __restrict__ rel_w, const float* __restrict__ bias_table,
                                                            RvsaGeom g, float scale) {
    __shared__ __attribute__((aligned(16))) char Ks[64 * 128];
    __shared__ __attribute__((aligned(16))) char Vs[64 * 128];
    __shared__ __attribute__((aligned(16))) char R2[2 * 64 * 128];      // P | dS images [query][key], later the dK_sel | dV_sel rows
    __shared__ float QR[26 * 64];      // (a pitch of 80 floats would take the two-way conflicts out of the phase-A reads, but 1.6 KB more LDS is a workgroup per CU less: 130 vs 108 us)
    constexpr int DQP = 65;      // row pitch of dQR: with 64 the table-gradient reads (16 lanes = 16 rows of one column) were 16-way bank conflicts --
                                 // all of the kernel's SQ_LDS_BANK_CONFLICT (1000 cycles per wave; round 6)
    __shared__ float dQR[26 * DQP];
    __shared__ float tab[176];
    __shared__ float lses[64];
    __shared__ float delta[64];
    __shared__ __attribute__((aligned(16))) float smp[SMP_F * 64];
    __shared__ float vsum[4 * 8];
    char* Pimg = R2;
    char* dSimg = R2 + 64 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, gq = lane >> 4;
    const int H = g.heads, nW = g.nh * g.nw;
    const int h = blockIdx.x % H, bw = blockIdx.x / H, b = bw / nW, win = bw % nW, wi = win / g.nw, wj = win % g.nw;
    const int C = H * HD, N = g.Hp * g.Wp;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* dob = dout + (int64_t)b * N * C + h * HD;
    const uint32_t ld2 = (uint32_t)(6 * C);      // bytes per qkv row

    if (tid < 176) {
        tab[tid] = tid < 169 ? bias_table[tid * H + h] : 0.f;
    }
    for (int i = tid; i < 26 * DQP; i += 256) dQR[i] = 0.f;
    // ---- this wave's query tile: Q / dO fragments and QR = tables x Q^T  (first: these loads depend on nothing, so they are in
    // flight together with the gather's)
    const int qt = wave;
    const int nA = 16 * qt + fr;
    const int qtokA = nA < 49 ? query_token(g, nA, wi, wj) : -1;
    uint4 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = row_frag(base, ld, qtokA, ks * 32 + gq * 8);
        dof[ks] = row_frag(dob, C, qtokA, ks * 32 + gq * 8);
    }
    // ---- gather: lane = (key of a group of 8, 16-B chunk of the 64-channel row) -- one wave instruction reads 8 WHOLE 128-B rows --
    // and delta = dO . O per query in the same lane arrangement.  Every lane computes the sample position of its own two keys (8
    // lanes share one; no wave-0-only phase, no barrier before the gather) and the chunk-0 lanes publish it in `smp` for the later
    // phases.  ALL loads of the phase (2 key groups x 4 neighbours x {K, V} + 2 query groups x {dO, O}) are issued before the
    // first one is used: the phase is a chain of HBM round trips (the qkv rows of a window are cold), and issued group by group
    // they cost one latency each (phase timing, MTP_RVSA_STOP).
    {
        const int kl = lane >> 3, ch = lane & 7;
        const uint32_t ch16 = (uint32_t)(16 * ch);
        uint4 kq[2][4], vq[2][4], da[2], oc[2];
        float wq[2][4];
        int qtok[2];
        {   // ---- ONE sample computation per key (round 5): lane l < 16 of a wave owns key (wave + 4 (l >> 3)) * 8 + (l & 7) -- the 16 keys whose rows this
            // wave gathers -- and publishes the record (position, the four neighbour tokens and weights) in `smp`; the other lanes repeat it idly (same
            // instruction stream).  The wave reads its OWN records back (LDS operations of one wave execute in order: no block barrier), so the gather
            // costs one sample + four neighbours per wave instruction stream instead of two + eight, and the coordinate-gradient phase recomputes nothing.
            const int l16 = lane & 15;
            const int key = (wave + 4 * (l16 >> 3)) * 8 + (l16 & 7), kc = key < 48 ? key : 48;
            Sample sm = make_sample(g, samp + (int64_t)bw * 5 * H, h, wi, wj, kc / 7, kc % 7);   // (unconditional: no branch around its loads)
            if (key >= 49) { sm.x0 = -100; sm.y0 = -100; sm.fx = 0.f; sm.fy = 0.f; }      // keys >= 49 are zero rows: every neighbour outside -> weight 0
            float wv[4];
            int tk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float w;
                const int tok = neighbour(g, sm.x0, sm.y0, sm.fx, sm.fy, k, w);
                tk[k] = tok;
                wv[k] = tok >= 0 ? w : 0.f;
            }
            if (lane < 16) {
                smp[0 * 64 + key] = sm.fx; smp[1 * 64 + key] = sm.fy; smp[2 * 64 + key] = __int_as_float(sm.x0); smp[3 * 64 + key] = __int_as_float(sm.y0);
                smp[4 * 64 + key] = sm.rx; smp[5 * 64 + key] = sm.ry; smp[6 * 64 + key] = sm.cs; smp[7 * 64 + key] = sm.sn;
                smp[8 * 64 + key] = sm.relx; smp[9 * 64 + key] = sm.rely;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    smp[(10 + k) * 64 + key] = __int_as_float(tk[k]);
                    smp[(14 + k) * 64 + key] = wv[k];
                }
            }
        }
        // (the records cross lanes: a wavefront-scope release + wave barrier keeps the compiler from moving the read-back above the writes; the hardware
        //  executes one wave's LDS operations in order, so this costs no instruction -- ADVICE r05)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int key = (wave + 4 * gi) * 8 + kl;      // 8 groups = 64 key rows
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int tok = __float_as_int(smp[(10 + k) * 64 + key]);
                const int tc = tok >= 0 ? tok : 0;
                wq[gi][k] = smp[(14 + k) * 64 + key];
                const uint32_t roff = (uint32_t)tc * ld2 + ch16;      // 32-bit byte offsets off the (image, head) base: no 64-bit address arithmetic
                kq[gi][k] = ldg16_at(base + C, roff);
                vq[gi][k] = ldg16_at(base + 2 * C, roff);
            }
            const int n = (wave + 4 * gi) * 8 + kl;
            qtok[gi] = n < 49 ? query_token(g, n, wi, wj) : -1;
            const int tc = qtok[gi] >= 0 ? qtok[gi] : 0;
            da[gi] = ldg16_at(dob, (uint32_t)tc * (uint32_t)(2 * C) + ch16);
            oc[gi] = ldg16_at(o + (int64_t)b * N * C + h * HD, (uint32_t)tc * (uint32_t)(2 * C) + ch16);
        }
        // QR = tables x Q^T of this wave's query tile, while the gather's loads are in flight
    #pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* tb = t ? rel_w : rel_h;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            acc = mma(table_frag(tb, fr, 13, gq * 8), qf[0], acc);
            acc = mma(table_frag(tb, fr, 13, 32 + gq * 8), qf[1], acc);
    #pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (4 * gq + rr < 13) QR[(t * 13 + 4 * gq + rr) * 64 + nA] = acc[rr];
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int key = (wave + 4 * gi) * 8 + kl;
            float ks[8], vs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { ks[e] = 0.f; vs[e] = 0.f; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t kw[4] = {kq[gi][k].x, kq[gi][k].y, kq[gi][k].z, kq[gi][k].w}, vw[4] = {vq[gi][k].x, vq[gi][k].y, vq[gi][k].z, vq[gi][k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ks[2 * e] += wq[gi][k] * bf16_bits_to_f32(kw[e] & 0xffffu); ks[2 * e + 1] += wq[gi][k] * bf16_bits_to_f32(kw[e] >> 16);
                    vs[2 * e] += wq[gi][k] * bf16_bits_to_f32(vw[e] & 0xffffu); vs[2 * e + 1] += wq[gi][k] * bf16_bits_to_f32(vw[e] >> 16);
                }
            }
            *reinterpret_cast<uint4*>(Ks + swz(key, ch)) = pack_bf16x8(ks[0], ks[1], ks[2], ks[3], ks[4], ks[5], ks[6], ks[7]);
            *reinterpret_cast<uint4*>(Vs + swz(key, ch)) = pack_bf16x8(vs[0], vs[1], vs[2], vs[3], vs[4], vs[5], vs[6], vs[7]);
            // delta = dO . O of query n = key index, lse
            const uint32_t aw[4] = {da[gi].x, da[gi].y, da[gi].z, da[gi].w}, cw[4] = {oc[gi].x, oc[gi].y, oc[gi].z, oc[gi].w};
            float dl = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) dl = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, aw[e]), __builtin_bit_cast(bf16x2v_t, cw[e]), dl, false);
            dl += lane_xor<1>(dl);
            dl += lane_xor<2>(dl);
            dl += lane_xor<4>(dl);
            if (ch == 0) {
                delta[key] = qtok[gi] >= 0 ? dl : 0.f;
                lses[key] = key < 49 ? lse[(int64_t)blockIdx.x * 49 + key] : 0.f;
            }
        }
    }
    __syncthreads();

    // ================= phase A: wave = query tile; lane (query; 4 keys) -> dQ, dQR, P^T / dS^T images ===============
    {
        uint4 kf[4][2], vf[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[kt][ks] = ld16(Ks + swz(16 * kt + fr, ks * 4 + gq));
                vf[kt][ks] = ld16(Vs + swz(16 * kt + fr, ks * 4 + gq));
            }
        const int n = nA, nq = n < 48 ? n : 48;
        const int aq = (nq * 37) >> 8, bq = nq - 7 * aq;
        const float ls = lses[n], dl = delta[n];
        f32x4_t sT[4], dpT[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            sT[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dpT[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                sT[kt] = mma(kf[kt][ks], qf[ks], sT[kt]);
                dpT[kt] = mma(vf[kt][ks], dof[ks], dpT[kt]);
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            float pv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * gq + r, kc = key < 48 ? key : 48;
                const int ak = (kc * 37) >> 8, bk = kc - 7 * ak, dh = aq - ak + 6, dw = bq - bk + 6;
                const float v = scale * sT[kt][r] + QR[dh * 64 + n] + QR[(13 + dw) * 64 + n] + tab[dh * 13 + dw];
                float p = __expf(fminf(v - ls, 30.f));
                p = (key < 49 && n < 49) ? p : 0.f;
                const float ds = p * (dpT[kt][r] - dl);
                sT[kt][r] = ds * scale;
                pv[r] = p;
            }
            // P and dS as [query][key] rows (round 6): the lane's four keys of this tile are 8 contiguous bytes of row n -- one store per image and key tile instead
            // of four 2-byte stores into a [key][query] image; phase B reads its (key-major) fragments back through ds_read_b64_tr_b16
            const int o = swz(n, 2 * kt + (gq >> 1)) + 8 * (gq & 1);
            *reinterpret_cast<uint2*>(Pimg + o) = make_uint2(pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]));
            *reinterpret_cast<uint2*>(dSimg + o) = make_uint2(pack_bf16x2(sT[kt][0], sT[kt][1]), pack_bf16x2(sT[kt][2], sT[kt][3]));
        }
        uint4 dsf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            dsf[kk] = pack_bf16x8(sT[2 * kk][0], sT[2 * kk][1], sT[2 * kk][2], sT[2 * kk][3], sT[2 * kk + 1][0], sT[2 * kk + 1][1], sT[2 * kk + 1][2], sT[2 * kk + 1][3]);
        {   // d(q.Rh)[q][aq - ak + 6] = sum over the 7 keys of window row ak of dS (same for columns): segmented sums done as
            // E[a][key] (0/1 indicator, MFMA A layout) x dS^T on the matrix cores -- LDS float atomics cost ~190 LDS cycles per
            // instruction here and were 40 % of this kernel's wave time (SQ_WAIT_INST_LDS).
            f32x4_t dqh = {0.f, 0.f, 0.f, 0.f}, dqw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint32_t wh[4] = {0u, 0u, 0u, 0u}, ww[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = 32 * kk + 4 * gq + (j & 3) + (j >> 2) * 16;
                    const int ak = (key * 37) >> 8, bk = key - 7 * ak;
                    const uint32_t one = 0x3f80u << ((j & 1) * 16);
                    wh[j >> 1] |= (key < 49 && ak == fr) ? one : 0u;
                    ww[j >> 1] |= (key < 49 && bk == fr) ? one : 0u;
                }
                dqh = mma(make_uint4(wh[0], wh[1], wh[2], wh[3]), dsf[kk], dqh);
                dqw = mma(make_uint4(ww[0], ww[1], ww[2], ww[3]), dsf[kk], dqw);
            }
            const float inv_scale = 1.0f / scale;   // dsf carries dS * scale
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 4 * gq + r;
                if (a < 7) {
                    dQR[(aq - a + 6) * DQP + n] = dqh[r] * inv_scale;
                    dQR[(13 + bq - a + 6) * DQP + n] = dqw[r] * inv_scale;
                }
            }
        }
        __syncthreads();   // dQR / P^T / dS^T complete
        float e[8], f[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const int r = 8 * gq + x;
            e[x] = r < 13 ? dQR[r * DQP + n] : 0.f;
            f[x] = r < 13 ? dQR[(13 + r) * DQP + n] : 0.f;
        }
        const uint4 eh = pack_bf16x8(e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7]);
        const uint4 ew = pack_bf16x8(f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) acc = mma(rows_frag_tr(Ks, 32 * kk + 4 * gq, dt, fr), dsf[kk], acc);      // K^T fragment: (d = 16 dt + fr; keys 32 kk + 4 gq .. + 3, + 16 ..)
            acc = mma(table_frag_t(rel_h, 16 * dt + fr, 13, 8 * gq), eh, acc);
            acc = mma(table_frag_t(rel_w, 16 * dt + fr, 13, 8 * gq), ew, acc);
            if (qtokA >= 0) store4(dqkv + ((int64_t)b * N + qtokA) * ld + h * HD + 16 * dt + 4 * gq, make_float4(acc[0], acc[1], acc[2], acc[3]));
        }
    }
    __syncthreads();   // every wave is past its last read of the K_sel / V_sel rows: Ks | Vs become the Q | dO row images
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {      // (out of the fragments loaded at the top: no second trip to global memory, no 2-byte transposing stores -- round 6)
        *reinterpret_cast<uint4*>(Ks + swz(nA, ks * 4 + gq)) = qf[ks];
        *reinterpret_cast<uint4*>(Vs + swz(nA, ks * 4 + gq)) = dof[ks];
    }
    __syncthreads();
    {   // ---- table gradients: wave = d tile
        float* rp = rel_part + (int64_t)blockIdx.x * 26 * HD;
        const int dt = wave;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) v[x] = fr < 13 ? dQR[(t * 13 + fr) * DQP + 32 * ks + 4 * gq + (x & 3) + (x >> 2) * 16] : 0.f;      // queries in the transpose read's slot order
                acc = mma(pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]), rows_frag_tr(Ks, 32 * ks + 4 * gq, dt, fr), acc);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (4 * gq + rr < 13) rp[(t * 13 + 4 * gq + rr) * HD + 16 * dt + fr] = acc[rr];
        }
        if (tid < 169) {   // bias-table gradient: thread = table bin (dh, dw), sum of dS over the (query, key) pairs at that offset,
                           // read back from the dS image (query = key + 7(dh-6) + (dw-6)); no atomics
            const int da = tid / 13 - 6, db = tid % 13 - 6, dn = 7 * da + db;
            float acc = 0.f;
#pragma unroll
            for (int key = 0; key < 49; ++key) {
                const int ak = key / 7, bk = key % 7;
                const bool ok = (unsigned)(ak + da) < 7u && (unsigned)(bk + db) < 7u;
                const int nn = ok ? key + dn : 0;
                const float v = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(dSimg + nn * 128 + ((key * 2) ^ ((nn & 7) << 4))));
                acc += ok ? v : 0.f;
            }
            tab_part[((int64_t)bw * H + h) * 169 + tid] = acc / scale;   // (window, head, 169): 676 contiguous bytes per workgroup
                                                                       // (laid out like the parameter, (169, heads), it was 169 four-byte stores 64 B apart)
        }
    }
    // ================= phase B: wave = key tile; lane (key; 4 queries) -> dK_sel^T, dV_sel^T, scatter, coordinate gradients ==
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    {
        const int kt = wave;
        const int key = 16 * kt + fr, kc = key < 48 ? key : 48;
        uint4 pfb[2], dsfb[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {      // (key = 16 kt + fr; queries 32 kk + 4 gq .. + 3, + 16 ..): columns of the [query][key] images
            pfb[kk] = rows_frag_tr(Pimg, 32 * kk + 4 * gq, kt, fr);
            dsfb[kk] = rows_frag_tr(dSimg, 32 * kk + 4 * gq, kt, fr);
        }
        f32x4_t dks[4], dvs[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dks[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dvs[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint4 qtf = rows_frag_tr(Ks, 32 * kk + 4 * gq, dt, fr);      // (d = 16 dt + fr; queries 32 kk + 4 gq .. + 3, + 16 ..) of the Q | dO row images
                const uint4 dotf = rows_frag_tr(Vs, 32 * kk + 4 * gq, dt, fr);
                dks[dt] = mma(qtf, dsfb[kk], dks[dt]);     // lane (key = fr; d = 16dt + 4gq + r)
                dvs[dt] = mma(dotf, pfb[kk], dvs[dt]);
            }
        }
        __syncthreads();   // every wave holds its P / dS fragments: their images become the dK_sel | dV_sel rows
        // dK_sel / dV_sel rows (bf16, [key][16-B chunk] images) over P | dS: the coordinate gradients below read them chunk-wise next to the neighbour rows
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int o = swz(16 * kt + fr, 2 * dt + (gq >> 1)) + 8 * (gq & 1);
            *reinterpret_cast<uint2*>(Pimg + o) = make_uint2(pack_bf16x2(dks[dt][0], dks[dt][1]), pack_bf16x2(dks[dt][2], dks[dt][3]));
            *reinterpret_cast<uint2*>(dSimg + o) = make_uint2(pack_bf16x2(dvs[dt][0], dvs[dt][1]), pack_bf16x2(dvs[dt][2], dvs[dt][3]));
        }
    }
    __syncthreads();   // dK_sel / dV_sel rows complete
    {   // dK_sel | dV_sel (49 x 64 bf16 each) of this (image, window, head) -> the scratch buffer, row-major: 98 rows x 8 chunks of 16 B (rvsa_scatter_gemm_kernel sums them per token)
        bf16_t* out = reinterpret_cast<bf16_t*>(dkv) + (int64_t)blockIdx.x * (2 * 49 * HD);
        for (int idx = tid; idx < 2 * 49 * 8; idx += 256) {
            const int m = idx / (49 * 8), rem = idx - m * (49 * 8), key = rem >> 3, ch = rem & 7;
            *reinterpret_cast<uint4*>(out + (m * 49 + key) * HD + 8 * ch) = *reinterpret_cast<const uint4*>((m ? dSimg : Pimg) + swz(key, ch));
        }
    }
    {   // ---- coordinate gradients: lane = (key of a group of 8, 16-B chunk): d(K_sel, V_sel)/d(ix, iy) needs the four neighbour rows
        // (all neighbour loads of the wave's two key groups issued before the first use, as in the gather)
        const int kl = lane >> 3, ch = lane & 7;
        const uint32_t ch16 = (uint32_t)(16 * ch);
        uint4 kq[2][4], vq[2][4];
        bool live[2][4];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int key = (wave + 4 * gi) * 8 + kl;      // keys 0 .. 63 (49 real); the neighbour tokens come out of the gather's records
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int tok = __float_as_int(smp[(10 + k) * 64 + key]);
                const int tc = tok >= 0 ? tok : 0;
                live[gi][k] = tok >= 0;
                const uint32_t roff = (uint32_t)tc * ld2 + ch16;
                kq[gi][k] = ldg16_at(base + C, roff);
                vq[gi][k] = ldg16_at(base + 2 * C, roff);
            }
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int key = (wave + 4 * gi) * 8 + kl, kc = key < 48 ? key : 48;
            const float fx = smp[0 * 64 + kc], fy = smp[1 * 64 + kc];
            // dK_sel / dV_sel chunk (8 bf16 each, as stored) . neighbour row chunk with v_dot2c_f32_bf16: two exact bf16 products + f32 accumulate per instruction
            // (round 6: unpacked to f32 it was 16 conversions + 16 multiply-adds per neighbour)
            const uint4 dkq = *reinterpret_cast<const uint4*>(Pimg + swz(key, ch)), dvq = *reinterpret_cast<const uint4*>(dSimg + swz(key, ch));
            const uint32_t dkw[4] = {dkq.x, dkq.y, dkq.z, dkq.w}, dvw[4] = {dvq.x, dvq.y, dvq.z, dvq.w};
            float dix = 0.f, diy = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t kw[4] = {kq[gi][k].x, kq[gi][k].y, kq[gi][k].z, kq[gi][k].w}, vw[4] = {vq[gi][k].x, vq[gi][k].y, vq[gi][k].z, vq[gi][k].w};
                float dot = 0.f, dot2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dot = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, dkw[e]), __builtin_bit_cast(bf16x2v_t, kw[e]), dot, false);
                    dot2 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v_t, dvw[e]), __builtin_bit_cast(bf16x2v_t, vw[e]), dot2, false);
                }
                dot += dot2;
                dot = live[gi][k] ? dot : 0.f;
                dot += lane_xor<1>(dot);
                dot += lane_xor<2>(dot);
                dot += lane_xor<4>(dot);
                const int dx = k & 1, dy = k >> 1;
                dix += dot * (dy ? fy : 1.0f - fy) * (dx ? 1.0f : -1.0f);
                diy += dot * (dx ? fx : 1.0f - fx) * (dy ? 1.0f : -1.0f);
            }
            if (ch == 0 && key < 49) {
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
    if (lane == 0) {   // per-wave partials, summed in a fixed order: LDS atomics here made dsamp differ by an ulp from run to run,
                       // and downstream bf16 roundings turned that into 1e-4 gradient differences (tools/probes/race_finder.py)
        vsum[8 * wave + 0] = v0; vsum[8 * wave + 1] = v1; vsum[8 * wave + 2] = v2; vsum[8 * wave + 3] = v3; vsum[8 * wave + 4] = v4;
    }
    __syncthreads();
    if (tid < 5) {
        float* dp = dsamp + (int64_t)bw * 5 * H;
        const float sum = (vsum[tid] + vsum[8 + tid]) + (vsum[16 + tid] + vsum[24 + tid]);
        dp[tid < 2 ? 2 * h + tid : tid < 4 ? 2 * H + 2 * h + (tid - 2) : 4 * H + h] = sum;
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 3) void rvsa_bwd4_mfma_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ samp, const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout,
                                                            const float* __restrict__ lse, bf16_t* __restrict__ dqkv, float* __restrict__ dkv, float* __restrict__ dsamp,
                                                            float* __restrict__ rel_part, float* __restrict__ tab_part,
                                                            const float* __restrict__ rel_h, const float* __restrict__ rel_w, const float* __restrict__ bias_table,
                                                            RvsaGeom g, float scale) {
    constexpr int dense_scatter = MODE;
    __shared__ __attribute__((aligned(16))) char Ks[64 * 128];
    __shared__ __attribute__((aligned(16))) char Vs[64 * 128];
    __shared__ __attribute__((aligned(16))) char R2[2 * 64 * TP];
    __shared__ float QR[26 * 64];
    __shared__ float dQR[26 * 64];
    __shared__ float tab[176];
    __shared__ float lses[64];
    __shared__ float delta[64];
    __shared__ __attribute__((aligned(16))) float smp[SMP_F * 64];
    __shared__ float vsum[4 * 8];
    char* Kt = R2;
    char* Qt = R2;
    char* dOt = R2 + 64 * TP;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, gq = lane >> 4;
    const int H = g.heads, nW = g.nh * g.nw;
    const int h = blockIdx.x % H, bw = blockIdx.x / H, b = bw / nW, win = bw % nW, wi = win / g.nw, wj = win % g.nw;
    const int C = H * HD, N = g.Hp * g.Wp;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;
    const bf16_t* dob = dout + (int64_t)b * N * C + h * HD;
    const uint32_t ld2 = (uint32_t)(6 * C);      // bytes per qkv row

    if (tid < 176) {
        tab[tid] = tid < 169 ? bias_table[tid * H + h] : 0.f;
    }
    for (int i = tid; i < 26 * 64; i += 256) dQR[i] = 0.f;
    // ---- this wave's query tile: Q / dO fragments and QR = tables x Q^T  (first: these loads depend on nothing, so they are in
    // flight together with the gather's)
    const int qt = wave;
    const int nA = 16 * qt + fr;
    const int qtokA = nA < 49 ? query_token(g, nA, wi, wj) : -1;
    uint4 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = row_frag(base, ld, qtokA, ks * 32 + gq * 8);
        dof[ks] = row_frag(dob, C, qtokA, ks * 32 + gq * 8);
    }
    // ---- gather: lane = (key of a group of 8, 16-B chunk of the 64-channel row) -- one wave instruction reads 8 WHOLE 128-B rows --
    // and delta = dO . O per query in the same lane arrangement.  Every lane computes the sample position of its own two keys (8
    // lanes share one; no wave-0-only phase, no barrier before the gather) and the chunk-0 lanes publish it in `smp` for the later
    // phases.  ALL loads of the phase (2 key groups x 4 neighbours x {K, V} + 2 query groups x {dO, O}) are issued before the
    // first one is used: the phase is a chain of HBM round trips (the qkv rows of a window are cold), and issued group by group
    // they cost one latency each (phase timing, MTP_RVSA_STOP).
    {
        const int kl = lane >> 3, ch = lane & 7;
        const uint32_t ch16 = (uint32_t)(16 * ch);
        uint4 kq[2][4], vq[2][4], da[2], oc[2];
        float wq[2][4];
        int qtok[2];
        {   // ---- ONE sample computation per key (round 5): lane l < 16 of a wave owns key (wave + 4 (l >> 3)) * 8 + (l & 7) -- the 16 keys whose rows this
            // wave gathers -- and publishes the record (position, the four neighbour tokens and weights) in `smp`; the other lanes repeat it idly (same
            // instruction stream).  The wave reads its OWN records back (LDS operations of one wave execute in order: no block barrier), so the gather
            // costs one sample + four neighbours per wave instruction stream instead of two + eight, and the coordinate-gradient phase recomputes nothing.
            const int l16 = lane & 15;
            const int key = (wave + 4 * (l16 >> 3)) * 8 + (l16 & 7), kc = key < 48 ? key : 48;
            Sample sm = make_sample(g, samp + (int64_t)bw * 5 * H, h, wi, wj, kc / 7, kc % 7);   // (unconditional: no branch around its loads)
            if (key >= 49) { sm.x0 = -100; sm.y0 = -100; sm.fx = 0.f; sm.fy = 0.f; }      // keys >= 49 are zero rows: every neighbour outside -> weight 0
            float wv[4];
            int tk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float w;
                const int tok = neighbour(g, sm.x0, sm.y0, sm.fx, sm.fy, k, w);
                tk[k] = tok;
                wv[k] = tok >= 0 ? w : 0.f;
            }
            if (lane < 16) {
                smp[0 * 64 + key] = sm.fx; smp[1 * 64 + key] = sm.fy; smp[2 * 64 + key] = __int_as_float(sm.x0); smp[3 * 64 + key] = __int_as_float(sm.y0);
                smp[4 * 64 + key] = sm.rx; smp[5 * 64 + key] = sm.ry; smp[6 * 64 + key] = sm.cs; smp[7 * 64 + key] = sm.sn;
                smp[8 * 64 + key] = sm.relx; smp[9 * 64 + key] = sm.rely;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    smp[(10 + k) * 64 + key] = __int_as_float(tk[k]);
                    smp[(14 + k) * 64 + key] = wv[k];
                }
            }
        }
        // (the records cross lanes: a wavefront-scope release + wave barrier keeps the compiler from moving the read-back above the writes; the hardware
        //  executes one wave's LDS operations in order, so this costs no instruction -- ADVICE r05)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int key = (wave + 4 * gi) * 8 + kl;      // 8 groups = 64 key rows
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int tok = __float_as_int(smp[(10 + k) * 64 + key]);
                const int tc = tok >= 0 ? tok : 0;
                wq[gi][k] = smp[(14 + k) * 64 + key];
                const uint32_t roff = (uint32_t)tc * ld2 + ch16;      // 32-bit byte offsets off the (image, head) base: no 64-bit address arithmetic
                kq[gi][k] = ldg16_at(base + C, roff);
                vq[gi][k] = ldg16_at(base + 2 * C, roff);
            }
            const int n = (wave + 4 * gi) * 8 + kl;
            qtok[gi] = n < 49 ? query_token(g, n, wi, wj) : -1;
            const int tc = qtok[gi] >= 0 ? qtok[gi] : 0;
            da[gi] = ldg16_at(dob, (uint32_t)tc * (uint32_t)(2 * C) + ch16);
            oc[gi] = ldg16_at(o + (int64_t)b * N * C + h * HD, (uint32_t)tc * (uint32_t)(2 * C) + ch16);
        }
        // QR = tables x Q^T of this wave's query tile, while the gather's loads are in flight
    #pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float* tb = t ? rel_w : rel_h;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            acc = mma(table_frag(tb, fr, 13, gq * 8), qf[0], acc);
            acc = mma(table_frag(tb, fr, 13, 32 + gq * 8), qf[1], acc);
    #pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (4 * gq + rr < 13) QR[(t * 13 + 4 * gq + rr) * 64 + nA] = acc[rr];
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int key = (wave + 4 * gi) * 8 + kl;
            float ks[8], vs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { ks[e] = 0.f; vs[e] = 0.f; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t kw[4] = {kq[gi][k].x, kq[gi][k].y, kq[gi][k].z, kq[gi][k].w}, vw[4] = {vq[gi][k].x, vq[gi][k].y, vq[gi][k].z, vq[gi][k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ks[2 * e] += wq[gi][k] * bf16_bits_to_f32(kw[e] & 0xffffu); ks[2 * e + 1] += wq[gi][k] * bf16_bits_to_f32(kw[e] >> 16);
                    vs[2 * e] += wq[gi][k] * bf16_bits_to_f32(vw[e] & 0xffffu); vs[2 * e + 1] += wq[gi][k] * bf16_bits_to_f32(vw[e] >> 16);
                }
            }
            *reinterpret_cast<uint4*>(Ks + swz(key, ch)) = pack_bf16x8(ks[0], ks[1], ks[2], ks[3], ks[4], ks[5], ks[6], ks[7]);
            *reinterpret_cast<uint4*>(Vs + swz(key, ch)) = pack_bf16x8(vs[0], vs[1], vs[2], vs[3], vs[4], vs[5], vs[6], vs[7]);
#pragma unroll
            for (int e = 0; e < 8; ++e) *reinterpret_cast<uint16_t*>(Kt + (8 * ch + e) * TP + key * 2) = (uint16_t)f32_to_bf16_bits(ks[e]);
            // delta = dO . O of query n = key index, lse
            const uint32_t aw[4] = {da[gi].x, da[gi].y, da[gi].z, da[gi].w}, cw[4] = {oc[gi].x, oc[gi].y, oc[gi].z, oc[gi].w};
            float dl = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                dl += bf16_bits_to_f32(aw[e] & 0xffffu) * bf16_bits_to_f32(cw[e] & 0xffffu) + bf16_bits_to_f32(aw[e] >> 16) * bf16_bits_to_f32(cw[e] >> 16);
            dl += lane_xor<1>(dl);
            dl += lane_xor<2>(dl);
            dl += lane_xor<4>(dl);
            if (ch == 0) {
                delta[key] = qtok[gi] >= 0 ? dl : 0.f;
                lses[key] = key < 49 ? lse[(int64_t)blockIdx.x * 49 + key] : 0.f;
            }
        }
    }
    __syncthreads();

    // ================= phase A: wave = query tile; lane (query; 4 keys) -> dQ, dQR, P^T / dS^T images ===============
    {
        uint4 kf[4][2], vf[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kf[kt][ks] = ld16(Ks + swz(16 * kt + fr, ks * 4 + gq));
                vf[kt][ks] = ld16(Vs + swz(16 * kt + fr, ks * 4 + gq));
            }
        __syncthreads();   // every wave holds its K_sel / V_sel fragments: Ks / Vs may now be overwritten with P^T / dS^T
        const int n = nA, nq = n < 48 ? n : 48;
        const int aq = (nq * 37) >> 8, bq = nq - 7 * aq;
        const float ls = lses[n], dl = delta[n];
        f32x4_t sT[4], dpT[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            sT[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dpT[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                sT[kt] = mma(kf[kt][ks], qf[ks], sT[kt]);
                dpT[kt] = mma(vf[kt][ks], dof[ks], dpT[kt]);
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
                p = (key < 49 && n < 49) ? p : 0.f;
                const float ds = p * (dpT[kt][r] - dl);
                sT[kt][r] = ds * scale;
                const int off = key * 128 + ((n * 2) ^ ((key & 7) << 4));
                *reinterpret_cast<uint16_t*>(Ks + off) = (uint16_t)f32_to_bf16_bits(p);
                *reinterpret_cast<uint16_t*>(Vs + off) = (uint16_t)f32_to_bf16_bits(ds * scale);
            }
        uint4 dsf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            dsf[kk] = pack_bf16x8(sT[2 * kk][0], sT[2 * kk][1], sT[2 * kk][2], sT[2 * kk][3], sT[2 * kk + 1][0], sT[2 * kk + 1][1], sT[2 * kk + 1][2], sT[2 * kk + 1][3]);
        {   // d(q.Rh)[q][aq - ak + 6] = sum over the 7 keys of window row ak of dS (same for columns): segmented sums done as
            // E[a][key] (0/1 indicator, MFMA A layout) x dS^T on the matrix cores -- LDS float atomics cost ~190 LDS cycles per
            // instruction here and were 40 % of this kernel's wave time (SQ_WAIT_INST_LDS).
            f32x4_t dqh = {0.f, 0.f, 0.f, 0.f}, dqw = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                uint32_t wh[4] = {0u, 0u, 0u, 0u}, ww[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = 32 * kk + 4 * gq + (j & 3) + (j >> 2) * 16;
                    const int ak = (key * 37) >> 8, bk = key - 7 * ak;
                    const uint32_t one = 0x3f80u << ((j & 1) * 16);
                    wh[j >> 1] |= (key < 49 && ak == fr) ? one : 0u;
                    ww[j >> 1] |= (key < 49 && bk == fr) ? one : 0u;
                }
                dqh = mma(make_uint4(wh[0], wh[1], wh[2], wh[3]), dsf[kk], dqh);
                dqw = mma(make_uint4(ww[0], ww[1], ww[2], ww[3]), dsf[kk], dqw);
            }
            const float inv_scale = 1.0f / scale;   // dsf carries dS * scale
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 4 * gq + r;
                if (a < 7) {
                    dQR[(aq - a + 6) * 64 + n] = dqh[r] * inv_scale;
                    dQR[(13 + bq - a + 6) * 64 + n] = dqw[r] * inv_scale;
                }
            }
        }
        __syncthreads();   // dQR / P^T / dS^T complete
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
            acc = mma(table_frag_t(rel_h, 16 * dt + fr, 13, 8 * gq), eh, acc);
            acc = mma(table_frag_t(rel_w, 16 * dt + fr, 13, 8 * gq), ew, acc);
            if (qtokA >= 0) store4(dqkv + ((int64_t)b * N + qtokA) * ld + h * HD + 16 * dt + 4 * gq, make_float4(acc[0], acc[1], acc[2], acc[3]));
        }
    }
    __syncthreads();   // K^T no longer needed: R2 becomes Q^T | dO^T
    {   // ---- thread = (query = lane, 16-channel quarter = wave)
        const int tok = lane < 49 ? query_token(g, lane, wi, wj) : -1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int d0 = 16 * wave + 8 * i;
            const uint4 rq = row_frag(base, ld, tok, d0), rd = row_frag(dob, C, tok, d0);
            const uint32_t wq[4] = {rq.x, rq.y, rq.z, rq.w}, wd[4] = {rd.x, rd.y, rd.z, rd.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<uint16_t*>(Qt + (d0 + 2 * e) * TP + lane * 2) = (uint16_t)(wq[e] & 0xffffu);
                *reinterpret_cast<uint16_t*>(Qt + (d0 + 2 * e + 1) * TP + lane * 2) = (uint16_t)(wq[e] >> 16);
                *reinterpret_cast<uint16_t*>(dOt + (d0 + 2 * e) * TP + lane * 2) = (uint16_t)(wd[e] & 0xffffu);
                *reinterpret_cast<uint16_t*>(dOt + (d0 + 2 * e + 1) * TP + lane * 2) = (uint16_t)(wd[e] >> 16);
            }
        }
    }
    __syncthreads();
    {   // ---- table gradients: wave = d tile
        float* rp = rel_part + (int64_t)blockIdx.x * 26 * HD;
        const int dt = wave;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            const char* row = Qt + (16 * dt + fr) * TP;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) v[x] = fr < 13 ? dQR[(t * 13 + fr) * 64 + 32 * ks + 8 * gq + x] : 0.f;
                acc = mma(pack_bf16x8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]), ld8x2(row + (32 * ks + 8 * gq) * 2, row + (32 * ks + 8 * gq + 4) * 2), acc);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (4 * gq + rr < 13) rp[(t * 13 + 4 * gq + rr) * HD + 16 * dt + fr] = acc[rr];
        }
        if (tid < 169) {   // bias-table gradient: thread = table bin (dh, dw), sum of dS over the (query, key) pairs at that offset,
                           // read back from the dS^T image (query = key + 7(dh-6) + (dw-6)); no atomics
            const int da = tid / 13 - 6, db = tid % 13 - 6, dn = 7 * da + db;
            float acc = 0.f;
#pragma unroll
            for (int key = 0; key < 49; ++key) {
                const int ak = key / 7, bk = key % 7;
                const bool ok = (unsigned)(ak + da) < 7u && (unsigned)(bk + db) < 7u;
                const int nn = ok ? key + dn : 0;
                const float v = bf16_bits_to_f32(*reinterpret_cast<const uint16_t*>(Vs + key * 128 + ((nn * 2) ^ ((key & 7) << 4))));
                acc += ok ? v : 0.f;
            }
            tab_part[((int64_t)bw * H + h) * 169 + tid] = acc / scale;   // (window, head, 169): 676 contiguous bytes per workgroup
                                                                       // (laid out like the parameter, (169, heads), it was 169 four-byte stores 64 B apart)
        }
    }
    // ================= phase B: wave = key tile; lane (key; 4 queries) -> dK_sel^T, dV_sel^T, scatter, coordinate gradients ==
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
    {
        const int kt = wave;
        const int key = 16 * kt + fr, kc = key < 48 ? key : 48;
        uint4 pfb[2], dsfb[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int o0 = key * 128 + (((32 * kk + 4 * gq) * 2) ^ ((key & 7) << 4));
            const int o1 = key * 128 + (((32 * kk + 16 + 4 * gq) * 2) ^ ((key & 7) << 4));
            pfb[kk] = ld8x2(Ks + o0, Ks + o1);
            dsfb[kk] = ld8x2(Vs + o0, Vs + o1);
        }
        f32x4_t dks[4], dvs[4], dk2[4], dv2[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dks[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dvs[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dk2[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv2[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const char* rq = Qt + (16 * dt + fr) * TP;
            const char* rd = dOt + (16 * dt + fr) * TP;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const uint4 qtf = ld8x2(rq + (32 * kk + 4 * gq) * 2, rq + (32 * kk + 16 + 4 * gq) * 2);
                const uint4 dotf = ld8x2(rd + (32 * kk + 4 * gq) * 2, rd + (32 * kk + 16 + 4 * gq) * 2);
                dks[dt] = mma(qtf, dsfb[kk], dks[dt]);     // lane (key = fr; d = 16dt + 4gq + r)   -> coordinate gradients
                dvs[dt] = mma(dotf, pfb[kk], dvs[dt]);
                dk2[dt] = mma(dsfb[kk], qtf, dk2[dt]);     // lane (d = 16dt + fr; key = 4gq + r)   -> coalesced scatter
                dv2[dt] = mma(pfb[kk], dotf, dv2[dt]);
            }
        }
        __syncthreads();   // every wave holds its P^T / dS^T fragments and is past its last read of Q^T | dO^T
        // dK_sel / dV_sel rows (bf16, the [key][16-B chunk] image of K_sel / V_sel) over P^T | dS^T: the coordinate gradients
        // below read them chunk-wise next to the neighbour rows
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int o = swz(16 * kt + fr, 2 * dt + (gq >> 1)) + 8 * (gq & 1);
            *reinterpret_cast<uint2*>(Ks + o) = make_uint2(pack_bf16x2(dks[dt][0], dks[dt][1]), pack_bf16x2(dks[dt][2], dks[dt][3]));
            *reinterpret_cast<uint2*>(Vs + o) = make_uint2(pack_bf16x2(dvs[dt][0], dvs[dt][1]), pack_bf16x2(dvs[dt][2], dvs[dt][3]));
        }
        if constexpr (dense_scatter == 4) {
            // "gemm" scatter: the rows just written to Ks / Vs leave the kernel (below); rvsa_scatter_gemm_kernel sums them per token
        } else {
            // dK_sel^T / dV_sel^T -> bf16 [d][key] images over Q^T | dO^T; the scatter itself runs after the coordinate
            // gradients, see the end of the kernel
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int o = (16 * dt + fr) * TP + (16 * kt + 4 * gq) * 2;
                *reinterpret_cast<uint2*>(R2 + o) = make_uint2(pack_bf16x2(dk2[dt][0], dk2[dt][1]), pack_bf16x2(dk2[dt][2], dk2[dt][3]));
                *reinterpret_cast<uint2*>(R2 + 64 * TP + o) = make_uint2(pack_bf16x2(dv2[dt][0], dv2[dt][1]), pack_bf16x2(dv2[dt][2], dv2[dt][3]));
            }
        }
    }
    __syncthreads();   // dK_sel / dV_sel rows complete
    if constexpr (dense_scatter == 4) {
        // dK_sel | dV_sel (49 x 64 bf16 each) of this (image, window, head) -> the scratch buffer, row-major: 98 rows x 8 chunks of 16 B
        bf16_t* out = reinterpret_cast<bf16_t*>(dkv) + (int64_t)blockIdx.x * (2 * 49 * HD);
        for (int idx = tid; idx < 2 * 49 * 8; idx += 256) {
            const int m = idx / (49 * 8), rem = idx - m * (49 * 8), key = rem >> 3, ch = rem & 7;
            *reinterpret_cast<uint4*>(out + (m * 49 + key) * HD + 8 * ch) = *reinterpret_cast<const uint4*>((m ? Vs : Ks) + swz(key, ch));
        }
    }
    {   // ---- coordinate gradients: lane = (key of a group of 8, 16-B chunk): d(K_sel, V_sel)/d(ix, iy) needs the four neighbour rows
        // (all neighbour loads of the wave's two key groups issued before the first use, as in the gather)
        const int kl = lane >> 3, ch = lane & 7;
        const uint32_t ch16 = (uint32_t)(16 * ch);
        uint4 kq[2][4], vq[2][4];
        bool live[2][4];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int key = (wave + 4 * gi) * 8 + kl;      // keys 0 .. 63 (49 real); the neighbour tokens come out of the gather's records
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int tok = __float_as_int(smp[(10 + k) * 64 + key]);
                const int tc = tok >= 0 ? tok : 0;
                live[gi][k] = tok >= 0;
                const uint32_t roff = (uint32_t)tc * ld2 + ch16;
                kq[gi][k] = ldg16_at(base + C, roff);
                vq[gi][k] = ldg16_at(base + 2 * C, roff);
            }
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int key = (wave + 4 * gi) * 8 + kl, kc = key < 48 ? key : 48;
            const float fx = smp[0 * 64 + kc], fy = smp[1 * 64 + kc];
            float dk[8], dv[8];
            load8(reinterpret_cast<const bf16_t*>(Ks + swz(key, ch)), dk);
            load8(reinterpret_cast<const bf16_t*>(Vs + swz(key, ch)), dv);
            float dix = 0.f, diy = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t kw[4] = {kq[gi][k].x, kq[gi][k].y, kq[gi][k].z, kq[gi][k].w}, vw[4] = {vq[gi][k].x, vq[gi][k].y, vq[gi][k].z, vq[gi][k].w};
                float dot = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dot += dk[2 * e] * bf16_bits_to_f32(kw[e] & 0xffffu) + dk[2 * e + 1] * bf16_bits_to_f32(kw[e] >> 16)
                         + dv[2 * e] * bf16_bits_to_f32(vw[e] & 0xffffu) + dv[2 * e + 1] * bf16_bits_to_f32(vw[e] >> 16);
                dot = live[gi][k] ? dot : 0.f;
                dot += lane_xor<1>(dot);
                dot += lane_xor<2>(dot);
                dot += lane_xor<4>(dot);
                const int dx = k & 1, dy = k >> 1;
                dix += dot * (dy ? fy : 1.0f - fy) * (dx ? 1.0f : -1.0f);
                diy += dot * (dx ? fx : 1.0f - fx) * (dy ? 1.0f : -1.0f);
            }
            if (ch == 0 && key < 49) {
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
    if (lane == 0) {   // per-wave partials, summed in a fixed order: LDS atomics here made dsamp differ by an ulp from run to run,
                       // and downstream bf16 roundings turned that into 1e-4 gradient differences (tools/probes/race_finder.py)
        vsum[8 * wave + 0] = v0; vsum[8 * wave + 1] = v1; vsum[8 * wave + 2] = v2; vsum[8 * wave + 3] = v3; vsum[8 * wave + 4] = v4;
    }
    __syncthreads();
    if (tid < 5) {
        float* dp = dsamp + (int64_t)bw * 5 * H;
        const float sum = (vsum[tid] + vsum[8 + tid]) + (vsum[16 + tid] + vsum[24 + tid]);
        dp[tid < 2 ? 2 * h + tid : tid < 4 ? 2 * H + 2 * h + (tid - 2) : 4 * H + h] = sum;
    }
    if constexpr (dense_scatter != 4) {
        // ================= scatter of dK_sel / dV_sel through the bilinear weights, as a product on the matrix cores =========
        // dK[token][d] += sum_key W[token][key] dK_sel[key][d],  W = hat(ix_key - X_token) hat(iy_key - Y_token) -- the same four
        // corner weights, summed per TOKEN before they leave the workgroup.  The memory side retires ~31 G 64-byte f32 atomics/s
        // (measured) and the per-(key, corner) scatter issued 49 x 4 x 8 of them per workgroup: it was the
        // kernel's floor.  Neighbouring keys share corners, so per token it is ~81 x 8 for near-identity sampling.
        // (Tried in round 2: packed-bf16 atomics, global_atomic_pk_add_bf16, straight into dqkv -- half the requests, no f32 scratch, no
        //  conversion pass -- measured SLOWER end to end: 287 vs 276 us per call.)
        // wave = token tiles w, w+4, ... of 16 tokens inside the row range the samples can touch.
        const char* dKt = R2;
        const char* dVt = R2 + 64 * TP;
        float kx[16], ky[16];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e4 = 0; e4 < 2; ++e4) {
                const int k0 = 32 * kk + 8 * gq + 4 * e4;
                const float4 fxv = *reinterpret_cast<const float4*>(smp + 0 * 64 + k0), fyv = *reinterpret_cast<const float4*>(smp + 1 * 64 + k0);
                const float4 x0v = *reinterpret_cast<const float4*>(smp + 2 * 64 + k0), y0v = *reinterpret_cast<const float4*>(smp + 3 * 64 + k0);
                const int i = 8 * kk + 4 * e4;
                kx[i] = (float)__float_as_int(x0v.x) + fxv.x; kx[i + 1] = (float)__float_as_int(x0v.y) + fxv.y;
                kx[i + 2] = (float)__float_as_int(x0v.z) + fxv.z; kx[i + 3] = (float)__float_as_int(x0v.w) + fxv.w;
                ky[i] = (float)__float_as_int(y0v.x) + fyv.x; ky[i + 1] = (float)__float_as_int(y0v.y) + fyv.y;
                ky[i + 2] = (float)__float_as_int(y0v.z) + fyv.z; ky[i + 3] = (float)__float_as_int(y0v.w) + fyv.w;
            }
        const float iy = (float)__float_as_int(smp[3 * 64 + lane]) + smp[1 * 64 + lane];   // clamped to [-4, He + 4] by make_sample
        const float ymin = -wave_max(lane < 49 ? -iy : -1e30f), ymax = wave_max(lane < 49 ? iy : -1e30f);
        const int ty_lo = min(max((int)floorf(ymin) - g.pad_t, 0), g.Hp - 1), ty_hi = min(max((int)floorf(ymax) + 1 - g.pad_t, 0), g.Hp - 1);
        const int t_lo = __builtin_amdgcn_readfirstlane((ty_lo * g.Wp) >> 4), t_hi = __builtin_amdgcn_readfirstlane(((ty_hi + 1) * g.Wp - 1) >> 4);
        uint4 bk[4][2], bv[4][2];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int o = (16 * dt + fr) * TP + (32 * kk + 8 * gq) * 2;
                bk[dt][kk] = ld8x2(dKt + o, dKt + o + 8);
                bv[dt][kk] = ld8x2(dVt + o, dVt + o + 8);
            }
        for (int ti = t_lo + wave; ti <= t_hi; ti += 4) {
            const int tok = 16 * ti + fr, tyy = tok / g.Wp, txx = tok - tyy * g.Wp;
            const float X = (float)(txx + g.pad_l), Y = (float)(tyy + g.pad_t);
            const float live = tok < N ? 1.f : 0.f;
            uint4 af[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                float w[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    w[e] = live * fmaxf(0.f, 1.f - fabsf(kx[8 * kk + e] - X)) * fmaxf(0.f, 1.f - fabsf(ky[8 * kk + e] - Y));
                af[kk] = pack_bf16x8(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                f32x4_t ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    ak = mma(af[kk], bk[dt][kk], ak);     // lane (token = 16 ti + 4 gq + r; d = 16 dt + fr)
                    av = mma(af[kk], bv[dt][kk], av);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t2 = 16 * ti + 4 * gq + r;
                    if (t2 < N) {
                        float* drow = dkv + ((int64_t)b * N + t2) * (2 * C) + h * HD + 16 * dt + fr;
                        if (ak[r] != 0.f) atomicAdd(drow, ak[r]);
                        if (av[r] != 0.f) atomicAdd(drow + C, av[r]);
                    }
                }
            }
        }
    }
}

// ===================================================================================================================
// "gemm" scatter (round 2): dK[token] = sum over ALL samples of the image (windows x 49 keys) of
// W[token][sample] dK_sel[sample],  W = hat(ix_s - X_t) hat(iy_s - Y_t) -- the bilinear scatter of a whole (image, head) as ONE dense
// product on the matrix cores, every token row written exactly once: no atomics, no f32 scratch, no clearing pass, no conversion
// pass (those were 26 + ~50 + 24 us per block at ViT-L, B = 64).  The backward kernel above leaves dK_sel / dV_sel (bf16 rows) in the
// scratch buffer; one workgroup per (image, head) stages them row-major in LDS (K^T-style fragments come out of
// ds_read_b64_tr_b16), recomputes the sample positions from the five sampling scalars and builds the W fragments in registers.
// LDS: Kimg | Vimg (224 x 128 B, swizzled) | xs | ys
// ===================================================================================================================
constexpr int SCB = 224;      // tokens per workgroup (band) = samples per staged chunk: 7 token tiles of 32 / 7 k-steps of 32

// grid (B * heads, bands of SCB tokens).  The samples of the (image, head) are staged SCB at a time; a chunk none of whose samples can
// touch the band's token rows is skipped before its rows are loaded (for near-identity sampling a band sees the windows of its own rows
// only, so larger grids cost about one chunk per band, not windows x bands).  <= 224 tokens (the 14 x 14 grid): one band, one chunk.
__global__ __launch_bounds__(256, 2) void rvsa_scatter_gemm_kernel(const bf16_t* __restrict__ dsel, const float* __restrict__ samp, bf16_t* __restrict__ dqkv,
                                                                  RvsaGeom g) {
    __shared__ __attribute__((aligned(16))) char Kimg[SCB * 128];
    __shared__ __attribute__((aligned(16))) char Vimg[SCB * 128];
    __shared__ __attribute__((aligned(16))) float xs[SCB];
    __shared__ __attribute__((aligned(16))) float ys[SCB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, gq = lane >> 4;
    const int H = g.heads, nW = g.nh * g.nw, S = nW * 49;
    const int h = blockIdx.x % H, b = blockIdx.x / H;
    const int C = H * HD, N = g.Hp * g.Wp;
    const int t0 = blockIdx.y * SCB, nt = (N - t0) < SCB ? (N - t0) : SCB;      // this band: tokens [t0, t0 + nt)
    const int64_t ld = 3 * (int64_t)C;
    // wave w owns token tiles w, w + 4, w + 8, w + 12 of the band (14 tiles): the K / V fragments and the sample coordinates of a
    // 32-sample step are read from LDS ONCE and used for all of them (with the tile loop outside the kernel was LDS-issue bound)
    float X[4], Y[4], live[4];
    f32x4_t dk[4][4], dv[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tl = 16 * (wave + 4 * i) + fr;
        const int tc = t0 + (tl < nt ? tl : nt - 1);
        const int ty = tc / g.Wp, tx = tc - ty * g.Wp;
        X[i] = (float)(tx + g.pad_l);
        Y[i] = (float)(ty + g.pad_t);
        live[i] = tl < nt ? 1.f : 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            dk[i][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            dv[i][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int ntile = (nt + 15) / 16;
    const float ylo = (float)(t0 / g.Wp + g.pad_t) - 1.0f, yhi = (float)((t0 + nt - 1) / g.Wp + g.pad_t) + 1.0f;   // a sample outside (ylo, yhi) has weight 0 on every token row of the band
    for (int c0 = 0; c0 < S; c0 += SCB) {
        __syncthreads();      // the previous chunk's fragments have been read
        int hit = 0;
        if (tid < SCB) {
            const int sg = c0 + tid;
            float x = -1.0e4f, y = -1.0e4f;
            if (sg < S) {
                const int w = sg / 49, k = sg - 49 * w;
                const Sample sa = make_sample(g, samp + (int64_t)(b * nW + w) * 5 * H, h, w / g.nw, w % g.nw, k / 7, k % 7);
                x = (float)sa.x0 + sa.fx;
                y = (float)sa.y0 + sa.fy;
            }
            xs[tid] = x;
            ys[tid] = y;
            hit = (y > ylo && y < yhi) ? 1 : 0;
        }
        if (!__syncthreads_or(hit)) continue;       // (uniform) nothing in this chunk reaches the band
        // staging in two batches of unconditional loads on clamped rows (round 4): written as load -> store per iteration it was seven
        // serialised global round trips per workgroup (a branch around a load makes hipcc wait for it inside the branch)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            constexpr int IT = 4;      // 2 x 4 x 256 >= SCB * 8 = 1792
            uint4 kv[IT], vv[IT];
#pragma unroll
            for (int i = 0; i < IT; ++i) {
                const int idx = tid + 256 * (IT * half + i), sl = idx >> 3, ch = idx & 7, sg = c0 + sl;
                const int sc = sg < S ? sg : S - 1;
                const int w = sc / 49, k = sc - 49 * w;
                const bf16_t* src = dsel + ((int64_t)(b * nW + w) * H + h) * (2 * 49 * HD) + k * HD + 8 * ch;
                kv[i] = ldg16(src);
                vv[i] = ldg16(src + 49 * HD);
            }
#pragma unroll
            for (int i = 0; i < IT; ++i) {
                const int idx = tid + 256 * (IT * half + i), sl = idx >> 3, ch = idx & 7, sg = c0 + sl;
                if (idx < SCB * 8) {
                    const bool ok = sg < S;
                    *reinterpret_cast<uint4*>(Kimg + swz(sl, ch)) = ok ? kv[i] : make_uint4(0u, 0u, 0u, 0u);
                    *reinterpret_cast<uint4*>(Vimg + swz(sl, ch)) = ok ? vv[i] : make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
        __syncthreads();
        for (int kk = 0; kk < SCB / 32; ++kk) {
            const int s0 = 32 * kk + 4 * gq;
            const float4 xa = *reinterpret_cast<const float4*>(xs + s0), xb = *reinterpret_cast<const float4*>(xs + s0 + 16);
            const float4 ya = *reinterpret_cast<const float4*>(ys + s0), yb = *reinterpret_cast<const float4*>(ys + s0 + 16);
            const float sx[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w}, sy[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
            uint4 kf[4], vf[4];
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                kf[dt] = rows_frag_tr(Kimg, s0, dt, fr);
                vf[dt] = rows_frag_tr(Vimg, s0, dt, fr);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (wave + 4 * i < ntile) {      // (wave-uniform)
                    float w[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = live[i] * fmaxf(0.f, 1.f - fabsf(sx[e] - X[i])) * fmaxf(0.f, 1.f - fabsf(sy[e] - Y[i]));
                    const uint4 wf = pack_bf16x8(w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        dk[i][dt] = mma(kf[dt], wf, dk[i][dt]);     // D[d = 16 dt + 4 gq + r][token fr]
                        dv[i][dt] = mma(vf[dt], wf, dv[i][dt]);
                    }
                }
            }
        }
    }
    // results -> bf16 rows in LDS (over the K / V images, which nobody reads any more) -> whole 128-byte rows to global: out of the MFMA
    // layout a lane holds 4 channels of one token, i.e. 8-byte stores 6 KiB apart (measured: 11.5 of the kernel's 40 us)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tl = 16 * (wave + 4 * i) + fr;
        if (tl < nt) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int o = tl * 128 + (((2 * dt + (gq >> 1)) ^ (tl & 7)) << 4) + (gq & 1) * 8;
                *reinterpret_cast<uint2*>(Kimg + o) = make_uint2(pack_bf16x2(dk[i][dt][0], dk[i][dt][1]), pack_bf16x2(dk[i][dt][2], dk[i][dt][3]));
                *reinterpret_cast<uint2*>(Vimg + o) = make_uint2(pack_bf16x2(dv[i][dt][0], dv[i][dt][1]), pack_bf16x2(dv[i][dt][2], dv[i][dt][3]));
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < nt * 16; idx += 256) {       // (token, K | V, 16-byte chunk)
        const int tl = idx >> 4, m = (idx >> 3) & 1, ch = idx & 7;
        const uint4 v = *reinterpret_cast<const uint4*>((m ? Vimg : Kimg) + swz(tl, ch));
        *reinterpret_cast<uint4*>(dqkv + ((int64_t)b * N + t0 + tl) * ld + (1 + m) * C + h * HD + 8 * ch) = v;
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

// 4 = the scatter runs as rvsa_scatter_gemm_kernel (the caller then skips the clearing and conversion passes of the f32 scratch),
// 1 = f32 atomics per token tile inside the backward kernel
int mtp_rvsa_bwd_mfma_scatter_mode(int64_t Hp, int64_t Wp, int64_t heads) {
    static const bool dense = []() { const char* e = getenv("MTP_RVSA_SCATTER"); return e && e[0] == 'd'; }();   // MTP_RVSA_SCATTER=dense: the atomic form everywhere (tests)
    if (dense) return 1;
    const RvsaGeom g = make_geom(Hp, Wp, heads);
    const int64_t N = Hp * Wp, nW = (int64_t)g.nh * g.nw;
    const bool fits = Hp <= 64 && Wp <= 64 && nW * (2 * 49 * HD * 2) <= N * 2 * HD * 4;   // (the rows live in the caller's f32 scratch)
    return fits ? 4 : 1;
}

int mtp_rvsa_bwd_mfma_launch(const void* qkv, const float* samp, const void* o, const void* dout, const float* lse, void* dqkv, float* dkv, float* dsamp,
                             float* rel_part, float* tab_part, const float* rel_h, const float* rel_w, const float* bias_table,
                             int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s) {
    const RvsaGeom g = make_geom(Hp, Wp, heads);
    const int mode = mtp_rvsa_bwd_mfma_scatter_mode(Hp, Wp, heads);
    const dim3 grid((unsigned)(B * g.nh * g.nw * heads));
    if (mode == 4) {
        static const bool old4 = []() { const char* e = getenv("MTP_RVSA_BWD"); return e && e[0] == '4'; }();      // MTP_RVSA_BWD=4: the kernel of rounds 2-5 (A/B, bit-identity test)
        if (old4)
            hipLaunchKernelGGL(rvsa_bwd4_mfma_kernel<4>, grid, dim3(256), 0, s, (const bf16_t*)qkv, samp, (const bf16_t*)o, (const bf16_t*)dout, lse,
                               (bf16_t*)dqkv, dkv, dsamp, rel_part, tab_part, rel_h, rel_w, bias_table, g, scale);
        else
            hipLaunchKernelGGL(rvsa_bwd5_mfma_kernel, grid, dim3(256), 0, s, (const bf16_t*)qkv, samp, (const bf16_t*)o, (const bf16_t*)dout, lse,
                               (bf16_t*)dqkv, dkv, dsamp, rel_part, tab_part, rel_h, rel_w, bias_table, g, scale);
        const int64_t N = Hp * Wp;
        hipLaunchKernelGGL(rvsa_scatter_gemm_kernel, dim3((unsigned)(B * heads), (unsigned)((N + SCB - 1) / SCB)), dim3(256), 0, s, (const bf16_t*)dkv, samp, (bf16_t*)dqkv, g);
    } else {
        hipLaunchKernelGGL(rvsa_bwd4_mfma_kernel<1>, grid, dim3(256), 0, s, (const bf16_t*)qkv, samp, (const bf16_t*)o, (const bf16_t*)dout, lse,
                           (bf16_t*)dqkv, dkv, dsamp, rel_part, tab_part, rel_h, rel_w, bias_table, g, scale);
    }
    return mtp_launch_status();
}
