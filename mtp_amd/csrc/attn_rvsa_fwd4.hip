// 4-wave variant of the RVSA forward (see attn_mfma.hip for the algorithm).
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
    s.cs = __cosf(ang);      // v_cos / v_sin (abs error ~1e-6 on |ang| < pi): shared with the backward kernel
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


typedef short tr4s_t __attribute__((ext_vector_type(4)));
// transposed fragment out of a row-major swizzled bf16 image (ds_read_b64_tr_b16): lane (fr, gq) gets column 16 dt + fr of rows row0 .. row0 + 3 and row0 + 16 .. + 19
__device__ __forceinline__ uint4 rows_frag_tr(const char* img, int row0, int dt, int fr) {
    const int c = 16 * dt + 4 * (fr & 3);
    const int ra = row0 + (fr >> 2), rb = ra + 16;
    const int oa = ra * 128 + (((c >> 3) ^ (ra & 7)) << 4) + (c & 7) * 2, ob = rb * 128 + (((c >> 3) ^ (rb & 7)) << 4) + (c & 7) * 2;
    const tr4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4s_t*)(img + oa));
    const tr4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4s_t*)(img + ob));
    const uint2 l = __builtin_bit_cast(uint2, lo), hh = __builtin_bit_cast(uint2, hi);
    return make_uint4(l.x, l.y, hh.x, hh.y);
}

// ===================================================================================================================
// RVSA forward, 4 waves per (image, window, head): the bilinear gather is split over 256 threads (key x 16-channel quarter)
// and wave w owns query tile w, so 24 waves share a CU (6 workgroups x 4) instead of 4-6 single-wave problems.
// ===================================================================================================================
__global__ __launch_bounds__(256, 4) void rvsa_fwd4_mfma_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ samp, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                            const float* __restrict__ rel_h, const float* __restrict__ rel_w, const float* __restrict__ bias_table,
                                                            RvsaGeom g, float scale) {
    __shared__ __attribute__((aligned(16))) char Ks[64 * 128];
    __shared__ __attribute__((aligned(16))) char Vs[64 * 128];      // V_sel rows like Ks (round 6; it was a [d][key] image written in 2-byte units)
    __shared__ float QR[26 * 64];
    __shared__ float tab[176];
    __shared__ float rec[8 * 64];      // per key: the four neighbour tokens and weights of its bilinear sample
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), fr = lane & 15, gq = lane >> 4;
    const int H = g.heads, nW = g.nh * g.nw;
    const int h = blockIdx.x % H, bw = blockIdx.x / H, b = bw / nW, win = bw % nW, wi = win / g.nw, wj = win % g.nw;
    const int C = H * HD, N = g.Hp * g.Wp;
    const int64_t ld = 3 * (int64_t)C;
    const bf16_t* base = qkv + (int64_t)b * N * ld + h * HD;

    if (tid < 169) tab[tid] = bias_table[tid * H + h];
    // ---- this wave's query tile first: these loads depend on nothing, so they fly under the sample computation and the gather
    const int qt = wave;
    const int n = 16 * qt + fr;
    const int qtok = n < 49 ? query_token(g, n, wi, wj) : -1;
    uint4 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = row_frag(base, ld, qtok, ks * 32 + gq * 8);
    {   // ---- gather: lane = (key of a group of 8, 16-B chunk of the 64-channel row): a wave instruction reads 8 WHOLE 128-B rows
        // (with lane = key it touched 49 cache lines for 16 B each); all 16 loads of the wave's two key groups are issued before
        // the first use.  Every lane computes the sample position of its own key (8 lanes share one: no barrier needed).
        const int kl = lane >> 3, ch = lane & 7;
        uint4 kq[2][4], vq[2][4];
        float wq[2][4];
        {   // ONE sample + four neighbours per key and wave instruction stream (round 5; it was two + eight: every lane its own two keys): lane l < 16 owns
            // key (wave + 4 (l >> 3)) * 8 + (l & 7), writes the neighbour record to `rec`, and the wave reads its own records back -- LDS operations of one
            // wave execute in order, so no barrier is involved (attn_rvsa_bwd4.hip does the same and keeps the records for its coordinate-gradient phase)
            const int l16 = lane & 15;
            const int key = (wave + 4 * (l16 >> 3)) * 8 + (l16 & 7), kc = key < 48 ? key : 48;
            const Sample sm = make_sample(g, samp + (int64_t)bw * 5 * H, h, wi, wj, kc / 7, kc % 7);
            float wv[4];
            int tk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float w;
                const int tok = key < 49 ? neighbour(g, sm.x0, sm.y0, sm.fx, sm.fy, k, w) : -1;      // keys >= 49 are zero rows
                tk[k] = tok;
                wv[k] = tok >= 0 ? w : 0.f;
            }
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    rec[k * 64 + key] = __int_as_float(tk[k]);
                    rec[(4 + k) * 64 + key] = wv[k];
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
                const int tok = __float_as_int(rec[k * 64 + key]);
                const int tc = tok >= 0 ? tok : 0;
                wq[gi][k] = rec[(4 + k) * 64 + key];
                const uint32_t roff = (uint32_t)tc * (uint32_t)(6 * C) + (uint32_t)(16 * ch);      // 32-bit byte offsets off the (image, head) base
                kq[gi][k] = ldg16_at(base + C, roff);
                vq[gi][k] = ldg16_at(base + 2 * C, roff);
            }
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
                if (4 * gq + rr < 13) QR[(t * 13 + 4 * gq + rr) * 64 + n] = acc[rr];
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
        }
    }
    __syncthreads();
    f32x4_t s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) s[kt] = mma(ld16(Ks + swz(16 * kt + fr, ks * 4 + gq)), qf[ks], s[kt]);
    }
    const int nq = n < 48 ? n : 48;
    const int aq = (nq * 37) >> 8, bq = nq - 7 * aq;
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * gq + r, kc = key < 48 ? key : 48;
            const int ak = (kc * 37) >> 8, bk = kc - 7 * ak, dh = aq - ak + 6, dw = bq - bk + 6;
            float v = scale * s[kt][r] + QR[dh * 64 + n] + QR[(13 + dw) * 64 + n] + tab[dh * 13 + dw];
            v = key < 49 ? v : -INFINITY;
            s[kt][r] = v;
            m = fmaxf(m, v);
        }
    m = xor16_max(m);
    m = xor32_max(m);
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = __expf(s[kt][r] - m);
            s[kt][r] = p;
            l += p;
        }
    l = xor16_sum(l);
    l = xor32_sum(l);
    if (gq == 0 && n < 49) lse[(int64_t)blockIdx.x * 49 + n] = m + __logf(l);
    f32x4_t oa[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) oa[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const uint4 pf = pack_bf16x8(s[2 * kk][0], s[2 * kk][1], s[2 * kk][2], s[2 * kk][3], s[2 * kk + 1][0], s[2 * kk + 1][1], s[2 * kk + 1][2], s[2 * kk + 1][3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            oa[dt] = mma(rows_frag_tr(Vs, 32 * kk + 4 * gq, dt, fr), pf, oa[dt]);      // V^T fragment (d = 16 dt + fr; keys 32 kk + 4 gq .. + 3, + 16 ..) by transpose read
        }
    }
    if (qtok >= 0) {
        const float inv = 1.0f / l;
        bf16_t* op = o + ((int64_t)b * N + qtok) * C + h * HD + 4 * gq;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) store4(op + 16 * dt, make_float4(oa[dt][0] * inv, oa[dt][1] * inv, oa[dt][2] * inv, oa[dt][3] * inv));
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

int mtp_rvsa_fwd_mfma_launch(const void* qkv, const float* samp, void* o, float* lse, const float* rel_h, const float* rel_w, const float* bias_table,
                             int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s) {
    const RvsaGeom g = make_geom(Hp, Wp, heads);
    hipLaunchKernelGGL(rvsa_fwd4_mfma_kernel, dim3((unsigned)(B * g.nh * g.nw * heads)), dim3(256), 0, s, (const bf16_t*)qkv, samp, (bf16_t*)o, lse,
                       rel_h, rel_w, bias_table, g, scale);
    return mtp_launch_status();
}
