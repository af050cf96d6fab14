// DCNv3 core operator (InternImage) for gfx950: the reference's only native extension, dcnv3_forward / dcnv3_backward
// (Multi-Task_Pretrain/backbone/ops_dcnv3/src/dcnv3.h:20-59, kernels in src/cuda/dcnv3_im2col_cuda.cuh), rebuilt for 64-wide
// waves.  HBM / gather-bound byte work: no matrix cores here.
//
//   out[n, ho, wo, g, c] = sum_p mask[n, ho, wo, g, p] * bilinear(input[n, :, :, g, c], loc_h(p), loc_w(p))
//   loc_w = p0_w - ((dil_w (kw-1)) >> 1) os + (i dil_w + off_x) os,  p0_w = ((dil_w (kw-1)) >> 1) - pad_w + wo stride_w
//   point p = i * kh + j (i over kernel_w, j over kernel_h), the centre skipped when remove_center
//
// Forward: one lane = (pixel, group, 8-channel chunk) -- 16-byte gathers, two lanes per 16-channel group, every load
// unconditional on a clamped address with the border rule folded into the weights (a branch around a load makes hipcc wait
// for each load separately).  Backward: one lane = (pixel, group, channel) so that the 16 lanes of a group issue ONE
// 64-byte-coalesced atomic per corner (the reference's layout too), the channel sums for d(offset) / d(mask) are butterfly
// reductions inside the 16-lane row instead of the reference's shared-memory tree.
#include "common.h"

namespace {

struct DcnGeom {
    int N, H, W, Ho, Wo, G, GC, kh, kw, sh, sw, ph, pw, dh, dw, P, remove_center;
    float os;
    int xcd_order;   // 1: each of the 8 XCDs takes a contiguous range of workgroups (= of pixels), see block_index()
    int scatter_bwd; // 1: always the per-corner atomic scatter backward (A/B, variant bit 1)
    int fwd_generic; // 1: the generic forward also for 3 x 3 kernels (A/B: MTP_DCN_FWD=generic)
    int form3x3_bwd; // 1: the 3 x 3 form of the gather backward where it applies (variant bit 2): faster while offsets stay below a pixel, slower beyond
    void* goff_act;  // mtp_dcnv3_bwd_act: also write grad_offset in the input dtype, rows of goff_act_ld >= G * P * 2 elements (pad columns zeroed)
    int goff_act_ld;
};

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Neighbouring pixels gather from / scatter into the
// same input rows, so with the plain order every XCD touches every line; giving each XCD a contiguous range of the flat
// (n, ho, wo, g) order keeps a line's readers and its atomic writers on one L2 (bijective for any grid size).
__device__ __forceinline__ int64_t block_index(const DcnGeom& g) {
    if (!g.xcd_order) return blockIdx.x;
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
    return (int64_t)(xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

struct Item {   // (pixel, group, chunk) of a flat lane index
    int64_t item, pix;
    int n, ho, wo, gi, chunk;
};
__device__ __forceinline__ Item decode(const DcnGeom& g, int64_t idx, int chunks) {
    Item it;
    it.chunk = (int)(idx % chunks);
    it.item = idx / chunks;
    it.gi = (int)(it.item % g.G);
    it.pix = it.item / g.G;
    it.wo = (int)(it.pix % g.Wo);
    const int64_t t = it.pix / g.Wo;
    it.ho = (int)(t % g.Ho);
    it.n = (int)(t / g.Ho);
    return it;
}

struct Point {   // one sampling point: corner rows (clamped), corner flags and bilinear fractions
    int o00, o01, o10, o11;       // element offsets of the four corner pixels inside one image (clamped into the map; H*W*C < 2^31)
    float lh, lw;                 // fractions; hh = 1 - lh, hw = 1 - lw
    float k00, k01, k10, k11;     // 1.0 where the corner is inside the map AND the point is valid, else 0.0
};
__device__ __forceinline__ Point make_point(const DcnGeom& g, float loc_h, float loc_w, int C) {
    Point p;
    const bool valid = loc_h > -1.f && loc_w > -1.f && loc_h < (float)g.H && loc_w < (float)g.W;   // false for NaN too
    // keep floor / int conversion defined for far-away or NaN locations (they carry zero weight)
    const float ch = fminf(fmaxf(loc_h, -2.f), (float)g.H + 1.f), cw = fminf(fmaxf(loc_w, -2.f), (float)g.W + 1.f);
    const float fh = floorf(ch), fw = floorf(cw);
    const int h0 = (int)fh, w0 = (int)fw, h1 = h0 + 1, w1 = w0 + 1;
    p.lh = ch - fh;
    p.lw = cw - fw;
    const bool a0 = h0 >= 0, a1 = h1 <= g.H - 1, b0 = w0 >= 0, b1 = w1 <= g.W - 1;
    p.k00 = (valid && a0 && b0) ? 1.f : 0.f;
    p.k01 = (valid && a0 && b1) ? 1.f : 0.f;
    p.k10 = (valid && a1 && b0) ? 1.f : 0.f;
    p.k11 = (valid && a1 && b1) ? 1.f : 0.f;
    const int h0c = min(max(h0, 0), g.H - 1), h1c = min(max(h1, 0), g.H - 1), w0c = min(max(w0, 0), g.W - 1), w1c = min(max(w1, 0), g.W - 1);
    const int r0 = h0c * g.W, r1 = h1c * g.W;
    p.o00 = (r0 + w0c) * C;
    p.o01 = (r0 + w1c) * C;
    p.o10 = (r1 + w0c) * C;
    p.o11 = (r1 + w1c) * C;
    return p;
}

template <typename T, int CPL>
__device__ __forceinline__ void load_chunk(const T* p, float (&v)[CPL]) {
    if constexpr (CPL == 8) {
        load8(p, v);
    } else {
#pragma unroll
        for (int c = 0; c < CPL; ++c) v[c] = Elem<T>::load(p + c);
    }
}
template <typename T, int CPL>
__device__ __forceinline__ void store_chunk(T* p, const float (&v)[CPL]) {
    if constexpr (CPL == 8) {
        store8(p, v);
    } else {
#pragma unroll
        for (int c = 0; c < CPL; ++c) Elem<T>::store(p + c, v[c]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int CPL>
__global__ __launch_bounds__(256) void dcnv3_fwd_kernel(const T* __restrict__ input, const T* __restrict__ offset, const T* __restrict__ mask, T* __restrict__ out,
                                                        DcnGeom g, int64_t total) {
    const int64_t idx = block_index(g) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int chunks = g.GC / CPL;
    const Item it = decode(g, idx, chunks);
    const int C = g.G * g.GC;
    const int chan = it.gi * g.GC + it.chunk * CPL;
    const T* in_n = input + (int64_t)it.n * g.H * g.W * C + chan;
    const T* offp = offset + it.item * (2 * g.P);
    const T* mp = mask + it.item * g.P;
    const int halfw = (g.dw * (g.kw - 1)) >> 1, halfh = (g.dh * (g.kh - 1)) >> 1;
    const float p0w = (float)(halfw - g.pw + it.wo * g.sw) - (float)halfw * g.os;
    const float p0h = (float)(halfh - g.ph + it.ho * g.sh) - (float)halfh * g.os;
    const int cw = g.kw / 2, chh = g.kh / 2;
    float acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[c] = 0.f;
    int p = 0;
    for (int i = 0; i < g.kw; ++i) {
        for (int j = 0; j < g.kh; ++j) {
            if (g.remove_center && i == cw && j == chh) continue;
            const float ow = Elem<T>::load(offp + 2 * p), oh = Elem<T>::load(offp + 2 * p + 1), m = Elem<T>::load(mp + p);
            ++p;
            const Point pt = make_point(g, p0h + ((float)(j * g.dh) + oh) * g.os, p0w + ((float)(i * g.dw) + ow) * g.os, C);
            const float hh = 1.f - pt.lh, hw = 1.f - pt.lw;
            const float w00 = hh * hw * pt.k00 * m, w01 = hh * pt.lw * pt.k01 * m, w10 = pt.lh * hw * pt.k10 * m, w11 = pt.lh * pt.lw * pt.k11 * m;
            float v00[CPL], v01[CPL], v10[CPL], v11[CPL];
            load_chunk<T, CPL>(in_n + pt.o00, v00);
            load_chunk<T, CPL>(in_n + pt.o01, v01);
            load_chunk<T, CPL>(in_n + pt.o10, v10);
            load_chunk<T, CPL>(in_n + pt.o11, v11);
#pragma unroll
            for (int c = 0; c < CPL; ++c) acc[c] += w00 * v00[c] + w01 * v01[c] + w10 * v10[c] + w11 * v11[c];
        }
    }
    store_chunk<T, CPL>(out + it.pix * C + chan, acc);
}

// The 3 x 3 forward (every DCNv3 layer of InternImage: kernel 3, no remove_center), 8 channels per lane: compile-time point loops, the 18 offsets and 9 mask
// values of the lane's (pixel, group) loaded up front ((dx, dy) of a point = one 4-byte load for bf16), and the gathers issued three points = 12 loads at a time
// before the first is used -- the generic kernel above walks the points in a run-time loop, 3 two-byte loads then 4 gathers per trip, one round trip after the other.
template <typename T> struct Raw8;
template <> struct Raw8<float> {
    struct type { float4 a, b; };
    static __device__ __forceinline__ type load(const float* p) { return type{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)}; }
    static __device__ __forceinline__ void cvt(const type& v, float (&o)[8]) { o[0] = v.a.x; o[1] = v.a.y; o[2] = v.a.z; o[3] = v.a.w; o[4] = v.b.x; o[5] = v.b.y; o[6] = v.b.z; o[7] = v.b.w; }
};
template <> struct Raw8<bf16_t> {
    typedef uint4 type;
    static __device__ __forceinline__ type load(const bf16_t* p) { return ldg16(reinterpret_cast<const char*>(p)); }
    static __device__ __forceinline__ void cvt(const type& v, float (&o)[8]) {
        o[0] = bf16_bits_to_f32(v.x & 0xffffu); o[1] = bf16_bits_to_f32(v.x >> 16); o[2] = bf16_bits_to_f32(v.y & 0xffffu); o[3] = bf16_bits_to_f32(v.y >> 16);
        o[4] = bf16_bits_to_f32(v.z & 0xffffu); o[5] = bf16_bits_to_f32(v.z >> 16); o[6] = bf16_bits_to_f32(v.w & 0xffffu); o[7] = bf16_bits_to_f32(v.w >> 16);
    }
};

template <typename T> struct OffPair;
template <> struct OffPair<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float& a, float& b) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
        a = bf16_bits_to_f32(w & 0xffffu);
        b = bf16_bits_to_f32(w >> 16);
    }
};
template <> struct OffPair<float> {
    static __device__ __forceinline__ void load(const float* p, float& a, float& b) {
        const float2 w = *reinterpret_cast<const float2*>(p);
        a = w.x;
        b = w.y;
    }
};
template <typename T>
__global__ __launch_bounds__(256) void dcnv3_fwd9_kernel(const T* __restrict__ input, const T* __restrict__ offset, const T* __restrict__ mask, T* __restrict__ out,
                                                         DcnGeom g, int64_t total) {
    const int64_t idx = block_index(g) * 256 + threadIdx.x;
    if (idx >= total) return;
    const int chunks = g.GC / 8;
    const Item it = decode(g, idx, chunks);
    const int C = g.G * g.GC;
    const int chan = it.gi * g.GC + it.chunk * 8;
    const T* in_n = input + (int64_t)it.n * g.H * g.W * C + chan;
    const T* offp = offset + it.item * 18;
    const T* mp = mask + it.item * 9;
    const float p0w = (float)(g.dw - g.pw + it.wo * g.sw) - (float)g.dw * g.os;
    const float p0h = (float)(g.dh - g.ph + it.ho * g.sh) - (float)g.dh * g.os;
    float ow[9], oh[9], m[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        OffPair<T>::load(offp + 2 * p, ow[p], oh[p]);
        m[p] = Elem<T>::load(mp + p);
    }
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {      // point p = 3 i + j: i over kernel_w, j over kernel_h
        float w[3][4];
        typename Raw8<T>::type raw[3][4];      // (kept as loaded: 4 registers per 16-byte gather, converted when used)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = 3 * i + j;
            const Point pt = make_point(g, p0h + ((float)(j * g.dh) + oh[p]) * g.os, p0w + ((float)(i * g.dw) + ow[p]) * g.os, C);
            const float hh = 1.f - pt.lh, hw = 1.f - pt.lw;
            w[j][0] = hh * hw * pt.k00 * m[p]; w[j][1] = hh * pt.lw * pt.k01 * m[p]; w[j][2] = pt.lh * hw * pt.k10 * m[p]; w[j][3] = pt.lh * pt.lw * pt.k11 * m[p];
            raw[j][0] = Raw8<T>::load(in_n + pt.o00);
            raw[j][1] = Raw8<T>::load(in_n + pt.o01);
            raw[j][2] = Raw8<T>::load(in_n + pt.o10);
            raw[j][3] = Raw8<T>::load(in_n + pt.o11);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[8];
                Raw8<T>::cvt(raw[j][k], v);
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] += w[j][k] * v[c];
            }
        __builtin_amdgcn_sched_barrier(0);      // (three points = 12 gathers in flight per lane: without it hipcc either hoists all 36 -- 318 VGPRs with converted values -- or, on raw registers, schedules them so that the bf16 kernel is 5 % and the f32 kernel 25 % slower)
    }
    store8(out + it.pix * C + chan, acc);
}

// ---------------------------------------------------------------------------------------------------------------------
// SHFL: group_channels is a power of two <= 64, so the lanes of one (pixel, group) are an aligned lane group of the wave and
// the channel sums are butterflies + one plain store; otherwise they go through f32 atomics into zero-filled outputs.
template <typename T, bool SHFL>
__global__ __launch_bounds__(256) void dcnv3_bwd_kernel(const T* __restrict__ input, const T* __restrict__ offset, const T* __restrict__ mask, const T* __restrict__ grad_out,
                                                        float* __restrict__ grad_input, float* __restrict__ grad_offset, float* __restrict__ grad_mask, DcnGeom g, int64_t total) {
    const int64_t idx = block_index(g) * 256 + threadIdx.x;
    if (idx >= total) return;
    const Item it = decode(g, idx, g.GC);
    const int C = g.G * g.GC;
    const int chan = it.gi * g.GC + it.chunk;
    const int64_t img = (int64_t)it.n * g.H * g.W * C + chan;
    const T* in_n = input + img;
    float* gin_n = grad_input + img;
    const T* offp = offset + it.item * (2 * g.P);
    const T* mp = mask + it.item * g.P;
    float* goffp = grad_offset + it.item * (2 * g.P);
    float* gmp = grad_mask + it.item * g.P;
    const float top = Elem<T>::load(grad_out + it.pix * C + chan);
    const int halfw = (g.dw * (g.kw - 1)) >> 1, halfh = (g.dh * (g.kh - 1)) >> 1;
    const float p0w = (float)(halfw - g.pw + it.wo * g.sw) - (float)halfw * g.os;
    const float p0h = (float)(halfh - g.ph + it.ho * g.sh) - (float)halfh * g.os;
    const int cw = g.kw / 2, chh = g.kh / 2;
    int p = 0;
    for (int i = 0; i < g.kw; ++i) {
        for (int j = 0; j < g.kh; ++j) {
            if (g.remove_center && i == cw && j == chh) continue;
            const float ow = Elem<T>::load(offp + 2 * p), oh = Elem<T>::load(offp + 2 * p + 1), m = Elem<T>::load(mp + p);
            const Point pt = make_point(g, p0h + ((float)(j * g.dh) + oh) * g.os, p0w + ((float)(i * g.dw) + ow) * g.os, C);
            const float hh = 1.f - pt.lh, hw = 1.f - pt.lw;
            const float v00 = Elem<T>::load(in_n + pt.o00) * pt.k00, v01 = Elem<T>::load(in_n + pt.o01) * pt.k01;
            const float v10 = Elem<T>::load(in_n + pt.o10) * pt.k10, v11 = Elem<T>::load(in_n + pt.o11) * pt.k11;
            const float tg = top * m;
            if (pt.k00 != 0.f) atomicAdd(gin_n + pt.o00, hh * hw * tg);
            if (pt.k01 != 0.f) atomicAdd(gin_n + pt.o01, hh * pt.lw * tg);
            if (pt.k10 != 0.f) atomicAdd(gin_n + pt.o10, pt.lh * hw * tg);
            if (pt.k11 != 0.f) atomicAdd(gin_n + pt.o11, pt.lh * pt.lw * tg);
            float gm = top * (hh * hw * v00 + hh * pt.lw * v01 + pt.lh * hw * v10 + pt.lh * pt.lw * v11);
            float gw = g.os * tg * (hh * (v01 - v00) + pt.lh * (v11 - v10));
            float gh = g.os * tg * (hw * (v10 - v00) + pt.lw * (v11 - v01));
            if constexpr (SHFL) {
                for (int o = g.GC >> 1; o > 0; o >>= 1) {
                    gm += __shfl_xor(gm, o, 64);
                    gw += __shfl_xor(gw, o, 64);
                    gh += __shfl_xor(gh, o, 64);
                }
                if (it.chunk == 0) {
                    gmp[p] = gm;
                    goffp[2 * p] = gw;
                    goffp[2 * p + 1] = gh;
                }
            } else {
                atomicAdd(gmp + p, gm);
                atomicAdd(goffp + 2 * p, gw);
                atomicAdd(goffp + 2 * p + 1, gh);
            }
            ++p;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward without the per-corner scatter (stride 1, "same" padding, 16-channel groups, <= 9 points: every InternImage level).
//
//   grad_input[i][c] = sum_o grad_out[o][c] * S(o, i),   S(o, i) = sum_p mask[o][p] * hat(loc_w(o, p) - i_x) * hat(loc_h(o, p) - i_y),
//   hat(d) = max(0, 1 - |d|)  (the bilinear corner weight written from the corner's side)
//
// S does not depend on the channel: one lane = one INPUT pixel of one group (16 f32 accumulators) walks the (2R+1)^2 output pixels
// around it, computes S from the sample locations staged once per workgroup in LDS, and adds S * grad_out[o][0..15] -- plain
// stores, no atomics, no zero-fill (the scatter form: 36 f32 atomics per element = 67 % of the old kernel).
// A lane looks at the output pixels within R of its own pixel, so exactly those corners of a sample are counted here that lie within R pixels
// (per axis) of the sample's own output pixel; the corners beyond that reach are scattered with atomics by the offset / mask kernel below,
// which applies the complementary test per corner: together every corner is counted exactly once.  R = ceil(half kernel span *
// offset_scale) + 1, i.e. offsets up to one pixel outwards stay entirely on the fast path (3 for InternImage's 3 x 3, offset_scale 2).
constexpr int DT_TILE = 16;   // input tile edge: 256 lanes = 16 x 16 pixels of one group

template <typename T>
struct DtLds {   // per window pixel: loc_w[9] | loc_h[9] | mask'[9] | pad (28 f32 = 112 B), then grad_out[16] -- pitch / 16 odd: conflict-free b128 rows
    static constexpr int kPitch = 112 + 16 * (int)sizeof(T);
};

__device__ __forceinline__ bool sample_valid(const DcnGeom& g, float loc_h, float loc_w) {
    return loc_h > -1.f && loc_w > -1.f && loc_h < (float)g.H && loc_w < (float)g.W;
}
template <typename T, int R>
__global__ __launch_bounds__(256, 2) void dcnv3_bwd_input_kernel(const T* __restrict__ offset, const T* __restrict__ mask, const T* __restrict__ grad_out,
                                                                 float* __restrict__ grad_input, DcnGeom g, int tiles_x, int tiles_y) {
    constexpr int WW = DT_TILE + 2 * R, NWIN = WW * WW, PITCH = DtLds<T>::kPitch;
    extern __shared__ __attribute__((aligned(16))) char dt_sm[];
    const int tid = threadIdx.x;
    int64_t b = block_index(g);
    const int gi = (int)(b % g.G);
    b /= g.G;
    const int tx0 = (int)(b % tiles_x) * DT_TILE;
    b /= tiles_x;
    const int ty0 = (int)(b % tiles_y) * DT_TILE, n = (int)(b / tiles_y);
    const int C = g.G * 16;
    const int halfw = (g.dw * (g.kw - 1)) >> 1, halfh = (g.dh * (g.kh - 1)) >> 1;
    const int cw = g.kw / 2, chh = g.kh / 2;
    // ---- stage the window: sample locations, masks (0 for invalid / far samples and for window pixels outside the map), grad_out
    for (int w = tid; w < NWIN; w += 256) {
        const int wy = w / WW, wx = w - wy * WW;
        const int ho = ty0 - R + wy, wo = tx0 - R + wx;
        const bool inmap = ho >= 0 && ho < g.H && wo >= 0 && wo < g.W;
        const int hc = min(max(ho, 0), g.H - 1), wc = min(max(wo, 0), g.W - 1);
        const int64_t pix = ((int64_t)n * g.H + hc) * g.W + wc, item = pix * g.G + gi;
        const T* offp = offset + item * (2 * g.P);
        const T* mp = mask + item * g.P;
        float* f = reinterpret_cast<float*>(dt_sm + w * PITCH);
        const float p0w = (float)(halfw - g.pw + wc) - (float)halfw * g.os;
        const float p0h = (float)(halfh - g.ph + hc) - (float)halfh * g.os;
        int p = 0;
        for (int i = 0; i < g.kw; ++i)
            for (int j = 0; j < g.kh; ++j) {
                if (g.remove_center && i == cw && j == chh) continue;
                const float ow = Elem<T>::load(offp + 2 * p), oh = Elem<T>::load(offp + 2 * p + 1), m = Elem<T>::load(mp + p);
                const float loc_h = p0h + ((float)(j * g.dh) + oh) * g.os, loc_w = p0w + ((float)(i * g.dw) + ow) * g.os;
                const bool take = inmap && sample_valid(g, loc_h, loc_w);      // (every corner within R of (ho, wo) is picked up below; the others: dcnv3_bwd_om_kernel)
                f[p] = take ? loc_w : -1e9f;
                f[9 + p] = take ? loc_h : -1e9f;
                f[18 + p] = take ? m : 0.f;
                ++p;
            }
        for (; p < 9; ++p) {
            f[p] = -1e9f;
            f[9 + p] = -1e9f;
            f[18 + p] = 0.f;
        }
        f[27] = 0.f;
        const T* gp = grad_out + pix * C + gi * 16;
        uint4* dst = reinterpret_cast<uint4*>(dt_sm + w * PITCH + 112);
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int c = 0; c < (int)sizeof(T) * 16 / 16; ++c) dst[c] = inmap ? ldg16(reinterpret_cast<const char*>(gp) + 16 * c) : z;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    const int iy = ty0 + ty, ix = tx0 + tx;
    const float fy = (float)iy, fx = (float)ix;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    const char* base = dt_sm + (ty * WW + tx) * PITCH;
#pragma unroll 1
    for (int dy = 0; dy <= 2 * R; ++dy) {
#pragma unroll
        for (int dx = 0; dx <= 2 * R; ++dx) {
            const char* e = base + (dy * WW + dx) * PITCH;
            float q[28];
#pragma unroll
            for (int v = 0; v < 7; ++v) {
                const float4 t = *reinterpret_cast<const float4*>(e + 16 * v);
                q[4 * v] = t.x; q[4 * v + 1] = t.y; q[4 * v + 2] = t.z; q[4 * v + 3] = t.w;
            }
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < 9; ++p) {
                const float wx = fmaxf(1.f - fabsf(q[p] - fx), 0.f), wy = fmaxf(1.f - fabsf(q[9 + p] - fy), 0.f);      // (__saturatef here: 556 -> 662 us at the 128 x 128 level)
                s = fmaf(wx * wy, q[18 + p], s);
            }
            if (__builtin_amdgcn_ballot_w64(s != 0.f) == 0) continue;      // (wave-uniform) no sample of these 64 output pixels reaches its lane's pixel
            float d[16];
            if constexpr (sizeof(T) == 2) {
                load8(reinterpret_cast<const bf16_t*>(e + 112), reinterpret_cast<float(&)[8]>(d[0]));
                load8(reinterpret_cast<const bf16_t*>(e + 128), reinterpret_cast<float(&)[8]>(d[8]));
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float4 t = *reinterpret_cast<const float4*>(e + 112 + 16 * v);
                    d[4 * v] = t.x; d[4 * v + 1] = t.y; d[4 * v + 2] = t.z; d[4 * v + 3] = t.w;
                }
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = fmaf(s, d[c], acc[c]);
        }
    }
    if (iy < g.H && ix < g.W) {
        float* dst = grad_input + (((int64_t)n * g.H + iy) * g.W + ix) * C + gi * 16;
#pragma unroll
        for (int v = 0; v < 4; ++v) *reinterpret_cast<float4*>(dst + 4 * v) = make_float4(acc[4 * v], acc[4 * v + 1], acc[4 * v + 2], acc[4 * v + 3]);
    }
}

// The same for InternImage's own geometry -- 3 x 3 points, dilation 1, offset_scale OS = 1 or 2 -- where point (pi, pj) of output pixel o sits
// nominally at o + OS (pi - 1, pj - 1): here the reach is the corners within one pixel of that nominal position, o + OS (pi - 1, pj - 1) +
// {-1, 0, 1}^2 (the others go through the atomics), so an input pixel has 9 x 9 = 81 candidate samples instead of the window's 9 (2R+1)^2 = 441,
// each at a compile-time LDS offset from the lane's own window entry.  S(o, i) is collected in (2R+1)^2 registers (R = OS + 1) and applied to
// grad_out[o] afterwards, skipping the output pixels no lane of the wave got a weight from.
template <typename T, int OS>
__global__ __launch_bounds__(256, 2) void dcnv3_bwd_input3x3_kernel(const T* __restrict__ offset, const T* __restrict__ mask, const T* __restrict__ grad_out,
                                                                    float* __restrict__ grad_input, DcnGeom g, int tiles_x, int tiles_y) {
    constexpr int R = OS + 1, D = 2 * R + 1, WW = DT_TILE + 2 * R, NWIN = WW * WW, PITCH = DtLds<T>::kPitch;
    extern __shared__ __attribute__((aligned(16))) char dt_sm[];
    const int tid = threadIdx.x;
    int64_t b = block_index(g);
    const int gi = (int)(b % g.G);
    b /= g.G;
    const int tx0 = (int)(b % tiles_x) * DT_TILE;
    b /= tiles_x;
    const int ty0 = (int)(b % tiles_y) * DT_TILE, n = (int)(b / tiles_y);
    const int C = g.G * 16;
    // ---- stage the window.  Entry: (loc_w, loc_h)[slot 0..8] | mask'[slot 0..8] | pad | grad_out[16]; slot = 3 pi + pj (the centre slot stays empty
    // under remove_center)
    for (int w = tid; w < NWIN; w += 256) {
        const int wy = w / WW, wx = w - wy * WW;
        const int ho = ty0 - R + wy, wo = tx0 - R + wx;
        const bool inmap = ho >= 0 && ho < g.H && wo >= 0 && wo < g.W;
        const int hc = min(max(ho, 0), g.H - 1), wc = min(max(wo, 0), g.W - 1);
        const int64_t pix = ((int64_t)n * g.H + hc) * g.W + wc, item = pix * g.G + gi;
        const T* offp = offset + item * (2 * g.P);
        const T* mp = mask + item * g.P;
        float* f = reinterpret_cast<float*>(dt_sm + w * PITCH);
        const float p0w = (float)(1 - g.pw + wc) - g.os, p0h = (float)(1 - g.ph + hc) - g.os;      // (halfw = halfh = 1)
        int p = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int q = 3 * i + j;
                float lw = -1e9f, lh = -1e9f, mm = 0.f;
                if (!(g.remove_center && i == 1 && j == 1)) {
                    const float ow = Elem<T>::load(offp + 2 * p), oh = Elem<T>::load(offp + 2 * p + 1), m = Elem<T>::load(mp + p);
                    const float loc_h = p0h + ((float)j + oh) * g.os, loc_w = p0w + ((float)i + ow) * g.os;
                    const bool take = inmap && sample_valid(g, loc_h, loc_w);      // (corners within one pixel of the nominal position are picked up below; the others: dcnv3_bwd_om_kernel)
                    lw = take ? loc_w : -1e9f;
                    lh = take ? loc_h : -1e9f;
                    mm = take ? m : 0.f;
                    ++p;
                }
                f[2 * q] = lw;
                f[2 * q + 1] = lh;
                f[18 + q] = mm;
            }
        f[27] = 0.f;
        const T* gp = grad_out + pix * C + gi * 16;
        uint4* dst = reinterpret_cast<uint4*>(dt_sm + w * PITCH + 112);
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int c = 0; c < (int)sizeof(T) * 16 / 16; ++c) dst[c] = inmap ? ldg16(reinterpret_cast<const char*>(gp) + 16 * c) : z;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    const int iy = ty0 + ty, ix = tx0 + tx;
    const float fy = (float)iy, fx = (float)ix;
    const char* base = dt_sm + (ty * WW + tx) * PITCH;
    float s[D * D];
#pragma unroll
    for (int d = 0; d < D * D; ++d) s[d] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int q = 3 * i + j;
#pragma unroll
            for (int ey = -1; ey <= 1; ++ey)
#pragma unroll
                for (int ex = -1; ex <= 1; ++ex) {
                    const int dx = ex - OS * (i - 1) + R, dy = ey - OS * (j - 1) + R;      // window entry of the output pixel o = input pixel + (dx - R, dy - R)
                    const char* e = base + (dy * WW + dx) * PITCH;
                    const float2 l = *reinterpret_cast<const float2*>(e + 8 * q);
                    const float m = *reinterpret_cast<const float*>(e + 72 + 4 * q);
                    const float wx = __saturatef(1.f - fabsf(l.x - fx)), wy = __saturatef(1.f - fabsf(l.y - fy));
                    s[dy * D + dx] = fmaf(wx * wy, m, s[dy * D + dx]);
                }
            // one point's 18 LDS reads in flight at a time: hipcc otherwise issues all 162 up front and sinks the weight arithmetic down to the
            // per-output-pixel blocks below, keeping 243 loaded values alive (256 VGPRs + scratch)
#pragma unroll
            for (int d = 0; d < D * D; ++d) asm volatile("" : "+v"(s[d]));
            __builtin_amdgcn_sched_barrier(0);
        }
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
    for (int d = 0; d < D * D; ++d) {
        if (__builtin_amdgcn_ballot_w64(s[d] != 0.f) == 0) continue;      // (wave-uniform)
        const char* e = base + ((d / D) * WW + (d % D)) * PITCH;
        float v[16];
        if constexpr (sizeof(T) == 2) {
            load8(reinterpret_cast<const bf16_t*>(e + 112), reinterpret_cast<float(&)[8]>(v[0]));
            load8(reinterpret_cast<const bf16_t*>(e + 128), reinterpret_cast<float(&)[8]>(v[8]));
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 t = *reinterpret_cast<const float4*>(e + 112 + 16 * u);
                v[4 * u] = t.x; v[4 * u + 1] = t.y; v[4 * u + 2] = t.z; v[4 * u + 3] = t.w;
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = fmaf(s[d], v[c], acc[c]);
    }
    if (iy < g.H && ix < g.W) {
        float* dst = grad_input + (((int64_t)n * g.H + iy) * g.W + ix) * C + gi * 16;
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(dst + 4 * u) = make_float4(acc[4 * u], acc[4 * u + 1], acc[4 * u + 2], acc[4 * u + 3]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// float64 (MTP_F64): the reference's double dispatch.  One lane = (pixel, group, channel), double arithmetic throughout (sample locations included), f64
// atomics for the three gradients (zero-filled by the launcher).  Same formulas as dcnv3_fwd_kernel / dcnv3_bwd_kernel above.
struct PointD {
    int o00, o01, o10, o11;
    double lh, lw, k00, k01, k10, k11;
};
__device__ __forceinline__ PointD make_point_d(const DcnGeom& g, double loc_h, double loc_w, int C) {
    PointD p;
    const bool valid = loc_h > -1.0 && loc_w > -1.0 && loc_h < (double)g.H && loc_w < (double)g.W;
    const double ch = fmin(fmax(loc_h, -2.0), (double)g.H + 1.0), cw = fmin(fmax(loc_w, -2.0), (double)g.W + 1.0);
    const double fh = floor(ch), fw = floor(cw);
    const int h0 = (int)fh, w0 = (int)fw, h1 = h0 + 1, w1 = w0 + 1;
    p.lh = ch - fh;
    p.lw = cw - fw;
    const bool a0 = h0 >= 0, a1 = h1 <= g.H - 1, b0 = w0 >= 0, b1 = w1 <= g.W - 1;
    p.k00 = (valid && a0 && b0) ? 1.0 : 0.0;
    p.k01 = (valid && a0 && b1) ? 1.0 : 0.0;
    p.k10 = (valid && a1 && b0) ? 1.0 : 0.0;
    p.k11 = (valid && a1 && b1) ? 1.0 : 0.0;
    const int h0c = min(max(h0, 0), g.H - 1), h1c = min(max(h1, 0), g.H - 1), w0c = min(max(w0, 0), g.W - 1), w1c = min(max(w1, 0), g.W - 1);
    p.o00 = (h0c * g.W + w0c) * C;
    p.o01 = (h0c * g.W + w1c) * C;
    p.o10 = (h1c * g.W + w0c) * C;
    p.o11 = (h1c * g.W + w1c) * C;
    return p;
}
template <bool BWD>
__global__ __launch_bounds__(256) void dcnv3_f64_kernel(const double* __restrict__ input, const double* __restrict__ offset, const double* __restrict__ mask,
                                                       const double* __restrict__ grad_out, double* __restrict__ out, double* __restrict__ grad_input,
                                                       double* __restrict__ grad_offset, double* __restrict__ grad_mask, DcnGeom g, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const Item it = decode(g, idx, g.GC);
    const int C = g.G * g.GC;
    const int chan = it.gi * g.GC + it.chunk;
    const int64_t img = (int64_t)it.n * g.H * g.W * C + chan;
    const double* in_n = input + img;
    const double* offp = offset + it.item * (2 * g.P);
    const double* mp = mask + it.item * g.P;
    const double os = (double)g.os;
    const int halfw = (g.dw * (g.kw - 1)) >> 1, halfh = (g.dh * (g.kh - 1)) >> 1;
    const double p0w = (double)(halfw - g.pw + it.wo * g.sw) - (double)halfw * os;
    const double p0h = (double)(halfh - g.ph + it.ho * g.sh) - (double)halfh * os;
    const int cw = g.kw / 2, chh = g.kh / 2;
    const double top = BWD ? grad_out[it.pix * C + chan] : 0.0;
    double acc = 0.0;
    int p = 0;
    for (int i = 0; i < g.kw; ++i)
        for (int j = 0; j < g.kh; ++j) {
            if (g.remove_center && i == cw && j == chh) continue;
            const double ow = offp[2 * p], oh = offp[2 * p + 1], m = mp[p];
            const PointD pt = make_point_d(g, p0h + ((double)(j * g.dh) + oh) * os, p0w + ((double)(i * g.dw) + ow) * os, C);
            const double hh = 1.0 - pt.lh, hw = 1.0 - pt.lw;
            const double v00 = in_n[pt.o00] * pt.k00, v01 = in_n[pt.o01] * pt.k01, v10 = in_n[pt.o10] * pt.k10, v11 = in_n[pt.o11] * pt.k11;
            if constexpr (!BWD) {
                acc += m * (hh * hw * v00 + hh * pt.lw * v01 + pt.lh * hw * v10 + pt.lh * pt.lw * v11);
            } else {
                const double tg = top * m;
                double* gin_n = grad_input + img;
                if (pt.k00 != 0.0) atomicAdd(gin_n + pt.o00, hh * hw * tg);
                if (pt.k01 != 0.0) atomicAdd(gin_n + pt.o01, hh * pt.lw * tg);
                if (pt.k10 != 0.0) atomicAdd(gin_n + pt.o10, pt.lh * hw * tg);
                if (pt.k11 != 0.0) atomicAdd(gin_n + pt.o11, pt.lh * pt.lw * tg);
                atomicAdd(grad_mask + it.item * g.P + p, top * (hh * hw * v00 + hh * pt.lw * v01 + pt.lh * hw * v10 + pt.lh * pt.lw * v11));
                atomicAdd(grad_offset + it.item * (2 * g.P) + 2 * p, os * tg * (hh * (v01 - v00) + pt.lh * (v11 - v10)));
                atomicAdd(grad_offset + it.item * (2 * g.P) + 2 * p + 1, os * tg * (hw * (v10 - v00) + pt.lw * (v11 - v01)));
            }
            ++p;
        }
    if constexpr (!BWD) out[it.pix * C + chan] = acc;
}

// d(offset), d(mask): one lane = (output pixel, group, 8-channel half) -- the sample location is computed twice (not once per channel lane as in the
// scatter kernel), the channel sums are 8 in-lane terms + one exchange with the neighbour lane, the four corner rows are one (bf16) / two (f32)
// 16-byte loads each, exactly the forward's gather.
// Corners beyond the gather form's reach (see above) scatter their data gradient here, the whole wave working on one sample at a time: lane =
// (corner, channel), so that a sample costs one atomic instruction of up to four 64-byte requests -- what the scatter kernel issues per sample,
// without its 16 lanes per (pixel, group).  The sample's owner passes location / weights through readlane, the pair's 16 grad_out values go
// through LDS.  NEAR = 1: the window form's reach R around the output pixel; NEAR = 2: the 3 x 3 form's reach of one pixel around the nominal
// position (OS = R - 1).
// eight consecutive elements as loaded (16 bytes of bf16 stay packed until they are used)
template <typename T, int NEAR>
__global__ __launch_bounds__(256) void dcnv3_bwd_om_kernel(const T* __restrict__ input, const T* __restrict__ offset, const T* __restrict__ mask, const T* __restrict__ grad_out,
                                                           float* __restrict__ grad_input, float* __restrict__ grad_offset, float* __restrict__ grad_mask, DcnGeom g, int64_t total,
                                                           int R) {
    __shared__ float tops[4][64][9];
    // the wave's results -- d(mask) 32 items x 9, d(offset) 32 items x 18 f32 -- collected here and written out as whole 16-byte pieces of the wave's contiguous
    // output ranges (round 6): item by item they were 9 four-byte + 9 eight-byte (+ 18 two-byte) stores per lane, 40 % of the kernel at the 128 x 128 level
    __shared__ __attribute__((aligned(16))) float stage[4][32 * 27];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t idx_raw = block_index(g) * 256 + threadIdx.x;
    const bool live = idx_raw < 2 * total;               // (no early return: the far-sample scatter below needs all 64 lanes)
    const int64_t item0 = (idx_raw - lane) >> 1;         // first item of this wave
    const bool staged = g.P == 9 && item0 + 32 <= total; // (wave-uniform) all 32 items live, 9 points: the coalesced write-out below
    float* stM = stage[wave];
    float* stO = stage[wave] + 32 * 9;
    const int64_t idx = live ? idx_raw : 2 * total - 1;
    const int half = (int)(idx & 1);
    const int64_t item = idx >> 1;
    const int gi = (int)(item % g.G);
    const int64_t pix = item / g.G;
    const int wo = (int)(pix % g.Wo);
    const int64_t t = pix / g.Wo;
    const int ho = (int)(t % g.Ho), n = (int)(t / g.Ho);
    const int C = g.G * 16;
    const int64_t img = (int64_t)n * g.H * g.W * C + gi * 16;
    const T* in_n = input + img + 8 * half;
    const T* offp = offset + item * (2 * g.P);
    const T* mp = mask + item * g.P;
    float* goffp = grad_offset + item * (2 * g.P);
    float* gmp = grad_mask + item * g.P;
    // every offset / mask value of the (pixel, group) first: the kernel is a chain of dependent loads (offset -> location -> corner rows), and
    // in the training step its operands are cold -- point by point it paid nine such chains one after the other (170 us even on 16 x 16 maps)
    float ow[9], oh[9], mk[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        const int pc = p < g.P ? p : g.P - 1;
        ow[p] = Elem<T>::load(offp + 2 * pc);
        oh[p] = Elem<T>::load(offp + 2 * pc + 1);
        mk[p] = Elem<T>::load(mp + pc);
    }
    float top[8];
    load8(grad_out + pix * C + gi * 16 + 8 * half, top);
#pragma unroll
    for (int c = 0; c < 8; ++c) tops[wave][lane][c] = top[c];      // (read by this wave only: LDS operations of one wave execute in order)
    const int halfw = (g.dw * (g.kw - 1)) >> 1, halfh = (g.dh * (g.kh - 1)) >> 1;
    const float p0w = (float)(halfw - g.pw + wo * g.sw) - (float)halfw * g.os;
    const float p0h = (float)(halfh - g.ph + ho * g.sh) - (float)halfh * g.os;
    const int centre = (g.kw / 2) * g.kh + g.kh / 2;
    if (g.goff_act && live && half == 0 && gi == 0)
        for (int c = g.G * 2 * g.P; c < g.goff_act_ld; ++c) Elem<T>::store(reinterpret_cast<T*>(g.goff_act) + pix * g.goff_act_ld + c, 0.f);
#pragma unroll
    for (int grp = 0; grp < 3; ++grp) {
        if (3 * grp >= g.P) break;
        // ---- three points: locations, then all twelve corner rows in flight together
        Point pt[3];
        float loc_h[3], loc_w[3];
        int pi[3], pj[3];
        typename Raw8<T>::type raw[3][4];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int p = 3 * grp + q;
            const int pp = p + ((g.remove_center && p >= centre) ? 1 : 0);     // index in the full kw x kh grid (i over kernel_w outer)
            pi[q] = pp / g.kh;
            pj[q] = pp - pi[q] * g.kh;
            loc_h[q] = p0h + ((float)(pj[q] * g.dh) + oh[p]) * g.os;
            loc_w[q] = p0w + ((float)(pi[q] * g.dw) + ow[p]) * g.os;
            pt[q] = make_point(g, loc_h[q], loc_w[q], C);
            raw[q][0] = Raw8<T>::load(in_n + pt[q].o00);
            raw[q][1] = Raw8<T>::load(in_n + pt[q].o01);
            raw[q][2] = Raw8<T>::load(in_n + pt[q].o10);
            raw[q][3] = Raw8<T>::load(in_n + pt[q].o11);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int p = 3 * grp + q;
            const bool have = p < g.P;
            const float m = mk[p];
            float d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[8];
                Raw8<T>::cvt(raw[q][k], v);
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) acc = fmaf(top[c], v[c], acc);
                d[k] = acc + __shfl_xor(acc, 1, 64);
            }
            const float d00 = d[0] * pt[q].k00, d01 = d[1] * pt[q].k01, d10 = d[2] * pt[q].k10, d11 = d[3] * pt[q].k11;
            const float lh = pt[q].lh, lw = pt[q].lw, hh = 1.f - lh, hw = 1.f - lw;
            if (staged) {
                if (half == 0) {      // (both lanes of the pair hold the sums)
                    const float gw_ = g.os * m * (hh * (d01 - d00) + lh * (d11 - d10)), gh_ = g.os * m * (hw * (d10 - d00) + lw * (d11 - d01));
                    stM[(lane >> 1) * 9 + p] = hh * hw * d00 + hh * lw * d01 + lh * hw * d10 + lh * lw * d11;
                    *reinterpret_cast<float2*>(stO + (lane >> 1) * 18 + 2 * p) = make_float2(gw_, gh_);
                }
            } else if (live && half == 0 && have) {
                gmp[p] = hh * hw * d00 + hh * lw * d01 + lh * hw * d10 + lh * lw * d11;
                const float gw_ = g.os * m * (hh * (d01 - d00) + lh * (d11 - d10)), gh_ = g.os * m * (hw * (d10 - d00) + lw * (d11 - d01));
                *reinterpret_cast<float2*>(goffp + 2 * p) = make_float2(gw_, gh_);
                if (g.goff_act) {      // the GEMM-operand copy of the same values (the offset head's dgrad / wgrad read it): no cast-and-pad pass
                    T* ga = reinterpret_cast<T*>(g.goff_act) + pix * g.goff_act_ld + gi * (2 * g.P) + 2 * p;
                    Elem<T>::store(ga, gw_);
                    Elem<T>::store(ga + 1, gh_);
                }
            }
            // corners the gather form does not reach (per corner, not per sample: a sample astride the edge of the reach sends only its outer corners here)
            const int h0 = (int)floorf(fminf(fmaxf(loc_h[q], -2.f), (float)g.H + 1.f)), w0 = (int)floorf(fminf(fmaxf(loc_w[q], -2.f), (float)g.W + 1.f));
            const int ch = NEAR == 2 ? ho + (R - 1) * (pj[q] - 1) : ho, cw = NEAR == 2 ? wo + (R - 1) * (pi[q] - 1) : wo, reach = NEAR == 2 ? 1 : R;
            const bool out_h0 = abs(h0 - ch) > reach, out_h1 = abs(h0 + 1 - ch) > reach, out_w0 = abs(w0 - cw) > reach, out_w1 = abs(w0 + 1 - cw) > reach;
            const bool scat = live && half == 0 && have && sample_valid(g, loc_h[q], loc_w[q]);
            const float w00 = (scat && (out_h0 || out_w0)) ? hh * hw * pt[q].k00 * m : 0.f, w01 = (scat && (out_h0 || out_w1)) ? hh * lw * pt[q].k01 * m : 0.f;
            const float w10 = (scat && (out_h1 || out_w0)) ? lh * hw * pt[q].k10 * m : 0.f, w11 = (scat && (out_h1 || out_w1)) ? lh * lw * pt[q].k11 * m : 0.f;
            const bool far = w00 != 0.f || w01 != 0.f || w10 != 0.f || w11 != 0.f;
            uint64_t fm = __builtin_amdgcn_ballot_w64(far);
            if (fm) {
                // (round 6 tried the far CORNERS of the wave as one list in LDS, 16 lanes per corner and four corners per atomic instruction -- 28 instead of 66 atomic
                //  instructions per wave on 32 x 32 maps: 108.8 vs 110.0 us there, 327 vs 282 us on 128 x 128 maps where few samples are far; what the section costs
                //  is the atomics themselves (25 us of 108 with the instructions not issued) and the per-sample reach tests, not the readlanes)
                const int k = lane >> 4, c = lane & 15;
                const uint32_t img_lo = (uint32_t)((uint64_t)img & 0xffffffffu), img_hi = (uint32_t)((uint64_t)img >> 32);
                while (fm) {
                    const int L = __builtin_ctzll(fm);      // an even lane: lanes L, L + 1 hold channels 0-7, 8-15 of the sample's grad_out
                    fm &= fm - 1;
                    const int s00 = __builtin_amdgcn_readlane(pt[q].o00, L), s01 = __builtin_amdgcn_readlane(pt[q].o01, L);
                    const int s10 = __builtin_amdgcn_readlane(pt[q].o10, L), s11 = __builtin_amdgcn_readlane(pt[q].o11, L);
                    const float x00 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w00), L)), x01 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w01), L));
                    const float x10 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w10), L)), x11 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w11), L));
                    const int64_t imgL = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)img_hi, L) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)img_lo, L));
                    const int ok = k == 0 ? s00 : k == 1 ? s01 : k == 2 ? s10 : s11;
                    const float wk = k == 0 ? x00 : k == 1 ? x01 : k == 2 ? x10 : x11;
                    if (wk != 0.f) atomicAdd(grad_input + imgL + ok + c, wk * tops[wave][L + (c >> 3)][c & 7]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // (keeps the batches of 12 gathers apart: with the results going to LDS nothing else orders the next batch's loads behind this one's use)
    }
    if (staged) {
        // (the staging rows cross lanes: wavefront-scope fences around a wave barrier -- one wave's LDS operations execute in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4* gm4 = reinterpret_cast<float4*>(grad_mask + item0 * 9);        // 288 floats = 72 pieces; item0 is a multiple of 32: 16-byte aligned
        float4* go4 = reinterpret_cast<float4*>(grad_offset + item0 * 18);     // 576 floats = 144 pieces
        const float4* sm4 = reinterpret_cast<const float4*>(stM);
        const float4* so4 = reinterpret_cast<const float4*>(stO);
        gm4[lane] = sm4[lane];
        if (lane < 8) gm4[64 + lane] = sm4[64 + lane];
        go4[lane] = so4[lane];
        go4[64 + lane] = so4[64 + lane];
        if (lane < 16) go4[128 + lane] = so4[128 + lane];
        if (g.goff_act) {
            T* ga = reinterpret_cast<T*>(g.goff_act);
            if (g.goff_act_ld == g.G * 18 && sizeof(T) == 2) {      // the operand rows are contiguous too (every InternImage level: 18 G is a multiple of 8): 576 bf16 = 72 pieces
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int i8 = 64 * r + lane;
                    if (i8 < 72) {
                        const float4 a = so4[2 * i8], b = so4[2 * i8 + 1];
                        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(ga) + (item0 * 18 + 8 * i8) * 2) = pack_bf16x8(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);
                    }
                }
            } else if (half == 0) {
                T* row = ga + pix * g.goff_act_ld + gi * 18;
#pragma unroll
                for (int e = 0; e < 18; ++e) Elem<T>::store(row + e, stO[(lane >> 1) * 18 + e]);
            }
        }
    }
}

int make_geom(const mtp_dcnv3_geom* a, DcnGeom& g) {
    MTP_CHECK_ARG(a != nullptr);
    MTP_CHECK_ARG(a->N > 0 && a->H > 0 && a->W > 0 && a->group > 0 && a->group_channels > 0);
    MTP_CHECK_ARG(a->kernel_h > 0 && a->kernel_w > 0 && a->stride_h > 0 && a->stride_w > 0 && a->dilation_h > 0 && a->dilation_w > 0 && a->pad_h >= 0 && a->pad_w >= 0);
    MTP_CHECK_ARG(a->N < (1 << 30) && a->H < (1 << 15) && a->W < (1 << 15));
    // the reference's batching rule (dcnv3_cuda.cu:46-49): batch must be a multiple of min(batch, im2col_step)
    MTP_CHECK_ARG(a->im2col_step > 0 && a->N % (a->N < a->im2col_step ? a->N : a->im2col_step) == 0);
    // remove_center is defined for square odd kernels only (dcnv3_func.py:184-185, modules/dcnv3.py:258-259)
    MTP_CHECK_ARG(!a->remove_center || (a->kernel_h == a->kernel_w && (a->kernel_h & 1)));
    const int64_t Ho = (a->H + 2 * a->pad_h - (a->dilation_h * (a->kernel_h - 1) + 1)) / a->stride_h + 1;
    const int64_t Wo = (a->W + 2 * a->pad_w - (a->dilation_w * (a->kernel_w - 1) + 1)) / a->stride_w + 1;
    MTP_CHECK_ARG(Ho > 0 && Wo > 0);
    g.N = (int)a->N; g.H = (int)a->H; g.W = (int)a->W; g.Ho = (int)Ho; g.Wo = (int)Wo;
    g.G = a->group; g.GC = a->group_channels;
    g.kh = a->kernel_h; g.kw = a->kernel_w; g.sh = a->stride_h; g.sw = a->stride_w; g.ph = a->pad_h; g.pw = a->pad_w; g.dh = a->dilation_h; g.dw = a->dilation_w;
    g.remove_center = a->remove_center ? 1 : 0;
    g.P = a->kernel_h * a->kernel_w - g.remove_center;
    MTP_CHECK_ARG(g.P > 0);
    if ((int64_t)g.H * g.W * g.G * g.GC >= ((int64_t)1 << 31)) return MTP_ERR_UNSUPPORTED;   // 32-bit element offsets inside one image
    g.os = a->offset_scale;
    g.xcd_order = (a->variant & 1) ? 0 : 1;
    g.scatter_bwd = (a->variant & 2) ? 1 : 0;
    g.form3x3_bwd = (a->variant & 4) ? 1 : 0;
    g.fwd_generic = (a->variant & 8) ? 1 : 0;
    g.goff_act = nullptr;
    g.goff_act_ld = 0;
    return 0;
}
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T>
int launch_fwd(const void* input, const void* offset, const void* mask, void* output, const DcnGeom& g, hipStream_t s) {
    const int64_t items = (int64_t)g.N * g.Ho * g.Wo * g.G;
    if (g.GC % 8 == 0 && aligned16(input) && aligned16(output) && g.kh == 3 && g.kw == 3 && !g.remove_center && !g.fwd_generic &&
        (reinterpret_cast<uintptr_t>(offset) & 7u) == 0) {
        const int64_t total = items * (g.GC / 8);
        hipLaunchKernelGGL((dcnv3_fwd9_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)input, (const T*)offset, (const T*)mask, (T*)output, g, total);
    } else if (g.GC % 8 == 0 && aligned16(input) && aligned16(output)) {
        const int64_t total = items * (g.GC / 8);
        hipLaunchKernelGGL((dcnv3_fwd_kernel<T, 8>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)input, (const T*)offset, (const T*)mask, (T*)output, g, total);
    } else {
        const int64_t total = items * g.GC;
        hipLaunchKernelGGL((dcnv3_fwd_kernel<T, 1>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)input, (const T*)offset, (const T*)mask, (T*)output, g, total);
    }
    return mtp_launch_status();
}

// OS3 = 0: the window form with reach R; OS3 = 1 / 2: the 3 x 3 form for offset_scale 1 / 2 (R = OS3 + 1)
template <typename T, int R, int OS3>
int launch_bwd_gather(const void* input, const void* offset, const void* mask, const void* grad_output, float* grad_input, float* grad_offset, float* grad_mask, const DcnGeom& g,
                      hipStream_t s, int tiles_x, int tiles_y, int64_t blocks, int64_t items) {
    constexpr int LDS = (DT_TILE + 2 * R) * (DT_TILE + 2 * R) * DtLds<T>::kPitch;
    static unsigned long long optin = 0;     // (one per instantiation of this launcher: per kernel; the bit set inside is per device)
    if (const int e = mtp_optin_lds(OS3 ? (const void*)dcnv3_bwd_input3x3_kernel<T, (OS3 ? OS3 : 1)> : (const void*)dcnv3_bwd_input_kernel<T, R>, LDS, optin)) return e;
    if constexpr (OS3 != 0)
        hipLaunchKernelGGL((dcnv3_bwd_input3x3_kernel<T, (OS3 ? OS3 : 1)>), dim3((unsigned)blocks), dim3(256), LDS, s, (const T*)offset, (const T*)mask, (const T*)grad_output, grad_input, g, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL((dcnv3_bwd_input_kernel<T, R>), dim3((unsigned)blocks), dim3(256), LDS, s, (const T*)offset, (const T*)mask, (const T*)grad_output, grad_input, g, tiles_x, tiles_y);
    hipLaunchKernelGGL((dcnv3_bwd_om_kernel<T, (OS3 ? 2 : 1)>), dim3((unsigned)((2 * items + 255) / 256)), dim3(256), 0, s, (const T*)input, (const T*)offset, (const T*)mask, (const T*)grad_output,
                       grad_input, grad_offset, grad_mask, g, items, R);
    return mtp_launch_status();
}

template <typename T>
int launch_bwd(const void* input, const void* offset, const void* mask, const void* grad_output, float* grad_input, float* grad_offset, float* grad_mask, const DcnGeom& g,
               hipStream_t s) {
    const int64_t items = (int64_t)g.N * g.Ho * g.Wo * g.G, total = items * g.GC;
    const bool shfl = g.GC <= 64 && (g.GC & (g.GC - 1)) == 0;
    // the gather form (no scatter): stride 1, "same" padding, 16-channel groups, <= 9 points, reach R <= 3
    const int halfw = (g.dw * (g.kw - 1)) >> 1, halfh = (g.dh * (g.kh - 1)) >> 1;
    const int reach = (int)ceilf((float)(halfw > halfh ? halfw : halfh) * fabsf(g.os)) + 1;
    if (!g.scatter_bwd && g.GC == 16 && g.P <= 9 && g.sh == 1 && g.sw == 1 && g.Ho == g.H && g.Wo == g.W && g.ph == halfh && g.pw == halfw && reach <= 3 &&
        aligned16(input) && aligned16(grad_output) && aligned16(grad_input) && (reinterpret_cast<uintptr_t>(grad_offset) & 7u) == 0) {
        const int tiles_x = (g.W + DT_TILE - 1) / DT_TILE, tiles_y = (g.H + DT_TILE - 1) / DT_TILE;
        const int64_t blocks = (int64_t)g.N * tiles_y * tiles_x * g.G;
        if (blocks < ((int64_t)1 << 31) && 2 * items < ((int64_t)1 << 32) - 256) {
#define MTP_DCN_GATHER(R_, OS3_) launch_bwd_gather<T, R_, OS3_>(input, offset, mask, grad_output, grad_input, grad_offset, grad_mask, g, s, tiles_x, tiles_y, blocks, items)
            // default: the window form.  The 3 x 3 form is 8 % faster for offsets below a pixel (a freshly initialised network: the offset head starts at
            // zero) but its reach is one pixel around the nominal position -- with offsets of sigma = 0.5 ... 1.6 px (bench.py re-draws the heads like
            // fixture f12) 13 ... 77 % of the samples leave it and go through the atomics: InternImage-XL step 72.7 vs 69.1 ms (same box).
            if (g.kh == 3 && g.kw == 3 && g.dh == 1 && g.dw == 1 && g.form3x3_bwd) {
                if (g.os == 1.0f) return MTP_DCN_GATHER(2, 1);
                if (g.os == 2.0f) return MTP_DCN_GATHER(3, 2);
            }
            return reach <= 2 ? MTP_DCN_GATHER(2, 0) : MTP_DCN_GATHER(3, 0);
#undef MTP_DCN_GATHER
        }
    }
    if (g.goff_act) return MTP_ERR_UNSUPPORTED;      // (the operand copy exists in the gather form only; the caller casts grad_offset itself)
    hipError_t e = hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)g.N * g.H * g.W * g.G * g.GC, s);   // the reference's at::zeros_like (dcnv3_cuda.cu:131)
    if (e != hipSuccess) return (int)e;
    if (!shfl) {
        e = hipMemsetAsync(grad_offset, 0, sizeof(float) * (size_t)items * 2 * g.P, s);
        if (e == hipSuccess) e = hipMemsetAsync(grad_mask, 0, sizeof(float) * (size_t)items * g.P, s);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL((dcnv3_bwd_kernel<T, false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)input, (const T*)offset, (const T*)mask, (const T*)grad_output,
                           grad_input, grad_offset, grad_mask, g, total);
    } else {
        hipLaunchKernelGGL((dcnv3_bwd_kernel<T, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)input, (const T*)offset, (const T*)mask, (const T*)grad_output,
                           grad_input, grad_offset, grad_mask, g, total);
    }
    return mtp_launch_status();
}

}  // namespace

extern "C" int mtp_dcnv3_out_size(const mtp_dcnv3_geom* geom, int64_t* Ho, int64_t* Wo) {
    DcnGeom g;
    const int rc = make_geom(geom, g);
    if (rc) return rc;
    MTP_CHECK_ARG(Ho && Wo);
    *Ho = g.Ho;
    *Wo = g.Wo;
    return 0;
}

extern "C" int mtp_dcnv3_fwd(const void* input, const void* offset, const void* mask, void* output, int dtype, const mtp_dcnv3_geom* geom, mtp_stream_t stream) {
    DcnGeom g;
    const int rc = make_geom(geom, g);
    if (rc) return rc;
    MTP_CHECK_ARG(input && offset && mask && output);
    if (dtype == MTP_F64) {
        const int64_t total = (int64_t)g.N * g.Ho * g.Wo * g.G * g.GC;
        hipLaunchKernelGGL(dcnv3_f64_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const double*)input, (const double*)offset,
                           (const double*)mask, (const double*)nullptr, (double*)output, (double*)nullptr, (double*)nullptr, (double*)nullptr, g, total);
        return mtp_launch_status();
    }
    MTP_CHECK_ARG(dtype == MTP_F32 || dtype == MTP_BF16);
    if ((int64_t)g.N * g.Ho * g.Wo * g.G * g.GC >= ((int64_t)1 << 32) - 256) return MTP_ERR_UNSUPPORTED;   // one lane per element at most
    hipStream_t s = (hipStream_t)stream;
    return dtype == MTP_F32 ? launch_fwd<float>(input, offset, mask, output, g, s) : launch_fwd<bf16_t>(input, offset, mask, output, g, s);
}

extern "C" int mtp_dcnv3_bwd_act(const void* input, const void* offset, const void* mask, const void* grad_output, int dtype, float* grad_input, float* grad_offset,
                                 float* grad_mask, void* grad_offset_act, int64_t act_ld, const mtp_dcnv3_geom* geom, mtp_stream_t stream) {
    DcnGeom g;
    const int rc = make_geom(geom, g);
    if (rc) return rc;
    MTP_CHECK_ARG(input && offset && mask && grad_output && grad_input && grad_offset && grad_mask && grad_offset_act);
    MTP_CHECK_ARG(dtype == MTP_F32 || dtype == MTP_BF16);
    MTP_CHECK_ARG(act_ld >= (int64_t)g.G * g.P * 2 && act_ld < ((int64_t)1 << 30));
    if ((int64_t)g.N * g.Ho * g.Wo * g.G * g.GC >= ((int64_t)1 << 32) - 256) return MTP_ERR_UNSUPPORTED;
    g.goff_act = grad_offset_act;
    g.goff_act_ld = (int)act_ld;
    hipStream_t s = (hipStream_t)stream;
    return dtype == MTP_F32 ? launch_bwd<float>(input, offset, mask, grad_output, grad_input, grad_offset, grad_mask, g, s)
                            : launch_bwd<bf16_t>(input, offset, mask, grad_output, grad_input, grad_offset, grad_mask, g, s);
}

extern "C" int mtp_dcnv3_bwd(const void* input, const void* offset, const void* mask, const void* grad_output, int dtype, float* grad_input, float* grad_offset,
                             float* grad_mask, const mtp_dcnv3_geom* geom, mtp_stream_t stream) {
    DcnGeom g;
    const int rc = make_geom(geom, g);
    if (rc) return rc;
    MTP_CHECK_ARG(input && offset && mask && grad_output && grad_input && grad_offset && grad_mask);
    MTP_CHECK_ARG(dtype == MTP_F32 || dtype == MTP_BF16 || dtype == MTP_F64);
    if ((int64_t)g.N * g.Ho * g.Wo * g.G * g.GC >= ((int64_t)1 << 32) - 256) return MTP_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == MTP_F64) {      // (the three gradient pointers are double buffers here, see mtp_hip.h)
        const int64_t items = (int64_t)g.N * g.Ho * g.Wo * g.G, total = items * g.GC;
        hipError_t e = hipMemsetAsync(grad_input, 0, sizeof(double) * (size_t)g.N * g.H * g.W * g.G * g.GC, s);
        if (e == hipSuccess) e = hipMemsetAsync(grad_offset, 0, sizeof(double) * (size_t)items * 2 * g.P, s);
        if (e == hipSuccess) e = hipMemsetAsync(grad_mask, 0, sizeof(double) * (size_t)items * g.P, s);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(dcnv3_f64_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const double*)input, (const double*)offset, (const double*)mask,
                           (const double*)grad_output, (double*)nullptr, reinterpret_cast<double*>(grad_input), reinterpret_cast<double*>(grad_offset),
                           reinterpret_cast<double*>(grad_mask), g, total);
        return mtp_launch_status();
    }
    return dtype == MTP_F32 ? launch_bwd<float>(input, offset, mask, grad_output, grad_input, grad_offset, grad_mask, g, s)
                            : launch_bwd<bf16_t>(input, offset, mask, grad_output, grad_input, grad_offset, grad_mask, g, s);
}
