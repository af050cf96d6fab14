// Probe: lane exchanges without the LDS crossbar on gfx950.  hipcc turns every __shfl_xor into ds_bpermute_b32 (an LDS instruction: issue slot + round trip);
// the xor-1 / 2 / 4 / 8 partners of a 16-lane butterfly can be reached with DPP modifiers instead:
//   xor 1 = quad_perm [1,0,3,2] (0xB1), xor 2 = quad_perm [2,3,0,1] (0x4E), xor 3 = quad_perm [3,2,1,0] (0x1B),
//   xor 7 = row_half_mirror (0x141), xor 15 = row_mirror (0x140); xor 4 = xor 7 o xor 3, xor 8 = xor 15 o xor 7.
// Checks the mapping lane by lane, the 16-lane butterfly sum against the __shfl_xor form (bit-identical for the mirror order the sums take),
// and times a dependent chain of butterflies both ways (clock64 around the loop of one wave per workgroup x 4 waves).
// Build + run on an MI355X:  hipcc -O2 --offload-arch=gfx950 tools/probes/dpp_probe.hip -o tools/_abl/dpp_probe && tools/_abl/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float xor1(float v) { return dpp<0xB1>(v); }
__device__ __forceinline__ float xor2(float v) { return dpp<0x4E>(v); }
__device__ __forceinline__ float xor4(float v) { return dpp<0x1B>(dpp<0x141>(v)); }
__device__ __forceinline__ float xor8(float v) { return dpp<0x141>(dpp<0x140>(v)); }
// butterfly sum over the 16 lanes of a row: after xor 1 and xor 2 every lane of a quad holds the quad's sum, so the mirror partners (other quad of
// the half row, other half of the row) serve as well as the xor-4 / xor-8 partners, with one DPP operation each
__device__ __forceinline__ float sum16_dpp(float v) {
    v += xor1(v);
    v += xor2(v);
    v += dpp<0x141>(v);
    v += dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float sum16_shfl(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

__global__ void check(const float* in, float* out) {
    const int l = threadIdx.x;
    const float v = in[l];
    out[0 * 64 + l] = xor1(v);  out[1 * 64 + l] = __shfl_xor(v, 1, 64);
    out[2 * 64 + l] = xor2(v);  out[3 * 64 + l] = __shfl_xor(v, 2, 64);
    out[4 * 64 + l] = xor4(v);  out[5 * 64 + l] = __shfl_xor(v, 4, 64);
    out[6 * 64 + l] = xor8(v);  out[7 * 64 + l] = __shfl_xor(v, 8, 64);
    out[8 * 64 + l] = sum16_dpp(v); out[9 * 64 + l] = sum16_shfl(v);
}

template <int MODE>
__global__ __launch_bounds__(256) void chain(float* out, int iters, long long* cyc) {
    float v = (float)(threadIdx.x & 63) * 0.001f + 1.0f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const float s = MODE ? sum16_dpp(v) : sum16_shfl(v);
        v = v * 0.5f + s * 0.03125f;      // dependent on the butterfly's result
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
    float h[64], r[10 * 64];
    for (int i = 0; i < 64; ++i) h[i] = 1.0f + 0.37f * (float)((i * 29) % 64);
    float *din, *dout; long long* cyc;
    hipMalloc(&din, 256); hipMalloc(&dout, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
    check<<<1, 64>>>(din, dout);
    hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
    const char* nm[5] = {"xor 1", "xor 2", "xor 4", "xor 8", "16-lane sum"};
    int bad_total = 0;
    for (int k = 0; k < 5; ++k) {
        int bad = 0;
        for (int l = 0; l < 64; ++l) bad += (r[2 * k * 64 + l] != r[(2 * k + 1) * 64 + l]);
        printf("%-12s DPP vs __shfl_xor: %d of 64 lanes differ%s\n", nm[k], bad, k == 4 && bad ? "  (sum order differs: compare to rounding)" : "");
        if (k < 4) bad_total += bad;
    }
    if (r[8 * 64] != r[9 * 64]) printf("16-lane sum lane 0: dpp %.9g shfl %.9g\n", r[8 * 64], r[9 * 64]);
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        long long c[1024];
        for (int rep = 0; rep < 2; ++rep) {
            if (mode) chain<1><<<256, 256>>>(dout, iters, cyc); else chain<0><<<256, 256>>>(dout, iters, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 1024; ++i) s += (double)c[i];
        printf("dependent 16-lane butterfly sum + 1 fma, %s: %.1f clk per iteration (4 waves per workgroup, 1 workgroup per CU)\n", mode ? "DPP      " : "__shfl_xor", s / 1024 / iters);
    }
    return bad_total != 0;
}
