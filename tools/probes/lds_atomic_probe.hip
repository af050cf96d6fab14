// Probe: cost of LDS float atomics (ds_add_f32, no return) on gfx950 next to plain ds_write / ds_read, for the access patterns a
// tile-accumulating DCNv3 backward would produce.  Build + run on an MI355X:
//   hipcc -O2 -munsafe-fp-atomics --offload-arch=gfx950 tools/probes/lds_atomic_probe.hip -o tools/_abl/lds_atomic_probe && tools/_abl/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, const int* idx, int iters, long long* cyc) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = 0.f;
    __syncthreads();
    const int base = idx[threadIdx.x];
    const long long t0 = clock64();
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int a = (base + it * 1031) & 16383;
        if (MODE == 0) atomicAdd(&lds[a], 1.0f);
        else if (MODE == 1) lds[a] = (float)it;
        else acc += lds[a];
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x] + acc;
}
int main() {
    const int iters = 4096;
    int h[256];
    float* out; int* d; long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&d, 1024); hipMalloc(&cyc, 256 * 8);
    const char* names[3] = {"lane-linear (conflict-free)", "16 channels x 4 random pixels per wave", "all lanes same address"};
    for (int pat = 0; pat < 3; ++pat) {
        for (int l = 0; l < 256; ++l) h[l] = pat == 0 ? l : pat == 1 ? ((l >> 4) * 977 * 16 + (l & 15)) & 16383 : 5;
        hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; ++mode) {
            long long c[256];
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) probe<0><<<256, 256>>>(out, d, iters, cyc);
                if (mode == 1) probe<1><<<256, 256>>>(out, d, iters, cyc);
                if (mode == 2) probe<2><<<256, 256>>>(out, d, iters, cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(c, cyc, 256 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += c[i];
            printf("%-42s %-10s %.1f clk per wave-instruction (4 waves per workgroup, 1 workgroup per CU)\n", names[pat], mode == 0 ? "ds_add_f32" : mode == 1 ? "ds_write" : "ds_read",
                   s / 256 / iters / 1.0);
        }
    }
    return 0;
}
