// Which CUs does a stream created with hipExtStreamCreateWithCUMask use on gfx950 (SPX mode: one device = 8 XCCs x 32 CUs)?
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/cu_mask_probe.hip -o gpurun_out/cu_mask_probe ; run under `timeout 60`.
// For every mask: workgroups per XCC and the number of distinct (XCC, SE, SH, CU) slots touched by a 2048-workgroup launch whose workgroups
// stay resident ~20 us each; then a bandwidth / MFMA pair on two masked streams (do two halves of the chip run independently?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d (%s) at %s:%d\n", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) where_kernel(int32_t* __restrict__ out, long long spin) {
    const long long t0 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x + 0] = (int32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20);
        out[2 * blockIdx.x + 1] = (int32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}

__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void __launch_bounds__(256) mfma_kernel(float* __restrict__ out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) out[0] = 1.f;
}

static int report(const char* name, const uint32_t* mask, int words, int32_t* dout, std::vector<int32_t>& h) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask);
    if (e != hipSuccess) { printf("%-34s create failed: %s\n", name, hipGetErrorString(e)); return 0; }
    const int blocks = 2048;
    CK(hipMemsetAsync(dout, 0xff, blocks * 8, s));
    hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(256), 0, s, dout, 40000ll);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), dout, blocks * 8, hipMemcpyDeviceToHost));
    int per_xcc[16] = {0};
    std::set<int> slots;
    std::set<int> per_xcc_slots[16];
    for (int b = 0; b < blocks; ++b) {
        int x = h[2 * b] & 15, hw = h[2 * b + 1];
        int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[x]++;
        slots.insert((x << 12) | (se << 8) | (sh << 4) | cu);
        per_xcc_slots[x].insert((se << 8) | (sh << 4) | cu);
    }
    uint32_t got[8] = {0};
    hipExtStreamGetCUMask(s, 8, got);
    printf("%-34s CUs touched %3zu | workgroups per XCC:", name, slots.size());
    for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
    printf(" | CUs per XCC:");
    for (int x = 0; x < 8; ++x) printf(" %2zu", per_xcc_slots[x].size());
    printf(" | getmask %08x %08x ..\n", got[0], got[1]);
    // the (se, cu) slots of XCC 0, to see which physical CUs a "half" is
    printf("    XCC0 slots (se.cu):");
    for (int v : per_xcc_slots[0]) printf(" %d.%d", v >> 8, v & 15);
    printf("\n");
    fflush(stdout);
    CK(hipStreamDestroy(s));
    return 0;
}

int main(int argc, char** argv) {
    const bool risky = argc > 1 && !strcmp(argv[1], "risky");
    int32_t* dout;
    CK(hipMalloc(&dout, 2048 * 8));
    std::vector<int32_t> h(2048 * 2);
    uint32_t m[8];
    auto fill = [&](auto pred) { memset(m, 0, sizeof(m)); for (int i = 0; i < 256; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); };
    fill([](int i) { return true; });
    if (report("all 256 bits", m, 8, dout, h)) return 1;
    fill([](int i) { return (i / 8) % 2 == 0; });
    if (report("(i/8) even  [16 CUs of every XCC?]", m, 8, dout, h)) return 1;
    fill([](int i) { return (i / 8) % 2 == 1; });
    if (report("(i/8) odd", m, 8, dout, h)) return 1;
    fill([](int i) { return (i / 8) < 16; });
    if (report("(i/8) < 16  [bits 0..127]", m, 8, dout, h)) return 1;
    fill([](int i) { return (i / 8) >= 16; });
    if (report("(i/8) >= 16 [bits 128..255]", m, 8, dout, h)) return 1;
    fill([](int i) { return (i / 8) < 8; });
    if (report("(i/8) < 8   [bits 0..63]", m, 8, dout, h)) return 1;
    fill([](int i) { return i % 8 != 7 || (i / 8) == 0; });
    if (report("XCC 7 reduced to one CU", m, 8, dout, h)) return 1;
    if (risky) {
        fill([](int i) { return i % 8 < 4; });
        if (report("i%8 < 4   [XCCs 0-3 only?]", m, 8, dout, h)) return 1;
        fill([](int i) { return i % 8 >= 4; });
        if (report("i%8 >= 4  [XCCs 4-7 only?]", m, 8, dout, h)) return 1;
    }
    // ---- do two masked streams run side by side?  copy (HBM-bound) on one half, MFMA loop on the other
    const size_t n = (size_t)1 << 26;     // 1 GiB of float4 each way
    float4 *a, *b;
    float* o;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 1, n * 16));
    hipStream_t sa, sb, full0, full1;
    fill([](int i) { return (i / 8) % 2 == 0; });
    CK(hipExtStreamCreateWithCUMask(&sa, 8, m));
    fill([](int i) { return (i / 8) % 2 == 1; });
    CK(hipExtStreamCreateWithCUMask(&sb, 8, m));
    CK(hipStreamCreateWithFlags(&full0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&full1, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2, e3;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
    auto run = [&](const char* what, hipStream_t s_copy, hipStream_t s_mfma, bool do_copy, bool do_mfma, int copy_blocks, int mfma_blocks) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            if (do_copy) { hipEventRecord(e0, s_copy); hipLaunchKernelGGL(copy_kernel, dim3(copy_blocks), dim3(256), 0, s_copy, a, b, n); hipEventRecord(e1, s_copy); }
            if (do_mfma) { hipEventRecord(e2, s_mfma); hipLaunchKernelGGL(mfma_kernel, dim3(mfma_blocks), dim3(256), 0, s_mfma, o, 40000); hipEventRecord(e3, s_mfma); }
            hipDeviceSynchronize();
        }
        float tc = 0, tm = 0;
        if (do_copy) hipEventElapsedTime(&tc, e0, e1);
        if (do_mfma) hipEventElapsedTime(&tm, e2, e3);
        printf("%-58s copy %7.3f ms (%5.2f TB/s)   mfma %7.3f ms (%6.0f TF/s)\n", what, tc, tc > 0 ? 2.0 * n * 16 / tc / 1e9 : 0.0, tm,
               tm > 0 ? (double)mfma_blocks * 4 * 40000 * 4 * 2.0 * 16 * 16 * 32 / tm / 1e9 : 0.0);
        fflush(stdout);
    };
    run("copy alone, whole chip", full0, full1, true, false, 2048, 0);
    run("mfma alone, whole chip (2048 wgs)", full0, full1, false, true, 0, 2048);
    run("copy alone, masked half", sa, sb, true, false, 1024, 0);
    run("mfma alone, masked half (1024 wgs)", sa, sb, false, true, 0, 1024);
    run("copy + mfma, two unmasked streams (2048 + 2048 wgs)", full0, full1, true, true, 2048, 2048);
    run("copy on half A + mfma on half B (1024 + 1024 wgs)", sa, sb, true, true, 1024, 1024);
    printf("done\n");
    return 0;
}
