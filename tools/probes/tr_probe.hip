// Probe of the gfx950 LDS transpose read (ds_read_b64_tr_b16): which lane gets which element.  Build + run on an MI355X:
//   hipcc -O2 --offload-arch=gfx950 tools/probes/tr_probe.hip -o tools/_abl/tr_probe && tools/_abl/tr_probe
// Result (pattern 0): in each 16-lane group, lane i supplies the address of row i>>2, columns 4(i&3)..+3 of a [4][16] b16 block and
// receives column i, rows 0..3 -- the layout gemm_tn_tr_kernel (mtp_amd/csrc/gemm.hip) is built on.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr_elems, short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    int a = addr_elems[threadIdx.x];
    v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + a));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    int h[64]; short o[256];
    int *d; short* dout;
    hipMalloc(&d, 256); hipMalloc(&dout, 512);
    // pattern A: lane i in 16-group g: row = 8g + i/4 (row stride 128 elems), col = 4*(i%4)
    for (int pat = 0; pat < 2; ++pat) {
        for (int l = 0; l < 64; ++l) { int i = l & 15, g = l >> 4;
            h[l] = pat == 0 ? (8 * g + i / 4) * 128 + 4 * (i % 4) : (8 * g + i % 4) * 128 + 4 * (i / 4); }
        hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(d, dout);
        hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) { printf("l%2d:", l); for (int j = 0; j < 4; ++j) printf(" r%d.c%d", o[l*4+j] / 128, o[l*4+j] % 128); printf("\n"); }
    }
    return 0;
}
