// Probe: buffer_load_dwordx4 ... lds (LDS-DMA) semantics on gfx950.
//   (1) lane i writes 16 B at M0-base + 16*i;  (2) an out-of-range buffer offset writes ZEROS to LDS (not "no write");
//   (3) EXEC-masked lanes leave their 16 B untouched.
// hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip -o lds_dma_probe && ./lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(const float* __restrict__ x, float* __restrict__ out, int nbytes) {
    __shared__ __attribute__((aligned(16))) float smem[3 * 256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 3 * 256; i += 64) smem[i] = -7.f;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, nbytes, 0x00020000);
    // (1)+(2): every lane loads; lanes 40..47 use an out-of-range offset
    unsigned voff = (lane >= 40 && lane < 48) ? 0x80000000u : (unsigned)lane * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
    // (3): only even lanes active, reversed source
    if ((lane & 1) == 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 256), 16, (unsigned)(63 - lane) * 16u, 0, 0, 0);
    // range check at the END of the buffer: offsets within 16 B of the end
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 512), 16, (unsigned)nbytes - 1024u + (unsigned)lane * 16u + 8u, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0)
    __syncthreads();
    for (int i = lane; i < 3 * 256; i += 64) out[i] = smem[i];
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *x, *o;
    hipMalloc(&x, n * 4); hipMalloc(&o, 768 * 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(x, o, n * 4);
    std::vector<float> r(768);
    hipMemcpy(r.data(), o, 768 * 4, hipMemcpyDeviceToHost);
    int ok1 = 1, ok2 = 1, ok3 = 1;
    for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) {
        const float v = r[l * 4 + k];
        if (l >= 40 && l < 48) { if (v != 0.f) ok2 = 0; } else if (v != (float)(l * 4 + k)) ok1 = 0;
        const float w = r[256 + l * 4 + k];
        if ((l & 1) == 0) { if (w != (float)((63 - l) * 4 + k)) ok3 = 0; } else if (w != -7.f) ok3 = 0;
    }
    printf("lane-linear placement: %s\nout-of-range -> zeros: %s (lane 40: %g %g %g %g)\nexec-masked lanes untouched: %s (lane 1: %g)\n",
           ok1 ? "yes" : "NO", ok2 ? "yes" : "NO", r[160], r[161], r[162], r[163], ok3 ? "yes" : "NO", r[256 + 4]);
    printf("tail (offset nbytes-1024+8+16*lane), last lanes: ");
    for (int l = 60; l < 64; ++l) printf("[%g %g %g %g] ", r[512 + l * 4], r[512 + l * 4 + 1], r[512 + l * 4 + 2], r[512 + l * 4 + 3]);
    printf("\n");
    return (ok1 && ok2 && ok3) ? 0 : 1;
}
