// Probe: what does one LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB) cost the ISSUING wave inside an MFMA-paced loop?
// One wave per SIMD (256 threads), 14 x v_mfma_f32_32x32x2_f32 per k-step (896 cycles of pipe), optional ds_reads (one per MFMA),
// optional DMA pieces.  Prints cycles per k-step for each variant.   hipcc --offload-arch=gfx950 -O3 dma_issue_probe.hip -o dma_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
constexpr int NT = 14, STEPS = 8, REPS = 512;

// VAR: 0 MFMA only; 1 + ds_read per MFMA; 2 + one DMA piece per 2 k-steps (builtin); 3 = 2 but DMA dwordx1 (256 B);
//      4 = 2 but the piece is issued by exec-masked 16 lanes only (256 B); 5 = 2 but global source is ONE cache line (all lanes same 16 B)
//      6 = 1 + one global_load_dwordx4 to VGPRs (no LDS) per 2 k-steps; 7 = DMA only, no ds_reads
template <int VAR>
__global__ __launch_bounds__(256, 1) void probe(const float* __restrict__ x, float* __restrict__ out, long long* __restrict__ cyc, int nbytes) {
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 256) smem[i] = (float)(i & 7);
    __syncthreads();
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, nbytes, 0x00020000);
    const float* bp = smem + lane;
    float b0[NT], b1[NT], sink = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) { b0[t] = bp[t * 64]; b1[t] = bp[t * 64 + 32]; }
    const unsigned vbase = (unsigned)(blockIdx.x * 65536 + wave * 16384 + lane * (VAR == 5 ? 0 : 16));
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
        for (int s = 0; s < STEPS; s += 2) {
            if (VAR == 2 || VAR == 5 || VAR == 7)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + 8192 + wave * 256), 16, vbase + (unsigned)(s * 1024), 0, 0, 0);
            if (VAR == 3)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + 8192 + wave * 256), 4, vbase + (unsigned)(s * 1024), 0, 0, 0);
            if (VAR == 4 && lane < 16)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + 8192 + wave * 256), 16, vbase + (unsigned)(s * 1024), 0, 0, 0);
            if (VAR == 6) {
                auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, vbase + (unsigned)(s * 1024), 0, 0);
                sink += __builtin_bit_cast(float, v[0]);
            }
            if (VAR >= 1 && VAR != 7) {
#pragma unroll
                for (int t = 0; t < NT; ++t) b1[t] = bp[((s + 1) * NT + t) * 64 % 8000];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[0], b0[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            if (VAR >= 1 && VAR != 7) {
#pragma unroll
                for (int t = 0; t < NT; ++t) b0[t] = bp[((s + 2) * NT + t) * 64 % 8000];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[0], b1[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_waitcnt(0x0f70);
    float s = sink;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * 256 + tid] = s + smem[8192 + tid];
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int VAR> static void run(const float* x, float* out, long long* cyc, int nbytes, const char* what) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int i = 0; i < 3; ++i) probe<VAR><<<256, 256, 65536>>>(x, out, cyc, nbytes);
    hipDeviceSynchronize();
    std::vector<long long> h(1024);
    hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost);
    std::vector<long long> v(h.begin(), h.end());
    std::sort(v.begin(), v.end());
    printf("%-86s %8.1f cycles per k-step (floor %d)\n", what, (double)v[512] / (REPS * STEPS), NT * 64);
}
int main() {
    const int nbytes = 256 * 65536 + 65536 + 1024 * 64;
    float *x, *out; long long* cyc;
    hipMalloc(&x, nbytes); hipMemset(x, 0, nbytes); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    run<0>(x, out, cyc, nbytes, "MFMA only");
    run<1>(x, out, cyc, nbytes, "+ one ds_read_b32 per MFMA");
    run<2>(x, out, cyc, nbytes, "+ one 1-KiB LDS-DMA piece per 2 k-steps (per wave)");
    run<7>(x, out, cyc, nbytes, "MFMA + the DMA piece, no ds_reads");
    run<3>(x, out, cyc, nbytes, "... piece = buffer_load_dword lds (256 B)");
    run<4>(x, out, cyc, nbytes, "... piece issued by 16 lanes only (256 B, exec-masked)");
    run<5>(x, out, cyc, nbytes, "... all 64 lanes read the same 16 B");
    run<6>(x, out, cyc, nbytes, "+ ds_reads + one buffer_load_dwordx4 to VGPRs per 2 k-steps (no LDS write)");
    return 0;
}
