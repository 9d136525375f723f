// Inner-loop study: MFMA f32 32x32x2 fed from LDS.  Variants:
//  0: pure MFMA (registers)           1: per k-step {1+NT ds_read_b32 -> NT MFMA}, compiler-scheduled
//  2: same, reads of step s+1 issued before the MFMAs of step s (explicit register double buffer + sched_barrier)
//  3: fwd-conv style: ds_read_b128 x3 per 8 MFMA (2 acc), prefetched one group ahead
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 7;
__device__ int g_random;
template <int VAR>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    extern __shared__ float lds[];
    // DATA=1: full-range pseudo-random operands (hash) -> realistic toggle rate; DATA=0: smooth small values
    for (int i = threadIdx.x; i < 16384; i += 256) {
        unsigned u = (unsigned)i * 2654435761u + blockIdx.x * 40503u; u ^= u >> 15; u *= 2246822519u; u ^= u >> 13;
        lds[i] = g_random ? ((int)(u & 0xFFFFFF) - 0x800000) * (1.0f / 0x800000) : 1e-3f * (i & 255);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    const float* ap0 = lds + l31;
    const float* bp0 = lds + 4096 + l31;
    int off[NT];
    for (int t = 0; t < NT; ++t) off[t] = (t * 37 + (blockIdx.x & 3)) * 32;
    if (VAR == 0) {
        float a = lds[lane], b = lds[lane + 64];
        for (int it = 0; it < iters * 64; ++it)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    } else if (VAR == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll 2
            for (int s = 0; s < 64; ++s) {
                const int v = 2 * s + h;
                const float a = ap0[v * 32];
                const float* bp = bp0 + (v & 63) * 32;
                float bv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t] = bp[off[t]];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc[t], 0, 0, 0);
            }
    } else if (VAR == 2) {
        for (int it = 0; it < iters; ++it) {
            float a = ap0[h * 32], bv[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) bv[t] = (bp0 + h * 32)[off[t]];
#pragma unroll 2
            for (int s = 0; s < 64; ++s) {
                const int vn = 2 * ((s + 1) & 63) + h;
                const float an = ap0[vn * 32];
                const float* bpn = bp0 + (vn & 63) * 32;
                float bn[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bn[t] = bpn[off[t]];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc[t], 0, 0, 0);
                a = an;
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t] = bn[t];
            }
        }
    } else if (VAR == 3) {
        const float* ap = lds + l31 * 36 + 4 * h;
        const float* bp = lds + 8192 + l31 * 36 + 4 * h;
        for (int it = 0; it < iters * 14; ++it) {     // 14 "taps" x 32 MFMA ~ same MFMA count as 64 x 7
            float4 a = *(const float4*)ap, b0 = *(const float4*)bp, b1 = *(const float4*)(bp + 32 * 36);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 an, b0n, b1n;
                if (q < 3) { an = *(const float4*)(ap + 8 * (q + 1)); b0n = *(const float4*)(bp + 8 * (q + 1)); b1n = *(const float4*)(bp + 32 * 36 + 8 * (q + 1)); }
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
                if (q < 3) { a = an; b0 = b0n; b1 = b1n; }
            }
        }
    }
    if (VAR >= 4 && VAR <= 7) {
        // conv-fwd tap structure: per "tap" 4 q-groups x 8 MFMA; fragments of q+1 prefetched inside the tap, but the
        // first group of each tap is read cold (VAR 4); + a workgroup barrier per tap (VAR 5); + a weight-panel
        // ds_write_b128 pair before the barrier (VAR 6); VAR 7 = VAR 4 with a per-tap moving address (VALU add)
        const float* ap = lds + l31 * 36 + 4 * h;
        const float* bp = lds + 8192 + l31 * 36 + 4 * h;
        float4 wv = make_float4(1e-3f * lane, 0.f, 1.f, 2.f);
        for (int it = 0; it < iters * 14; ++it) {
            const float* at = ap + ((VAR == 7) ? (it % 27) * 36 : 0);
            float4 a = *(const float4*)at, b0 = *(const float4*)bp, b1 = *(const float4*)(bp + 32 * 36);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 an, b0n, b1n;
                if (q < 3) { an = *(const float4*)(at + 8 * (q + 1)); b0n = *(const float4*)(bp + 8 * (q + 1)); b1n = *(const float4*)(bp + 32 * 36 + 8 * (q + 1)); }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
                if (q < 3) { a = an; b0 = b0n; b1 = b1n; }
            }
            if (VAR == 6) {
                *(float4*)(lds + 12288 + threadIdx.x * 4) = wv;
                *(float4*)(lds + 12288 + 1024 + threadIdx.x * 4) = wv;
            }
            if (VAR == 5 || VAR == 6) __syncthreads();
        }
    }
    if (VAR == 8) {
        // weights global -> VGPR (fragment-major panels: [tap][r 0..7][lane][4], 8 KB per tap, 54 taps cycling through
        // a 442 KB L2-resident buffer), prefetched one tap ahead; A fragments from LDS; no barrier, no weight LDS traffic
        const float* ap = lds + l31 * 36 + 4 * h;
        const float4* wg = reinterpret_cast<const float4*>(out + (1 << 20));       // scratch region of the out buffer
        float4 w[8], wn[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) w[r] = wg[r * 64 + lane];
        for (int it = 0; it < iters * 14; ++it) {
            const int tn = (it + 1) % 54;
#pragma unroll
            for (int r = 0; r < 8; ++r) wn[r] = wg[(tn * 8 + r) * 64 + lane];
            const float* at = ap + (it % 27) * 36;
            float4 a = *(const float4*)at;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 an;
                if (q < 3) an = *(const float4*)(at + 8 * (q + 1));
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w[2 * q].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w[2 * q + 1].x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w[2 * q].y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w[2 * q + 1].y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w[2 * q].z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w[2 * q + 1].z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w[2 * q].w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w[2 * q + 1].w, acc[1], 0, 0, 0);
                if (q < 3) a = an;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) w[r] = wn[r];
        }
    }
    float s = 0.f;
    for (int t = 0; t < NT; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int VAR> void run(const char* name, float* out, int wgs) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 600;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(s);
        hipLaunchKernelGGL(k<VAR>, dim3(wgs), dim3(256), 65536, 0, out, iters);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        const double nm = VAR >= 3 ? (double)iters * 14 * 32 : (double)iters * 64 * NT;
        if (rep == 5) printf("%-44s %d WG/CU: %.3f ms  %.1f TFLOP/s\n", name, wgs / 256, ms, wgs * 4 * nm * 4096.0 / ms / 1e9);
    }
}
// 8-wave workgroup (1 per CU): per stage = 3 taps x 32 MFMA per wave, then 4 ds_write_b128 of the next weight stage + barrier
__global__ __launch_bounds__(512, 1) void k9(float* out, int iters, int tapsPerStage) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 32768; i += 512) {
        unsigned u = (unsigned)i * 2654435761u + blockIdx.x * 40503u; u ^= u >> 15; u *= 2246822519u; u ^= u >> 13;
        lds[i] = ((int)(u & 0xFFFFFF) - 0x800000) * (1.0f / 0x800000);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    f32x16 acc[2];
    for (int t = 0; t < 2; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    const float* ap = lds + wave * 1152 + l31 * 36 + 4 * h;
    const float* bp = lds + 16384 + l31 * 36 + 4 * h;
    float4 wv = make_float4(1e-3f * lane, 0.f, 1.f, 2.f);
    const int stages = iters * 14 / tapsPerStage;
    for (int st = 0; st < stages; ++st) {
        for (int tp = 0; tp < tapsPerStage; ++tp) {
            const float* at = ap + ((st * tapsPerStage + tp) % 27) * 36;
            const float* bt = bp + tp * 2304;
            float4 a = *(const float4*)at, b0 = *(const float4*)bt, b1 = *(const float4*)(bt + 32 * 36);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 an, b0n, b1n;
                if (q < 3) { an = *(const float4*)(at + 8 * (q + 1)); b0n = *(const float4*)(bt + 8 * (q + 1)); b1n = *(const float4*)(bt + 32 * 36 + 8 * (q + 1)); }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc[1], 0, 0, 0);
                if (q < 3) { a = an; b0 = b0n; b1 = b1n; }
            }
        }
        // next stage's weights: tapsPerStage * 8 KB / 512 threads = tapsPerStage b128 writes per thread (other buffer)
        for (int r = 0; r < tapsPerStage; ++r) *(float4*)(lds + 24576 + ((st & 1) * 3 + r) * 2048 % 8192 + threadIdx.x * 4) = wv;
        __syncthreads();
    }
    float s = 0.f;
    for (int t = 0; t < 2; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
void run9(float* out, int tps) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int iters = 600;
    hipFuncSetAttribute((const void*)k9, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(s);
        hipLaunchKernelGGL(k9, dim3(256), dim3(512), 140 * 1024, 0, out, iters, tps);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        const double nm = (double)(iters * 14 / tps) * tps * 32;
        if (rep == 5) printf("8-wave WG, barrier + W write every %d taps        1 WG/CU: %.3f ms  %.1f TFLOP/s\n", tps, ms, 256 * 8 * nm * 4096.0 / ms / 1e9);
    }
}
int main(int argc, char** argv) {
    float* out; hipMalloc(&out, (4096 * 256 + (1 << 20)) * 4 + (1 << 21)); hipMemset(out, 0, (4096 * 256 + (1 << 20)) * 4 + (1 << 21));
    int rnd = argc > 1 ? atoi(argv[1]) : 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_random), &rnd, sizeof(int));
    printf("operands: %s\n", rnd ? "random full-range" : "smooth small");
    for (int w = 256; w <= 512; w += 256) {
        run<0>("pure MFMA, 7 acc", out, w);
        run<1>("LDS b32 x8 -> 7 MFMA (compiler order)", out, w);
        run<2>("LDS b32 x8 prefetched one step ahead", out, w);
        run<3>("LDS b128 x3 / 8 MFMA prefetched (fwd style)", out, w);
        run<4>("  + cold first group per tap", out, w);
        run<7>("  + cold first group, moving address", out, w);
        run<5>("  + cold first group + barrier per tap", out, w);
        run<6>("  + cold group + W ds_write + barrier per tap", out, w);
        run<8>("A from LDS, W global->VGPR prefetched, no barrier", out, w);
    }
    run9(out, 1); run9(out, 3); run9(out, 9);
    return 0;
}
