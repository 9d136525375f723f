// Pointwise convs / Linear layers with FEW input channels and many output channels (attention to_q 64 -> 512, feed-forward 64 -> 128:
// /root/reference/imagen_video.py:410-470, 984-1002): y[rows][Cout] = x[rows][64] W^T.  conv1x1_fwd_kernel (conv_mfma.hip) gives
// every (128-row tile, 64-channel block) its own workgroup: with K = 64 that is 64 MFMAs per wave behind a cold global load, x is
// fetched Cout / 64 times, and nothing overlaps the 32 stores per lane of the epilogue -- 74 TFLOP/s and 2.6 TB/s on 64 -> 512, a
// product that is bound by its OUTPUT stream (x 67 MB, y 537 MB at 262144 rows).  Here (216 us = 2.8 TB/s on that shape, 246 before):
//   * persistent workgroups (one per CU) walk row tiles of 128; the tile's x rows (32 KB, contiguous) arrive by LDS-DMA while the
//     previous tile computes (double-buffered), are read into registers ONCE (the A fragments of a wave's 32 rows: 32 VGPRs) and
//     serve every 64-channel block of the output;
//   * the weight panel of the next channel block (16 KB) is prefetched global -> registers during the current block's MFMAs and lands
//     in the other padded LDS buffer behind them: one barrier per block, the stores of block j overlap the MFMAs of block j + 1
//     of the other waves;
//   * same packed weights, fragment order, k order and epilogue as conv1x1_fwd_kernel: bit-identical results.
#include "common.h"
#include "conv_pw.h"

namespace diqt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4p __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_pw;

constexpr int PW_K = 64, PW_TR = 128, PW_CK = 32, PW_ROW = PW_CK + 4;       // K, rows per tile, packed chunk width, padded weight row
constexpr int PW_AB = PW_TR * PW_K * 4;                                       // bytes of an x tile
constexpr int PW_WB = 2 * 64 * PW_ROW * 4;                                    // bytes of a weight panel (2 chunks x 64 co, padded rows)
constexpr unsigned PW_OOB = 0x80000000u, PW_OOB_C = 0x40000000u;

__global__ __launch_bounds__(256, 1) void conv1x1_k64_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                             const float* __restrict__ bias, const float* __restrict__ residual,
                                                             float* __restrict__ y, long long rows, int Cout, int CoutPad,
                                                             unsigned xBytes, unsigned yBytes) {
    extern __shared__ __attribute__((aligned(1024))) char smem_pw[];
    float* const Ws = reinterpret_cast<float*>(smem_pw + 2 * PW_AB);          // [2][2 chunks][64][PW_ROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int nNt = CoutPad / 64;
    const long long nTiles = (rows + PW_TR - 1) / PW_TR;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)xBytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)yBytes, 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual), 0, residual ? (int)yBytes : 0, 0x00020000);
    const unsigned ldsBase = (unsigned)(size_t)(lds_void_pw*)smem_pw;

    auto dma_tile = [&](long long t, int buf) __attribute__((always_inline)) {       // 32 KB = 32 one-KiB pieces, 8 per wave
        const unsigned long long base = (unsigned long long)t * PW_AB;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned p = (unsigned)(wave + 4 * j) * 1024u + (unsigned)lane * 16u;
            const unsigned long long off = base + p;
            const unsigned voff = (t < nTiles && off < xBytes) ? (unsigned)off : PW_OOB;      // rows past the end: zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_pw*)(size_t)(ldsBase + (unsigned)buf * PW_AB + (unsigned)(wave + 4 * j) * 1024u),
                                                     16, voff, 0, 0, 0);
        }
    };
    // this thread's pieces of a weight panel: rows (tid >> 3) and + 32 of the 64 co, channel quad (tid & 7), both chunks
    const int prow = tid >> 3, pc4 = (tid & 7) * 4;
    f32x4p rw[4];
    auto load_w = [&](int nt) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* wchunk = wp + ((size_t)c * CoutPad + (size_t)nt * 64) * PW_CK;
            rw[2 * c] = *reinterpret_cast<const f32x4p*>(wchunk + (size_t)prow * PW_CK + pc4);
            rw[2 * c + 1] = *reinterpret_cast<const f32x4p*>(wchunk + (size_t)(prow + 32) * PW_CK + pc4);
        }
    };
    auto store_w = [&](int buf) __attribute__((always_inline)) {
        float* wb = Ws + buf * (2 * 64 * PW_ROW);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            *reinterpret_cast<f32x4p*>(wb + (c * 64 + prow) * PW_ROW + pc4) = rw[2 * c];
            *reinterpret_cast<f32x4p*>(wb + (c * 64 + prow + 32) * PW_ROW + pc4) = rw[2 * c + 1];
        }
    };

    long long t = blockIdx.x;
    if (t >= nTiles) return;
    dma_tile(t, 0);
    load_w(0);
    store_w(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0): this wave's x pieces
    __syncthreads();
    int abuf = 0, wbuf = 0;
    for (; t < nTiles; t += gridDim.x) {
        dma_tile(t + gridDim.x, abuf ^ 1);                 // next tile of this workgroup (dead pieces when there is none)
        // A fragments of this wave's 32 rows, both chunks: a[c][q] = x[row][32 c + 8 q + 4 h .. + 3]  (a strided LDS read once per tile)
        f32x4p a[2][4];
        const float* ap = reinterpret_cast<const float*>(smem_pw + abuf * PW_AB) + (wave * 32 + l31) * PW_K + 4 * h;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[c][q] = *reinterpret_cast<const f32x4p*>(ap + c * PW_CK + 8 * q);
        const long long r0 = t * PW_TR;
        for (int nt = 0; nt < nNt; ++nt) {
            const int ntn = nt + 1 < nNt ? nt + 1 : 0;     // the panel the next block (or the next tile's first block) needs
            load_w(ntn);
            f32x16 acc0, acc1;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
            const float* bp = Ws + wbuf * (2 * 64 * PW_ROW) + l31 * PW_ROW + 4 * h;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4p b0 = *reinterpret_cast<const f32x4p*>(bp + c * 64 * PW_ROW + 8 * q);
                    const f32x4p b1 = *reinterpret_cast<const f32x4p*>(bp + (c * 64 + 32) * PW_ROW + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][q][e], b0[e], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][q][e], b1[e], acc1, 0, 0, 0);
                    }
                }
            store_w(wbuf ^ 1);
            // ---- epilogue of the block: D[row][col = co]; row = (r & 3) + 8 (r >> 2) + 4 h ----
            const int n0 = nt * 64, co0 = n0 + l31, co1 = n0 + 32 + l31;
            const float bias0 = (bias && co0 < Cout) ? bias[co0] : 0.f;
            const float bias1 = (bias && co1 < Cout) ? bias[co1] : 0.f;
            const unsigned c0 = co0 < Cout ? (unsigned)co0 * 4u : PW_OOB_C, c1 = co1 < Cout ? (unsigned)co1 * 4u : PW_OOB_C;
            float rr0[16], rr1[16];
            if (residual) {            // kernel-uniform: all 32 loads in flight before the first add
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = r0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const unsigned off = row < rows ? (unsigned)(row * Cout * 4) : PW_OOB;
                    rr0[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off + c0, 0, 0));
                    rr1[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, off + c1, 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = r0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const unsigned off = row < rows ? (unsigned)(row * Cout * 4) : PW_OOB;
                float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
                if (residual) { v0 += rr0[r]; v1 += rr1[r]; }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs_y, off + c0, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rs_y, off + c1, 0, 0);
            }
            // The other panel is complete (LDS writes: lgkmcnt) and everybody is done with this one.  NOT __syncthreads(): its
            // vmcnt(0) would wait for the 32 stores above to be acknowledged by memory, a ~1 us stall per 4096-cycle block.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            wbuf ^= 1;
        }
        // the next tile's x pieces were issued a whole tile (>= 64 stores) ago; loads and stores retire in order, so leaving the
        // last block's 32 stores in flight still proves the pieces have landed
        asm volatile("s_waitcnt vmcnt(32)\n\ts_barrier" ::: "memory");
        abuf ^= 1;
    }
}

bool pw64_ok(long long rows, int Cin, int Cout, const void* x, const void* packed, const void* y) {
    static const bool off = [] { const char* e = getenv("DIQT_NO_PW64"); return e && e[0] == '1'; }();
    if (off || Cin != PW_K || Cout < 256 || rows < 4096) return false;      // measured: 64 -> 512 216 vs 246 us, 64 -> 128 66 vs 61 us
    if ((unsigned long long)rows * Cin * 4ull >= (1ull << 31) || (unsigned long long)rows * Cout * 4ull >= (1ull << 31)) return false;
    return ((size_t)x & 15) == 0 && ((size_t)packed & 15) == 0 && ((size_t)y & 3) == 0;
}

int pw64_launch(const float* x, const float* packed, const float* bias, const float* residual, float* y, long long rows, int Cout,
                int CoutPad, void* stream) {
    const size_t lds = 2 * (size_t)PW_AB + 2 * (size_t)PW_WB;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_k64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd(1x1x1, K = 64): hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = true;
    }
    const long long nTiles = (rows + PW_TR - 1) / PW_TR;
    const unsigned grid = (unsigned)(nTiles < 256 ? nTiles : 256);
    hipLaunchKernelGGL(conv1x1_k64_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, x, packed, bias, residual, y, rows, Cout, CoutPad,
                       (unsigned)((unsigned long long)rows * PW_K * 4ull), (unsigned)((unsigned long long)rows * Cout * 4ull));
    return check_launch("conv3d_fwd(1x1x1, K = 64)");
}

}  // namespace diqt
