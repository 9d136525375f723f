// conv_f9h_kernel -- the 16-bit forward conv (fp16 / bf16 operands, fp32 accumulate) rebuilt on conv_fwd9_kernel's structure (round 4).
// Included by the translation units that instantiate its variants (conv_f9h.hip, conv_f9h_b.hip, conv_f9h_c.hip).
//
// What bounded the round-3 kernels (conv_fwd_h_kernel / conv_fwd_hp_kernel): every operand byte went global -> VGPR -> LDS under one or
// two barriers per 4-36 MFMAs of a wave, two waves per SIMD at 256 registers, and a 4-way bank conflict on every fragment read
// (64-byte rows).  With a 32-cycle MFMA (16x the f32 rate) each of those costs 16x more, relatively, than in the f32 kernel.  Here:
//   * x is 16-bit in HBM (the GroupNorm-apply pass writes it so) and the halo image of a 32-channel chunk is copied VERBATIM by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`), NIMG - 1 chunks ahead of the MFMAs, into a ring of NIMG images (2 for 3x3x3, 3 for (1,3,3)):
//     no register staging, no conversion, no tables; zero padding / ragged tiles = out-of-range buffer offsets (the DMA writes zeros).
//   * SWIZZLED image, free at run time: the 16-byte octet o of halo row (hz, hy, hx) sits in slot o ^ g, g = ((hx >> 2) & 1) |
//     (((hy >> 1) & 1) << 1).  The DMA lane that fills slot s simply fetches octet s ^ g (its source offset is tile-independent), and
//     the reading lane's g depends only on its own voxel (j, i) inside a 4 x 8 block and on the tap's (ky, kx): KH x KW x 2 per-lane
//     base addresses, every tap / block offset an immediate.  With the voxel <-> lane map below the 16 lanes of each ds_read_b128
//     service group hit 16 distinct 16-byte slots of the 64 banks: conflict-free (the un-swizzled 64-byte rows were 4-way).
//   * TRANSPOSED product  D^T[co][voxel] = W[co][k] X^T[k][voxel]:  the A operand is the weight fragment, the B operand the voxel
//     fragment, so a lane ends up with 4 CONSECUTIVE output channels of ONE voxel per accumulator quad: the epilogue stores 8 B (16-bit
//     y) or 16 B (fp32 y) per lane and instruction instead of 2 / 4 B (128 -> 32 store instructions per wave and tile).
//   * ONE wave per SIMD, 256 threads, wave (a, b) = voxel half a x 32-channel half b of a (64 NVB)-voxel x 64-channel tile: NVB = 8
//     accumulator tiles.  The WEIGHT fragments never touch the LDS: each wave streams its 32 co x 32 ci panel of a tap (2 x 1 KiB) from
//     L2 into a 9-slot register ring, 8 taps ahead (loads retire in order behind the halo DMA pieces, so the distance has to cover an
//     HBM latency) -- hence NO barrier inside a chunk: one `s_barrier` per T taps (13.8k / 4.6k MFMA cycles for 27 / 9 taps), and no
//     s_waitcnt of our own at all: the wait hipcc puts in front of the last tap's weight fragment (issued after every halo piece of the
//     image the next chunk needs) is what guarantees that image before the barrier.
//   * persistent tile walk with the weight stream running on (the packed [chunk][tap] order is one linear sequence, wrapped per tile).
//   * epilogue: bias (fp32, behind the K sum: bit-identical to the older 16-bit kernels), round to the operand type (autocast semantics), optional fp32 residual, optional column sums (sum, sum of squares
//     of the STORED values) per (tile, voxel half) for the consumer's GroupNorm -- per-lane partials over the wave's blocks, reduced over
//     the 32 voxel lanes through a wave-private, bank-skewed LDS scratch in a fixed order (deterministic, no barrier).
// Lane <-> voxel map of a block (rows j = 0..3, columns i = 0..7), t = l31 >> 2:  j = 2 t2 + parity(t),  i = 4 t1 + (l31 & 3).
// ds_read_b128 serves lanes {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (per half) together = the even / odd parity t's = rows {0, 2} /
// {1, 3}: with HWd even, 2 j is constant mod 4 inside a group, so the four lanes that share R mod 4 differ in (i >> 2) or (j >> 1) --
// exactly the two bits of g.
// Reference call sites: Block.project under torch.autocast (/root/reference/imagen_pytorch3D.py:535-566, trainer.py:293-311) and the
// per-frame Conv2d of the pseudo-3D blocks (/root/reference/imagen_video.py:352-381, 671-697) inside ElucidatedImagen.sample.
#pragma once
#include "common.h"
#include "conv_f9h.h"

namespace diqt {
namespace h9 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr unsigned OOB = 0x80000000u;
constexpr int CK = 32, ROWB = 64;        // channels per chunk, bytes per halo row / weight row
constexpr int PD = 9;                    // slots of the weight-fragment ring (prefetch distance PD - 1 taps)
constexpr int SCRW = 32 * 144;           // epilogue transpose scratch per wave: 32 voxel rows x (128 B of fp32 channels + 16 B pad)

template <int KD_, int KH_, int KW_, int TD_, int TH_, int TW_, int NIMG_, int OCC_ = 1>
struct Cfg {
    static constexpr int KD = KD_, KH = KH_, KW = KW_, TD = TD_, TH = TH_, TW = TW_, NIMG = NIMG_, OCC = OCC_;      // OCC: workgroups per CU
    static constexpr int T = KD * KH * KW;
    static constexpr int HD = TD + KD - 1, HH = TH + KH - 1, HWd = TW + KW - 1, HV = HD * HH * HWd;
    static constexpr int HB = (HV * ROWB + 4095) / 4096 * 4096;        // image in whole 1-KiB DMA instructions, the same count per wave
    static constexpr int NPH = HB / 4096;                             // DMA pieces per wave and chunk
    static constexpr int NBH = TH / 4, NBW = TW / 8, NBLK = TD * NBH * NBW, NVB = NBLK / 2;
    static constexpr int LDS_BYTES = NIMG * HB;                        // (the epilogue's transpose scratch lives in the image that is free then)
    static_assert(TH % 4 == 0 && TW % 8 == 0 && NBLK % 2 == 0 && HWd % 2 == 0, "tile = 2 voxel halves x NVB blocks of 4 x 8 voxels; even halo pitch");
    static_assert(T % PD == 0, "the ring slot of a tap is static: T is a multiple of the ring length");
    static_assert(LDS_BYTES * OCC <= 160 * 1024 && NPH <= 16 && HB >= 4 * SCRW, "LDS budget; halo piece descriptors live in registers; scratch fits an image");
    __host__ __device__ static constexpr int tapoff(int t) { return ((t / (KW * KH)) * HH + (t / KW) % KH) * HWd + t % KW; }
    __host__ __device__ static constexpr int blockrow(int id) { return ((id / (NBH * NBW)) * HH + ((id / NBW) % NBH) * 4) * HWd + (id % NBW) * 8; }
    __host__ __device__ static constexpr bool blocks_regular() {
        for (int vb = 0; vb < NVB; ++vb) if (blockrow(NVB + vb) - blockrow(NVB) != blockrow(vb)) return false;
        return true;
    }
    static_assert(blocks_regular(), "the blocks of both voxel halves sit at the same offsets from the half's first block");
    static_assert((blockrow(NVB - 1) + tapoff(T - 1)) * ROWB + 48 < 65536, "16-bit LDS immediates");
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(size_t)lds, 16, voff, 0, 0, 0);
}

template <bool BF> __device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    if (BF) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <bool BF> __device__ __forceinline__ unsigned pack2(float a, float b) {
    if (BF) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        bf2 v = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, v);
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 v = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, v);
    }
}
template <bool BF> __device__ __forceinline__ float round_through(float a) { return BF ? (float)(__bf16)a : (float)(_Float16)a; }

// x: 16-bit NDHWC; wp: the packed 16-bit weights of conv_pack_weight_h_kernel, [chunk][tap][co pad 64][32 ci]; y: fp32 or (YH) 16-bit
template <class C, bool BF, bool YH>
__global__ __launch_bounds__(256, C::OCC) void conv_f9h_kernel(const void* __restrict__ xv, const unsigned short* __restrict__ wp,
                                                          const float* __restrict__ bias, const float* __restrict__ residual,
                                                          void* __restrict__ yv, H9Geom g) {
    constexpr int T = C::T, HB = C::HB, NPH = C::NPH, NVB = C::NVB, HH = C::HH, HWd = C::HWd, HV = C::HV, NIMG = C::NIMG;
    constexpr int KH = C::KH, KW = C::KW;
    constexpr int NSP = NIMG == 2 ? T - PD : T;       // taps of a chunk that issue halo pieces (two images: all older than the last tap's weight load)
    constexpr unsigned YE = YH ? 2u : 4u;
    static_assert(NSP >= 1, "halo pieces need a tap to ride on");
    auto nh_in_tap = [](int t_) constexpr { int n_ = 0; for (int r = 0; r < C::NPH; ++r) n_ += (r * NSP / C::NPH == t_) ? 1 : 0; return n_; };
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5;
    const int wa = wave >> 1, wb = wave & 1;            // voxel half, channel half of the tile
    const unsigned Gn = gridDim.x;
    const unsigned total = (unsigned)g.MT * g.nNt;
    unsigned L = xcd_remap(blockIdx.x, Gn);
    if (L >= total) return;
    const unsigned L0 = L;                               // first tile of this workgroup (L walks on with the tiles being computed)
    const int nMine = (int)((total - 1 - L) / Gn) + 1;
    const int n0 = (int)(L % g.nNt) * 64;
    const int nChunks = g.nChunks;

    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(xv), 0, (int)g.xBytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(wp), 0, (int)g.wBytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(yv, 0, (int)g.yBytes, 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual), 0, residual ? (int)g.rBytes : 0, 0x00020000);
    const unsigned ldsBase = (unsigned)(size_t)(lds_void*)smem;
    // diagnostic stamps (g.dbg set by DIQT_F9H_DBG=1 only): 32 slots per wave -- 0 start, 1 prologue done, then per unit (chunk start, before
    // the barrier, after it), per tile (epilogue done)
    int dslot = 0;
    auto stamp = [&]() __attribute__((always_inline)) {
        if (g.dbg) {
            const unsigned long long tnow = __builtin_readcyclecounter();
            // (DIQT_F9H_DBG=2: the stamps from slot 80 on instead of the first 32 -- steady state of a long tile walk)
            const int ds_ = dslot - g.dbgSkip;
            if (lane == 0 && dslot == 0) g.dbg[((size_t)blockIdx.x * 4 + wave) * 32 + 28] = __builtin_amdgcn_s_memrealtime();   // 100 MHz
            if (lane == 0) g.dbg[((size_t)blockIdx.x * 4 + wave) * 32 + 29] = __builtin_amdgcn_s_memrealtime();
            if (lane == 0 && ds_ >= 0 && ds_ < 28) g.dbg[((size_t)blockIdx.x * 4 + wave) * 32 + ds_] = tnow;
            if (lane == 0 && dslot == 0) g.dbg[((size_t)blockIdx.x * 4 + wave) * 32 + 30] = tnow;      // slot 30: the wave's start, 31: its latest stamp
            if (lane == 0) g.dbg[((size_t)blockIdx.x * 4 + wave) * 32 + 31] = tnow;
            ++dslot;
        }
    };
    stamp();

    // ---- tile-independent description of this lane's halo DMA pieces ----
    unsigned posH[NPH];              // hz | hy << 10 | hx << 20 | source octet << 30 (its slot ^ swizzle); rows past the image: hx = 1023 (never inside)
#pragma unroll
    for (int r = 0; r < NPH; ++r) {
        const int p = (wave + 4 * r) * 64 + lane;                  // 16-byte slot of the image: row p / 4, slot p % 4
        const int row = p >> 2;
        const int hx = row % HWd, hy = (row / HWd) % HH, hz = row / (HWd * HH);
        const int gsw = ((hx >> 2) & 1) | (((hy >> 1) & 1) << 1);
        posH[r] = (unsigned)hz | ((unsigned)hy << 10) | ((unsigned)(row < HV ? hx : 1023) << 20) | ((unsigned)((p & 3) ^ gsw) << 30);
    }
    const int Dm1 = g.D - 1, Hm1 = g.H - 1, Wm1 = g.W - 1;
    int tb, d0, h0, w0;                                    // tile being computed
    int bz = 0, by = 0, bxx = 0;                           // tile whose halo is being fetched (origin minus padding)
    unsigned baseX = 0, deadX = OOB;
    auto tile_of = [&](unsigned Lt, int& b_, int& d_, int& h_, int& w_) __attribute__((always_inline)) {
        int mt = (int)(Lt / g.nNt);
        const int tx = mt % g.tilesW; mt /= g.tilesW;
        const int ty = mt % g.tilesH; mt /= g.tilesH;
        const int tz = mt % g.tilesD;
        b_ = mt / g.tilesD; d_ = tz * C::TD; h_ = ty * C::TH; w_ = tx * C::TW;
    };
    auto set_fetch = [&](int fit_) __attribute__((always_inline)) {
        const bool live = fit_ < nMine;
        int b_, d_, h_, w_;
        tile_of(live ? L0 + (unsigned)fit_ * Gn : L0, b_, d_, h_, w_);
        bz = d_ - g.pd; by = h_ - g.ph; bxx = w_ - g.pw;
        baseX = (unsigned)((((b_ * g.D + bz) * g.H + by) * g.W + bxx) * g.Cin) * 2u;
        deadX = live ? 0u : OOB;
    };
    const int HWs = g.W * g.Cin * 2, HHs = g.H * HWs, Cs = g.Cin * 2;       // byte strides of x along H, D, W
    auto dma_h = [&](int r, unsigned imgBase, int chunk) __attribute__((always_inline)) {     // r static
        const unsigned p = posH[r];
        const int hz = (int)(p & 1023u), hy = (int)((p >> 10) & 1023u), hx = (int)((p >> 20) & 1023u);
        const int iz = bz + hz, iy = by + hy, ix = bxx + hx;
        const unsigned m = (unsigned)(iz | iy | ix) | (unsigned)((Dm1 - iz) | (Hm1 - iy) | (Wm1 - ix)) | deadX;
        const unsigned rel = (unsigned)(hz * HHs + hy * HWs + hx * Cs) + ((p >> 30) << 4);
        const unsigned voff = (baseX + rel + (unsigned)chunk * ROWB) | (m & OOB);
        dma16(rs_x, imgBase + (unsigned)(wave + 4 * r) * 1024u, voff);
    };

    // ---- weight fragments: A operand, row = co (32 wb + l31), k-octet 2 q + hf of the 32-channel chunk; a wave-load is 32 rows x 32 B ----
    const unsigned wLane = (unsigned)((n0 + 32 * wb + l31) * ROWB + hf * 16);
    const unsigned wStride = (unsigned)g.CoutPad * ROWB;      // bytes between consecutive (chunk, tap) panels
    const int WT = nChunks * T;                               // panels of the linear weight sequence of a tile
    u32x4 Wr[PD][2];
    int wlin = 0;
    unsigned wsoff = 0;
    auto w_load = [&](int slot) __attribute__((always_inline)) {                               // slot static
        Wr[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wLane, wsoff, 0);
        Wr[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wLane + 32u, wsoff, 0);
        wsoff += wStride;
        if (++wlin == WT) { wlin = 0; wsoff = 0; }
    };

    // ---- voxel fragments: B operand, column = this lane's voxel (lj, li) of a block ----
    const int tq = l31 >> 2;
    const int lj = 2 * (tq >> 2) + ((tq ^ (tq >> 1) ^ (tq >> 2)) & 1), li = 4 * ((tq >> 1) & 1) + (l31 & 3);
    int xa[KH][KW];                                           // LDS byte address of tap (ky, kx), k-half 0 at block 0 of this wave's voxel half, image `img`
#pragma unroll                                                // (k-half 1: the same ^ 32 -- bit 5 of the address is (q ^ gy), nothing carries into it)
    for (int ky = 0; ky < KH; ++ky)
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int gx = ((li + kx) >> 2) & 1, gy = ((lj + ky) >> 1) & 1;
            xa[ky][kx] = (C::blockrow(0) + (wa ? C::blockrow(NVB) : 0) + lj * HWd + li) * ROWB + ((hf ^ gx) << 4) + (gy << 5);
        }

    // output channels of this lane's accumulator rows: cb + 8 (r >> 2) + (r & 3).  The bias is added in fp32 BEHIND the K sum, as in
    // conv_fwd_h_kernel / conv_fwd_hp_kernel (same chunk -> tap -> k-half order of the sum too): the three kernels agree bit for bit.
    // (It is re-loaded per tile instead of held in 16 registers: the 256-register builds need them.)
    const int cb = n0 + 32 * wb + 4 * hf;
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias ? g.Cout * 4 : 0, 0x00020000);
    f32x16 acc[NVB];
#pragma unroll
    for (int vb = 0; vb < NVB; ++vb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[vb][i] = 0.f;

    // ---- prologue: the first NIMG - 1 units' halo images, the first PD - 1 weight panels ----
    int fit = 0, fc = 0;                                   // (tile iteration, chunk) of the unit being fetched
    set_fetch(0);
    auto advance_fetch = [&]() __attribute__((always_inline)) {
        if (++fc == nChunks) { fc = 0; ++fit; set_fetch(fit); }
    };
#pragma unroll
    for (int k = 0; k < NIMG - 1; ++k) {
#pragma unroll
        for (int r = 0; r < NPH; ++r) dma_h(r, ldsBase + (unsigned)k * HB, fc);
        advance_fetch();
    }
#pragma unroll
    for (int s = 0; s < PD - 1; ++s) w_load(s);
    __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0)
    __syncthreads();
    stamp();

    int img = 0, fimg = NIMG - 1;
    tile_of(L, tb, d0, h0, w0);
    for (int it = 0; it < nMine; ++it) {
        for (int c = 0; c < nChunks; ++c) {
            const unsigned fbase = ldsBase + (unsigned)fimg * HB;
            u32x4 X0[NVB], X1[NVB];
            auto rd = [&](u32x4 (&X)[NVB], int t, int q) __attribute__((always_inline)) {       // t, q static
                const int kz = t / (KH * KW), ky = (t / KW) % KH, kx = t % KW;
                const char* base = smem + (xa[ky][kx] ^ (q << 5));
#pragma unroll
                for (int vb = 0; vb < NVB; ++vb)
                    X[vb] = *reinterpret_cast<const u32x4*>(base + (C::blockrow(vb) - C::blockrow(0) + (kz * HH + ky) * HWd + kx) * ROWB);
            };
            auto mm = [&](u32x4 (&X)[NVB], int t, int q) __attribute__((always_inline)) {
#pragma unroll
                for (int vb = 0; vb < NVB; ++vb) acc[vb] = mfma16<BF>(Wr[t % PD][q], X[vb], acc[vb]);
            };
            stamp();
            rd(X0, 0, 0);                                  // cold read of the chunk's first half-tap (the image was published by the barrier)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                // ---- a tap = one scheduling region (fenced at its start: the memory instructions keep their order ACROSS taps -- halo pieces
                //      of the unit NIMG - 1 ahead before the weight panel PD - 1 taps ahead, which goes into the slot the previous tap has
                //      left).  Inside, the pipeline below: behind every MFMA one fragment read of the next half-tap, and the tap's (at most
                //      two + two) memory instructions one per MFMA gap instead of back to back in front of the first MFMA (where the matrix
                //      pipe drained behind their address arithmetic: 2.4k of a chunk's 17.6k cycles) ----
                __builtin_amdgcn_sched_barrier(0);
                rd(X1, t, 1);
                mm(X0, t, 0);
                // (in program order BEHIND the first half's fragment reads and in front of the second half's: to the compiler a halo DMA is
                // a store into the LDS that no fragment read may cross)
#pragma unroll
                for (int r = 0; r < NPH; ++r)
                    if (r * NSP / NPH == t) dma_h(r, fbase, fc);
                w_load((t + PD - 1) % PD);
                if (t + 1 < T) rd(X0, t + 1, 0);
                mm(X1, t, 1);
#pragma unroll
                for (int u = 0; u < 2 * NVB; ++u) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (u >= NVB && u < NVB + 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            (void)nh_in_tap;
            // End of the unit.  The image of the next unit is complete: its pieces were issued before the weight load whose fragment the
            // last tap has just consumed (loads retire in order).  This wave's reads of image `img` have been consumed by issued MFMAs.
            stamp();
            asm volatile("s_barrier" ::: "memory");
            stamp();
            advance_fetch();
            {
                const int nimg = img + 1 == NIMG ? 0 : img + 1;
                const int delta = (nimg - img) * HB;
#pragma unroll
                for (int ky = 0; ky < KH; ++ky)
#pragma unroll
                    for (int kx = 0; kx < KW; ++kx) xa[ky][kx] += delta;
                fimg = img;
                img = nimg;
            }
        }
        // ---- epilogue of the tile.  The accumulators hold D^T[row = co][col = voxel] + bias: lane (l31, hf) has, for voxel (lj, li) of block
        //      vb, channels cb + 8 g4 + e (four quads).  They are rounded to the operand type and TRANSPOSED through this wave's LDS scratch so
        //      that the lanes of a store instruction cover WHOLE 64-byte (16-bit y) / 128-byte (fp32 y) channel rows of 16 / 8 voxels: as 8- /
        //      16-byte pieces of 32 different rows per instruction the stores were bound by the L2's request rate (one per clock and channel:
        //      64 KB per tile and CU took 7k cycles, every CU in its epilogue at the same time).  Residual and statistics ride on the
        //      transposed layout: coalesced residual loads, 2 x PW partial sums per lane, shuffles over the lanes that share a piece ----
        {
            constexpr int NK = YH ? 2 : 4, PW = YH ? 8 : 4, LPV = YH ? 4 : 8;       // store instructions per block, channels per piece, lanes per voxel
            constexpr int SROW = YH ? 80 : 144;                                      // scratch row bytes (64 / 128 + pad)
            // scratch: the image the last unit was computed from (`fimg` after the rotation) -- free until the next unit's taps issue the
            // halo pieces of the unit NIMG - 1 ahead into it, which the barrier behind the epilogue holds back
            char* const scrb = smem + fimg * HB + wave * SCRW;
            const int piece = lane & (LPV - 1);
            const int cop = n0 + 32 * wb + piece * PW;                               // first channel of this lane's piece
            const bool cok = cop < g.Cout;
            const bool wantStats = g.stats != nullptr;                               // kernel-uniform
            float ssum[PW], ssq[PW];
#pragma unroll
            for (int e = 0; e < PW; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
            f32x4v b4[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                b4[g4] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (unsigned)(cb + 8 * g4) * 4u, 0, 0));
#pragma unroll
            for (int vb = 0; vb < NVB; ++vb) {
                const int id = wa * NVB + vb;
                const int od = d0 + id / (C::NBH * C::NBW), ohb = h0 + ((id / C::NBW) % C::NBH) * 4, owb = w0 + (id % C::NBW) * 8;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    if constexpr (YH) {
                        u32x2 p;
                        p.x = pack2<BF>(acc[vb][4 * g4] + b4[g4][0], acc[vb][4 * g4 + 1] + b4[g4][1]);
                        p.y = pack2<BF>(acc[vb][4 * g4 + 2] + b4[g4][2], acc[vb][4 * g4 + 3] + b4[g4][3]);
                        *reinterpret_cast<u32x2*>(scrb + l31 * SROW + 16 * g4 + 8 * hf) = p;
                    } else {
                        f32x4v p;
#pragma unroll
                        for (int e = 0; e < 4; ++e) p[e] = round_through<BF>(acc[vb][4 * g4 + e] + b4[g4][e]);
                        *reinterpret_cast<f32x4v*>(scrb + l31 * SROW + 32 * g4 + 16 * hf) = p;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[vb][4 * g4 + e] = 0.f;
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const int vloc = (64 / LPV) * k + lane / LPV;                  // voxel of the block this lane stores in instruction k
                    const int tqr = vloc >> 2;
                    const int rj = 2 * (tqr >> 2) + ((tqr ^ (tqr >> 1) ^ (tqr >> 2)) & 1), ri = 4 * ((tqr >> 1) & 1) + (vloc & 3);
                    const int oh = ohb + rj, ow = owb + ri;
                    const bool ok = od < g.Do && oh < g.Ho && ow < g.Wo;
                    const unsigned vox = (unsigned)(((tb * g.Do + od) * g.Ho + oh) * g.Wo + ow);
                    const unsigned off = (ok && cok) ? (vox * (unsigned)g.Cout + (unsigned)cop) * YE : OOB;
                    u32x4 d = *reinterpret_cast<const u32x4*>(scrb + vloc * SROW + piece * 16);
                    if constexpr (!YH) {
                        if (residual) {                    // kernel-uniform
                            const f32x4v r = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_r, off, 0, 0));
                            f32x4v f = __builtin_bit_cast(f32x4v, d);
#pragma unroll
                            for (int e = 0; e < 4; ++e) f[e] += r[e];
                            d = __builtin_bit_cast(u32x4, f);
                        }
                    }
                    if (wantStats) {
                        const float okf = ok ? 1.f : 0.f;
                        float v[PW];
                        if constexpr (YH) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned wd = d[e];
                                if (BF) {
                                    v[2 * e] = __builtin_bit_cast(float, wd << 16);
                                    v[2 * e + 1] = __builtin_bit_cast(float, wd & 0xffff0000u);
                                } else {
                                    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                                    const h2 hv = __builtin_bit_cast(h2, wd);
                                    v[2 * e] = (float)hv[0];
                                    v[2 * e + 1] = (float)hv[1];
                                }
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) { const unsigned wd = d[e]; v[e] = __builtin_bit_cast(float, wd); }   // (by value: see conv_half.hip asf)
                        }
#pragma unroll
                        for (int e = 0; e < PW; ++e) {
                            const float mv = okf * v[e];
                            ssum[e] += mv;
                            ssq[e] = fmaf(mv, v[e], ssq[e]);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs_y, off, 0, 0);
                }
            }
            if (wantStats) {
                // sums over the lanes that share a piece (lane bits above log2 LPV), fixed order; the piece's first lane writes the row
                const int tpb = g.tilesD * g.tilesH * g.tilesW, mtile = (int)(L / g.nNt);
                float* srow = g.stats + ((size_t)(mtile / tpb) * (2 * tpb) + (size_t)(mtile % tpb) * 2 + wa) * 2 * g.Cout;
#pragma unroll
                for (int e = 0; e < PW; ++e) {
#pragma unroll
                    for (int o = LPV; o < 64; o <<= 1) {
                        ssum[e] += __shfl_xor(ssum[e], o, 64);
                        ssq[e] += __shfl_xor(ssq[e], o, 64);
                    }
                }
                if (lane < LPV && cok) {
#pragma unroll
                    for (int e = 0; e < PW; ++e) { srow[cop + e] = ssum[e]; srow[g.Cout + cop + e] = ssq[e]; }
                }
            }
        }
        asm volatile("s_barrier" ::: "memory");              // every wave's scratch reads are done (their stores have been issued)
        stamp();
        L += Gn;
        if (it + 1 < nMine) tile_of(L, tb, d0, h0, w0);
    }
    // the (dead) halo pieces issued during the last units write zeros into this workgroup's LDS: they have to land before it is released
    __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0)
}

// the variants (filter, tile, images); H9Geom::variant indexes this list
using H9_333_512 = Cfg<3, 3, 3, 8, 8, 8, 2>;
using H9_333_256 = Cfg<3, 3, 3, 4, 8, 8, 2, 2>;      // 256-voxel tiles, TWO workgroups per CU (256 registers per wave): one's epilogue / barrier waits under the other's MFMAs
using H9_133_A = Cfg<1, 3, 3, 1, 16, 32, 3>;
using H9_133_B = Cfg<1, 3, 3, 2, 16, 16, 3>;
using H9_133_C = Cfg<1, 3, 3, 4, 8, 8, 3>;
using H9_133_D = Cfg<1, 3, 3, 1, 8, 32, 3, 2>;       // 256-voxel tiles, two workgroups per CU

template <class C> static int launch_cfg(const void* x, const unsigned short* wp, const float* bias, const float* residual, void* y,
                                         const H9Geom& g, size_t lds, unsigned grid, int bf16, bool yHalf, void* stream) {
    typedef void (*KP)(const void*, const unsigned short*, const float*, const float*, void*, H9Geom);
    const KP kern = yHalf ? (bf16 ? conv_f9h_kernel<C, true, true> : conv_f9h_kernel<C, false, true>)
                          : (bf16 ? conv_f9h_kernel<C, true, false> : conv_f9h_kernel<C, false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd_h(v9h): hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, (hipStream_t)stream, x, wp, bias, residual, y, g);
    return check_launch("conv3d_fwd_h(v9h)");
}

int launch_b(const void* x, const unsigned short* wp, const float* bias, const float* residual, void* y, const H9Geom& g, size_t lds,
             unsigned grid, int bf16, bool yHalf, void* stream);
int launch_c(const void* x, const unsigned short* wp, const float* bias, const float* residual, void* y, const H9Geom& g, size_t lds,
             unsigned grid, int bf16, bool yHalf, void* stream);

}  // namespace h9
}  // namespace diqt
