// conv_fwd9_kernel and its launcher template -- included by the translation units that instantiate its variants (conv_fwd9.hip,
// conv_fwd9_b.hip, conv_fwd9_c.hip: one variant takes ~1.5 minutes of hipcc, so they compile side by side).  See conv_fwd9.hip for
// the description of the kernel.
#pragma once
#include "common.h"
#include "conv_fwd9.h"

namespace diqt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void9;

constexpr unsigned F9_OOB = 0x80000000u;
constexpr int F9_CH = 16, F9_ROWB = F9_CH * 4;           // channels per chunk, bytes per halo voxel row / weight row
constexpr int F9_WTAP = 64 * F9_ROWB;                    // one tap's weight panel: 64 co x 16 ci (4 KiB = one DMA piece per wave)
constexpr int F9_TG = 2, F9_NWS = 3;                     // taps per step, slots of the weight ring
constexpr int F9_WSLOT = F9_TG * F9_WTAP;

// Filter (KD, KH, KW), tile (TD, TH, TW) and voxel blocks per wave NVB (4: 512-voxel tile, 2: 256).  A voxel BLOCK is 4 rows x 8
// columns of one plane = the 32 rows of an accumulator tile; block id = (plane * TH/4 + row group) * TW/8 + column group, wave = id / NVB.
template <int KD_, int KH_, int KW_, int TD_, int TH_, int TW_, int NVB_>
struct F9Cfg {
    static constexpr int KD = KD_, KH = KH_, KW = KW_, TD = TD_, TH = TH_, TW = TW_, NVB = NVB_;
    static constexpr int T = KD * KH * KW, NSTEP = (T + F9_TG - 1) / F9_TG;
    static constexpr int HD = TD + KD - 1, HH = TH + KH - 1, HWd = TW + KW - 1, HV = HD * HH * HWd;
    static constexpr int HB = (HV * F9_ROWB + 4095) / 4096 * 4096;      // halo image in whole 1-KiB DMA instructions, the same count per wave
    static constexpr int NPH = HB / 4096;                              // halo DMA pieces per wave and chunk
    static constexpr int NBH = TH / 4, NBW = TW / 8;
    static constexpr int LDS_BYTES = 2 * HB + F9_NWS * F9_WSLOT;
    static_assert(TD * TH * TW == 128 * NVB && TH % 4 == 0 && TW % 8 == 0, "tile = 4 waves x NVB blocks of 4 x 8 voxels");
    static_assert(HV * F9_ROWB <= 65536 && LDS_BYTES <= 160 * 1024, "halo image: 16-bit LDS immediates, two images + weight ring in LDS");
    static_assert(NPH <= 16, "halo piece descriptors live in registers");
    static_assert(T >= 3, "the prologue issues the weight groups of two steps");
    __host__ __device__ static constexpr int tapoff(int t) { return ((t / (KW * KH)) * HH + (t / KW) % KH) * HWd + t % KW; }
    __host__ __device__ static constexpr int blockrow(int id) { return ((id / (NBH * NBW)) * HH + ((id / NBW) % NBH) * 4) * HWd + (id % NBW) * 8; }
};

__device__ __forceinline__ void f9_dma(__amdgpu_buffer_rsrc_t rs, unsigned lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void9*)(size_t)lds, 16, voff, 0, 0, 0);
}

// Mish / SiLU derivative for the GroupNorm-backward epilogue (same expressions as act_grad in common.h, without its other cases:
// the call sits in a 128-fold unrolled store loop)
// Hardware reciprocals (v_rcp_f32, 1 ulp) instead of IEEE divisions: a division is ~10 instructions, and this runs 256 times per lane
// and tile on a wave that has nothing else to issue.
__device__ __forceinline__ float f9_act_grad(float x, int act) {
    if (act == DIQT_ACT_MISH) {
        const float xc = fminf(x, 20.f);                   // e^x (e^x + 2) stays finite; the derivative is 1 to fp32 precision beyond
        const float n = __expf(xc);
        const float m = n * (n + 2.f);
        const float t = m * __builtin_amdgcn_rcpf(m + 2.f);
        const float sg = n * __builtin_amdgcn_rcpf(1.f + n);
        const float r = t + xc * sg * (1.f - t * t);
        return x > 20.f ? 1.f : r;
    }
    const float s = __builtin_amdgcn_rcpf(1.f + __expf(-x));
    return s * (1.f + x * (1.f - s));
}

// end of a chunk in the GroupNorm-apply instantiations: the in-place transform's LDS stores have to be complete, too, before the barrier
// publishes the image (no fragment read is in flight across a chunk boundary: the first tap of a chunk is read cold)
__device__ __forceinline__ void f9_chunk_end_gna() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// s_waitcnt vmcnt(n) + s_barrier with nothing else attached (n is a constant after unrolling; the asm needs a literal)
__device__ __forceinline__ void f9_step_end(int n) {
#define F9_WB(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory"); break;
    switch (n) {
        F9_WB(0) F9_WB(1) F9_WB(2) F9_WB(3) F9_WB(4) F9_WB(5) F9_WB(6) F9_WB(7) F9_WB(8) F9_WB(9) F9_WB(10) F9_WB(11) F9_WB(12)
        F9_WB(13) F9_WB(14) F9_WB(15) F9_WB(16)
        default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
#undef F9_WB
}

// GNB: the GroupNorm-backward epilogue (F9Geom::gx).  A separate instantiation: behind a runtime flag in the plain kernel it cost the
// forward launches 10 % (448 vs 405 us on the dominant one); the epilogue itself is 8.6k instructions per tile (activation derivative).
// GNA (0: none, DIQT_ACT_MISH, DIQT_ACT_SILU): the GroupNorm-apply prologue (F9Geom::gcoef) -- x is the raw GroupNorm input; a wave
// rewrites each of ITS halo pieces in place, act(A x + Bc) with the per-(batch, channel) coefficients, two steps after it issued the
// piece's DMA (landed by then: the counted vmcnt of the step in between covers it; lane-private 16 bytes, so no barrier is involved
// until the chunk's last one).  Padding voxels (out-of-range pieces: the DMA wrote zeros) stay zero.  The pieces of the next chunk are
// issued over the first NSTEP - 2 steps so that the last ones can still be rewritten before the chunk ends.
template <class C, bool GNB, int GNA = 0>
__global__ __launch_bounds__(256, 1) void conv_fwd9_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           const float* __restrict__ bias, const float* __restrict__ residual,
                                                           float* __restrict__ y, F9Geom g) {
    static_assert(!(GNB && GNA) && (!GNA || C::NSTEP >= 3), "GroupNorm-apply prologue: forward launches of filters with >= 5 taps");
    constexpr int NSPREAD = GNA ? C::NSTEP - 2 : (C::NSTEP > 1 ? C::NSTEP - 1 : 1);   // steps that issue halo pieces of the next chunk
    auto nh_in_step = [](int s_) constexpr { int n_ = 0; for (int r = 0; r < C::NPH; ++r) n_ += (r * NSPREAD / C::NPH == s_) ? 1 : 0; return n_; };
    constexpr int T = C::T, NSTEP = C::NSTEP, HB = C::HB, NPH = C::NPH, NVB = C::NVB, HH = C::HH, HWd = C::HWd, HV = C::HV;
    constexpr int CH = F9_CH, ROWB = F9_ROWB, WTAP = F9_WTAP, WSLOT = F9_WSLOT, NWS = F9_NWS;
    constexpr unsigned OOB = F9_OOB;
    extern __shared__ __attribute__((aligned(1024))) char smem9[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hf = lane >> 5;
    const unsigned Gn = gridDim.x;
    const unsigned total = (unsigned)g.MT * g.nNt;
    unsigned L = xcd_remap(blockIdx.x, Gn);
    if (L >= total) return;
    const int nMine = (int)((total - 1 - L) / Gn) + 1;
    const int n0 = (int)(L % g.nNt) * 64;
    // split-K launches (gridDim.y > 1; small volumes: the 8^3 level of C2): this workgroup walks chunksPerSplit of the 16-channel
    // chunks and writes its partial sums into slab blockIdx.y (bias / residual / statistics come with conv_fwd_reduce_kernel)
    const int cBeg = (int)blockIdx.y * g.chunksPerSplit;
    const int cEnd = min(g.Cin / CH, cBeg + g.chunksPerSplit);

    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)g.xBytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, (int)g.wBytes, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)blockIdx.y * g.slabElems, 0, (int)g.yBytes, 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(residual), 0, residual ? (int)g.yBytes : 0, 0x00020000);
    const unsigned ldsBase = (unsigned)(size_t)(lds_void9*)smem9;
    const unsigned wringBase = ldsBase + 2 * HB;

    // ---- tile-independent description of this lane's halo DMA pieces (16 per chunk; registers for the whole kernel) ----
    unsigned posH[NPH], relH[NPH];
#pragma unroll
    for (int r = 0; r < NPH; ++r) {
        const int p = (wave + 4 * r) * 64 + lane;                  // 16-byte piece of the image: row p / 4, channel quad p % 4
        const int row = p >> 2;
        const int hx = row % HWd, hy = (row / HWd) % HH, hz = row / (HWd * HH);
        posH[r] = (unsigned)hz | ((unsigned)hy << 8) | ((unsigned)hx << 16) | (row < HV ? 0u : 1u << 24);
        relH[r] = (unsigned)(((hz * g.H + hy) * g.W + hx) * g.Cin) * 4u + (unsigned)(p & 3) * 16u;
    }
    // this lane's weight piece of a tap panel: co row 16 w + lane / 4, channel quad lane % 4 of the 16-channel sub-chunk
    const unsigned relW = (unsigned)((n0 + 16 * wave + (lane >> 2)) * 128 + (lane & 3) * 16);
    const unsigned tapStrideW = (unsigned)g.CoutPad * 128u;          // bytes between the panels of consecutive taps (32-wide packed rows)
    auto dma_w = [&](int c16, int tap, unsigned ldsSlot, int tapInStep) __attribute__((always_inline)) {
        const unsigned voff = ((unsigned)(c16 >> 1) * T + (unsigned)tap) * tapStrideW + (unsigned)(c16 & 1) * 64u + relW;
        f9_dma(rs_w, ldsSlot + (unsigned)tapInStep * WTAP + (unsigned)wave * 1024u, voff);
    };

    // ---- tile coordinates (wave-uniform) ----
    const int Dm1 = g.D - 1, Hm1 = g.H - 1, Wm1 = g.W - 1;
    int tb, d0, h0, w0;                                    // tile being computed
    int bz, by, bxx;                                       // tile whose halo is being fetched (origin minus padding) ...
    int fb = 0;                                            // ... and its batch entry (GroupNorm-apply prologue)
    unsigned baseX = 0, deadX = OOB;                       // ... its byte base, and 0x80000000 when there is none
    auto tile_of = [&](unsigned Lt, int& b_, int& d_, int& h_, int& w_) __attribute__((always_inline)) {
        int mt = (int)(Lt / g.nNt);
        const int tx = mt % g.tilesW; mt /= g.tilesW;
        const int ty = mt % g.tilesH; mt /= g.tilesH;
        const int tz = mt % g.tilesD;
        b_ = mt / g.tilesD; d_ = tz * C::TD; h_ = ty * C::TH; w_ = tx * C::TW;
    };
    auto set_fetch = [&](unsigned Lt, bool live) __attribute__((always_inline)) {
        int b_, d_, h_, w_;
        tile_of(live ? Lt : L, b_, d_, h_, w_);
        bz = d_ - g.pd; by = h_ - g.ph; bxx = w_ - g.pw;
        baseX = (unsigned)((((b_ * g.D + bz) * g.H + by) * g.W + bxx) * g.Cin) * 4u;
        deadX = live ? 0u : OOB;
        if constexpr (GNA != 0) fb = b_;
    };
    auto dma_h = [&](int r, unsigned hbuf, int c16) __attribute__((always_inline)) {     // r static
        const unsigned p = posH[r];
        const int iz = bz + (int)(p & 255u), iy = by + (int)((p >> 8) & 255u), ix = bxx + (int)((p >> 16) & 255u);
        const unsigned m = (unsigned)(iz | iy | ix) | (unsigned)((Dm1 - iz) | (Hm1 - iy) | (Wm1 - ix)) | (p << 7) | deadX;
        const unsigned voff = (baseX + relH[r] + (unsigned)c16 * ROWB) | (m & OOB);
        f9_dma(rs_x, hbuf + (unsigned)(wave + 4 * r) * 1024u, voff);
    };

    // ---- GroupNorm-apply prologue: coefficients of the fetched (batch entry, chunk) for this lane's channel quad, the in-place rewrite ----
    const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.gcoef), 0, GNA ? (int)(2u * (unsigned)g.B * (unsigned)g.Cin * 4u) : 0, 0x00020000);
    f32x4v cA = {0.f, 0.f, 0.f, 0.f}, cB = {0.f, 0.f, 0.f, 0.f};
    auto gna_coef = [&](int c16) __attribute__((always_inline)) {
        const unsigned off = ((unsigned)(fb * g.Cin + c16 * CH) + (unsigned)(lane & 3) * 4u) * 4u;
        cA = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_c, off, 0, 0));
        cB = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_c, off + (unsigned)(g.B * g.Cin) * 4u, 0, 0));
    };
    auto gna_mask = [&](int r) __attribute__((always_inline)) -> bool {                // r static: is this lane's piece r inside the volume?
        const unsigned p = posH[r];
        const int iz = bz + (int)(p & 255u), iy = by + (int)((p >> 8) & 255u), ix = bxx + (int)((p >> 16) & 255u);
        const unsigned m = (unsigned)(iz | iy | ix) | (unsigned)((Dm1 - iz) | (Hm1 - iy) | (Wm1 - ix)) | (p << 7) | deadX;
        return !(m & OOB);
    };
    auto gna_ptr = [&](int r, int himg) __attribute__((always_inline)) -> f32x4v* {
        return reinterpret_cast<f32x4v*>(smem9 + himg * HB + (wave + 4 * r) * 1024 + lane * 16);
    };
    // one element of a piece: hardware reciprocal (1 ulp) instead of act_fwd's IEEE division (~10 instructions) -- this arithmetic
    // shares the issue slots between the MFMAs of a wave that has the SIMD to itself
    auto gna_elem = [&](float xv, float a, float b, bool live) __attribute__((always_inline)) -> float {
        const float z = a * xv + b;
        float r;
        if constexpr (GNA == DIQT_ACT_MISH) {
            const float n = __expf(fminf(z, 20.f));
            const float mm = n * (n + 2.f);
            r = z > 20.f ? z : z * (mm * __builtin_amdgcn_rcpf(mm + 2.f));
        } else {
            r = z * __builtin_amdgcn_rcpf(1.f + __expf(-z));
        }
        return live ? r : 0.f;
    };
    auto gna_xform = [&](int r, int himg) __attribute__((always_inline)) {           // whole piece at once (prologue); r static
        f32x4v* q = gna_ptr(r, himg);
        f32x4v v = *q;
        const bool live = gna_mask(r);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gna_elem(v[e], cA[e], cB[e], live);
        *q = v;
    };
    // Inside the main loop the rewrite of a step's pieces (those issued two steps before) is cut into SLICES that the step's tap
    // loops place between their MFMA groups (F9_MMX): slice 0 reads the pieces back, the elements follow one or a few per slice, a
    // piece is stored with its last element.  (In one block in front of the step's MFMAs the same instructions left the matrix pipe
    // idle for ~1k cycles per step: +7 % on the dominant launch, more than the elementwise pass they replace.)
    constexpr int GNA_MAXP = (NPH + NSPREAD - 1) / NSPREAD + 1;                      // pieces rewritten per step, at most
    auto gna_np = [](int s_) constexpr { int n_ = 0; for (int r = 0; r < C::NPH; ++r) n_ += (s_ >= 2 && r * NSPREAD / C::NPH == s_ - 2) ? 1 : 0; return n_; };
    auto gna_piece = [](int s_, int p_) constexpr { int c_ = 0; for (int r = 0; r < C::NPH; ++r) if (s_ >= 2 && r * NSPREAD / C::NPH == s_ - 2) { if (c_ == p_) return r; ++c_; } return 0; };
    f32x4v gxv[GNA ? GNA_MAXP : 1];
    bool glive[GNA ? GNA_MAXP : 1];
    auto gna_slice = [&](int s_, int k, int ns, int himg) __attribute__((always_inline)) {   // s_, k, ns static
        const int np = gna_np(s_);
        if (np == 0) return;
        if (k == 0) {
#pragma unroll
            for (int p_ = 0; p_ < GNA_MAXP; ++p_)
                if (p_ < np) { gxv[p_] = *gna_ptr(gna_piece(s_, p_), himg); glive[p_] = gna_mask(gna_piece(s_, p_)); }
            return;
        }
        // element j (piece j / 4, component j % 4) goes to slice 1 + j (ns - 1) / (4 np)
#pragma unroll
        for (int j = 0; j < 4 * GNA_MAXP; ++j) {
            if (j < 4 * np && 1 + j * (ns - 1) / (4 * np) == k) {
                const int p_ = j / 4, e = j % 4;
                gxv[p_][e] = gna_elem(gxv[p_][e], cA[e], cB[e], glive[p_]);
                if (e == 3) *gna_ptr(gna_piece(s_, p_), himg) = gxv[p_];
            }
        }
    };

    f32x16 acc[NVB][2];
#pragma unroll
    for (int vb = 0; vb < NVB; ++vb)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[vb][ch][i] = 0.f;

    const int co0 = n0 + l31, co1 = n0 + 32 + l31;
    const float bias0 = (bias && co0 < g.Cout) ? bias[co0] : 0.f;
    const float bias1 = (bias && co1 < g.Cout) ? bias[co1] : 0.f;
    const unsigned c0o = co0 < g.Cout ? (unsigned)co0 * 4u : 0x40000000u, c1o = co1 < g.Cout ? (unsigned)co1 * 4u : 0x40000000u;

    // per-lane operand bases: A = halo row of the lane's voxel inside its block (row l31 / 8, column l31 % 8) + its 16-byte half of a
    // k-group; the blocks of this wave start at rows blockrow(wave * NVB + vb) (wave-uniform), taps add compile-time row offsets
    const unsigned aLane = (unsigned)((l31 >> 3) * HWd + (l31 & 7)) * ROWB + (unsigned)hf * 16u;
    const unsigned bLane = (unsigned)l31 * ROWB + (unsigned)hf * 16u;
    int brow[NVB], bd[NVB], bh4[NVB], bw8[NVB];            // halo row / plane / first row / first column of the wave's blocks
#pragma unroll
    for (int vb = 0; vb < NVB; ++vb) {
        const int id = wave * NVB + vb;
        bd[vb] = id / (C::NBH * C::NBW); bh4[vb] = ((id / C::NBW) % C::NBH) * 4; bw8[vb] = (id % C::NBW) * 8;
        brow[vb] = (bd[vb] * HH + bh4[vb]) * HWd + bw8[vb];
    }

    // ---- prologue: halo chunk 0 of the first tile -> image 0, weight groups of steps 0 and 1 -> ring slots 0 and 1 ----
    tile_of(L, tb, d0, h0, w0);
    set_fetch(L, true);
#pragma unroll
    for (int r = 0; r < NPH; ++r) dma_h(r, ldsBase, cBeg);
    dma_w(cBeg, 0, wringBase, 0); dma_w(cBeg, 1, wringBase, 1);
    dma_w(cBeg, 2, wringBase + WSLOT, 0);
    if (3 < T) dma_w(cBeg, 3, wringBase + WSLOT, 1);
    if constexpr (GNA != 0) gna_coef(cBeg);
    __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0)
    if constexpr (GNA != 0) {
#pragma unroll
        for (int r = 0; r < NPH; ++r) gna_xform(r, 0);
    }
    __syncthreads();

    int wcur = 0, hcur = 0;
    for (int it = 0; it < nMine; ++it) {
        const bool lastTile = it + 1 == nMine;
        for (int c = cBeg; c < cEnd; ++c) {
            // which (tile, chunk) the halo pieces issued during this chunk belong to, and the chunk the wrapped weight steps belong to
            const bool wrap = c + 1 == cEnd;
            const int cNext = wrap ? cBeg : c + 1;
            if (wrap) set_fetch(L + Gn, !lastTile);        // from here on the fetches are the next tile's first chunk (or dead)
            if constexpr (GNA != 0) gna_coef(cNext);       // in front of the chunk's DMAs: landed with the first step's counted wait
            const unsigned hbufN = ldsBase + (unsigned)(hcur ^ 1) * HB;
            const char* hb[NVB];
#pragma unroll
            for (int vb = 0; vb < NVB; ++vb) hb[vb] = smem9 + hcur * HB + aLane + brow[vb] * ROWB;
            f32x4v A0[NVB][2], A1[NVB][2], B0[2][2], B1[2][2];   // ping-pong fragments: [voxel block | co half][k-group]
#define F9_RD(Av, Bv, TAP, WS, TIS)                                                                                  \
    do {                                                                                                             \
        const char* wb_ = smem9 + 2 * HB + (WS) * WSLOT + (TIS) * WTAP + bLane;                                      \
        _Pragma("unroll") for (int vb = 0; vb < NVB; ++vb)                                                           \
            _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                            \
                Av[vb][q] = *reinterpret_cast<const f32x4v*>(hb[vb] + C::tapoff(TAP) * ROWB + q * 32);               \
        _Pragma("unroll") for (int ch = 0; ch < 2; ++ch)                                                             \
            _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                            \
                Bv[ch][q] = *reinterpret_cast<const f32x4v*>(wb_ + ch * 32 * ROWB + q * 32);                         \
    } while (0)
#define F9_MM(Av, Bv)                                                                                                \
    do {                                                                                                             \
        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                                \
            _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                            \
                _Pragma("unroll") for (int vb = 0; vb < NVB; ++vb)                                                   \
                    _Pragma("unroll") for (int ch = 0; ch < 2; ++ch)                                                 \
                        acc[vb][ch] = __builtin_amdgcn_mfma_f32_32x32x2f32(Av[vb][q][e], Bv[ch][q][e], acc[vb][ch], 0, 0, 0);   \
        /* the 2 NVB + 4 fragment reads of the next tap (issued above in program order) go between this tap's 16 NVB MFMAs */  \
        _Pragma("unroll") for (int u = 0; u < 2 * NVB + 4; ++u) {                                                    \
            __builtin_amdgcn_sched_group_barrier(0x008, NVB == 4 ? 5 : 3, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                       \
        }                                                                                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 16 * NVB - (2 * NVB + 4) * (NVB == 4 ? 5 : 3), 0);              \
    } while (0)
// GroupNorm-apply instantiations: the MFMAs of tap (Av, Bv) in groups of GS; behind group u the u-th fragment read of the NEXT tap
// (k-group 0 of every block first: needed first) and slice SL0 + u of the rewrite; sched_barrier(0) keeps the groups apart
#define F9_MMX(Av, Bv, An, Bn, DORD, TAPN, WSN, TISN, SL0)                                                           \
    do {                                                                                                             \
        const char* wb_ = smem9 + 2 * HB + (WSN) * WSLOT + (TISN) * WTAP + bLane;                                    \
        _Pragma("unroll") for (int u = 0; u <= NG; ++u) {                                                            \
            _Pragma("unroll") for (int m = u * GS; m < (u == NG ? 16 * NVB : (u + 1) * GS); ++m) {                   \
                const int ch = m % 2, vb = (m / 2) % NVB, e = (m / (2 * NVB)) % 4, q = m / (8 * NVB);                \
                acc[vb][ch] = __builtin_amdgcn_mfma_f32_32x32x2f32(Av[vb][q][e], Bv[ch][q][e], acc[vb][ch], 0, 0, 0);   \
            }                                                                                                        \
            if (u < NG) {                                                                                            \
                if (DORD) {                                                                                          \
                    const int q = u / (NVB + 2), j = u % (NVB + 2);                                                  \
                    if (j < NVB) An[j][q] = *reinterpret_cast<const f32x4v*>(hb[j] + C::tapoff(TAPN) * ROWB + q * 32);   \
                    else Bn[j - NVB][q] = *reinterpret_cast<const f32x4v*>(wb_ + (j - NVB) * 32 * ROWB + q * 32);    \
                }                                                                                                    \
                gna_slice(s, (SL0) + u, ns, hcur ^ 1);                                                               \
            }                                                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                       \
        }                                                                                                            \
    } while (0)
            F9_RD(A0, B0, 0, wcur, 0);                     // cold read of the chunk's first tap (prefetched across chunks would be the next step)
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int wnext = wcur == NWS - 1 ? 0 : wcur + 1;
                const int wnn = wnext == NWS - 1 ? 0 : wnext + 1;
                // ---- this step's DMA: the weight group two steps ahead, two halo pieces of the next chunk (steps 0..7) ----
                {
                    const int s2 = s + 2 < NSTEP ? s + 2 : s + 2 - NSTEP;
                    const int c2 = s + 2 < NSTEP ? c : cNext;
                    const unsigned slot = wringBase + (unsigned)wnn * WSLOT;
                    dma_w(c2, 2 * s2, slot, 0);
                    if (2 * s2 + 1 < T) dma_w(c2, 2 * s2 + 1, slot, 1);
#pragma unroll
                    for (int r = 0; r < NPH; ++r)                                   // the next chunk's pieces, spread over all steps
                        if (r * NSPREAD / NPH == s) dma_h(r, hbufN, cNext);         // but the last (whose end is the chunk's end)
                }
                // ---- taps 2s, 2s + 1 ----
                if constexpr (GNA != 0) {
                    // the same taps with the interleave written out: per group of MFMAs one fragment read of the next tap and one
                    // slice of the rewrite of the pieces issued two steps ago (landed: the previous step's counted wait), fenced
                    constexpr int NG = 2 * NVB + 4, GS = NVB == 4 ? 5 : 3;
                    const int two = 2 * s + 1 < T ? 1 : 0, ns = NG * (1 + two);
                    if (two) {
                        F9_MMX(A0, B0, A1, B1, true, 2 * s + 1, wcur, 1, 0);
                        F9_MMX(A1, B1, A0, B0, s + 1 < NSTEP, 2 * s + 2, wnext, 0, NG);
                    } else {
                        F9_MMX(A0, B0, A1, B1, false, 0, wcur, 0, 0);
                    }
                } else if (2 * s + 1 < T) {
                    F9_RD(A1, B1, 2 * s + 1, wcur, 1);
                    F9_MM(A0, B0);
                    if (s + 1 < NSTEP) F9_RD(A0, B0, 2 * s + 2, wnext, 0);
                    F9_MM(A1, B1);
                } else {
                    F9_MM(A0, B0);
                }
                // End of the step.  The weight group issued at its start (a whole step ago; needed from the next step on) has to be
                // in the LDS, and at the end of a chunk the next chunk's halo image; the halo pieces issued IN this step (behind the
                // weights in program order: loads retire in order) may stay in flight for another step -- `vmcnt(0)` here made every
                // piece's HBM latency (~3 us under load) the step's problem.  Raw s_barrier: __syncthreads() would bring its own
                // vmcnt(0) lgkmcnt(0), which also drains the fragment reads prefetched for the next step.
                if (GNA != 0 && s + 1 == NSTEP) f9_chunk_end_gna();
                else f9_step_end(s + 1 == NSTEP ? 0 : nh_in_step(s));
                wcur = wnext;
            }
#undef F9_RD
#undef F9_MM
#undef F9_MMX
            hcur ^= 1;
        }
        // ---- epilogue of the tile: D[row = voxel][col = co]; voxel of (block vb, register i, lane half): plane bd, row bh4 + (i >> 2),
        //      column bw8 + (i & 3) + 4 hf ----
        {
            float cs0 = 0.f, cq0 = 0.f, cs1 = 0.f, cq1 = 0.f;
            // GroupNorm-backward epilogue: z = gA x + gB (the forward's fused GN + scale/shift), xhat = (x - gm) gr for this lane's two channels
            float gA0 = 0.f, gB0 = 0.f, gm0 = 0.f, gr0 = 0.f, gA1 = 0.f, gB1 = 0.f, gm1 = 0.f, gr1 = 0.f;
            int gact = 0;
            if constexpr (GNB) {
                const F9GnParams gp = *g.gnp;              // scalar loads, here and not before the main loop
                gact = gp.act;
                const int cpg = g.Cout / gp.G;
                auto coef = [&](int co, float& A, float& Bc, float& m, float& r) __attribute__((always_inline)) {
                    const int cc = min(co, g.Cout - 1);
                    m = gp.mean[tb * gp.G + cc / cpg]; r = gp.rstd[tb * gp.G + cc / cpg];
                    const float ga = gp.gamma ? gp.gamma[cc] : 1.f, be = gp.beta ? gp.beta[cc] : 0.f;
                    const float sc = gp.scale ? gp.scale[tb * gp.cs + cc] + 1.f : 1.f, sf = gp.shift ? gp.shift[tb * gp.cs + cc] : 0.f;
                    A = r * ga * sc;
                    Bc = (be - m * r * ga) * sc + sf;
                };
                coef(co0, gA0, gB0, gm0, gr0);
                coef(co1, gA1, gB1, gm1, gr1);
            }
            const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.gx), 0, GNB ? (int)g.yBytes : 0, 0x00020000);
#pragma unroll
            for (int vb = 0; vb < NVB; ++vb) {
                const int od = d0 + bd[vb];
                unsigned offs[16];
                float r0[16], r1[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int oh = h0 + bh4[vb] + (i >> 2), ow = w0 + bw8[vb] + (i & 3) + 4 * hf;
                    const bool ok = od < g.Do && oh < g.Ho && ow < g.Wo;
                    offs[i] = ok ? (unsigned)((((tb * g.Do + od) * g.Ho + oh) * g.Wo + ow) * g.Cout) * 4u : OOB;
                }
                // r0 / r1: the residual of a forward launch, or (GNB: a backward-data launch, which has none) the GroupNorm input at this
                // block's voxels -- one set of 32 registers either way; all loads in flight before the first use
                if constexpr (GNB) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        r0[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, offs[i] + c0o, 0, 0));
                        r1[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, offs[i] + c1o, 0, 0));
                    }
                } else if (residual) {     // kernel-uniform
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        r0[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, offs[i] + c0o, 0, 0));
                        r1[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_r, offs[i] + c1o, 0, 0));
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v0 = acc[vb][0][i] + bias0, v1 = acc[vb][1][i] + bias1;
                    if constexpr (!GNB) { if (residual) { v0 += r0[i]; v1 += r1[i]; } }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs_y, offs[i] + c0o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), rs_y, offs[i] + c1o, 0, 0);
                    if constexpr (GNB) {
                        if (offs[i] != OOB) {      // out-of-range loads returned 0 and out-of-range channels are never read back
                            const float dz0 = v0 * f9_act_grad(gA0 * r0[i] + gB0, gact), dz1 = v1 * f9_act_grad(gA1 * r1[i] + gB1, gact);
                            cs0 += dz0; cq0 = fmaf(dz0, (r0[i] - gm0) * gr0, cq0);
                            cs1 += dz1; cq1 = fmaf(dz1, (r1[i] - gm1) * gr1, cq1);
                        }
                    } else if (g.stats && offs[i] != OOB) { cs0 += v0; cq0 = fmaf(v0, v0, cq0); cs1 += v1; cq1 = fmaf(v1, v1, cq1); }
                    acc[vb][0][i] = 0.f; acc[vb][1][i] = 0.f;
                }
            }
            if (g.stats) {         // kernel-uniform: fixed-order combine of the lane halves, then of the 4 waves through LDS
                cs0 += __shfl_xor(cs0, 32, 64); cq0 += __shfl_xor(cq0, 32, 64);
                cs1 += __shfl_xor(cs1, 32, 64); cq1 += __shfl_xor(cq1, 32, 64);
                // scratch: the halo image that is NOT being filled (the next tile's first chunk lands in image hcur)
                float* red = reinterpret_cast<float*>(smem9 + (hcur ^ 1) * HB);
                if (hf == 0) {
                    red[(wave * 4 + 0) * 32 + l31] = cs0; red[(wave * 4 + 1) * 32 + l31] = cq0;
                    red[(wave * 4 + 2) * 32 + l31] = cs1; red[(wave * 4 + 3) * 32 + l31] = cq1;
                }
                __syncthreads();
                if (tid < 128) {   // q = tid >> 5: 0 sum(co0) 1 sumsq(co0) 2 sum(co1) 3 sumsq(co1)
                    const int q = tid >> 5, l = tid & 31;
                    const float v = ((red[q * 32 + l] + red[(4 + q) * 32 + l]) + red[(8 + q) * 32 + l]) + red[(12 + q) * 32 + l];
                    const int co = n0 + (q >> 1) * 32 + l;
                    const int tpb = g.tilesD * g.tilesH * g.tilesW, mtile = (int)(L / g.nNt);
                    if (co < g.Cout) g.stats[(((size_t)(mtile / tpb) * tpb + mtile % tpb) * 2 + (q & 1)) * g.Cout + co] = v;
                }
                __syncthreads();   // the scratch is free again before the next chunk's pieces may land in it
            }
        }
        L += Gn;
        if (!lastTile) tile_of(L, tb, d0, h0, w0);
    }
}

template <class C, bool GNB = false, int GNA = 0> static int f9_launch(const float* x, const float* packed, const float* bias, const float* residual, float* y,
                                        const F9Geom& g, size_t lds, unsigned grid, void* stream) {
    auto kern = conv_fwd9_kernel<C, GNB, GNA>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_fwd(v9): hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(grid, g.ksplit), dim3(256), lds, (hipStream_t)stream, x, packed, bias, residual, y, g);
    return check_launch("conv3d_fwd(v9)");
}


// the variants (filter, tile, voxel blocks per wave); F9Geom::variant indexes this list
using F9_333_512 = F9Cfg<3, 3, 3, 8, 8, 8, 4>;
using F9_333_256 = F9Cfg<3, 3, 3, 4, 8, 8, 2>;
using F9_133_A = F9Cfg<1, 3, 3, 1, 16, 32, 4>;       // a 32-wide frame row pair per block row
using F9_133_B = F9Cfg<1, 3, 3, 2, 16, 16, 4>;
using F9_133_C = F9Cfg<1, 3, 3, 4, 8, 8, 2>;         // 8x8 frames: 256-voxel tiles
using F9_311_512 = F9Cfg<3, 1, 1, 8, 8, 8, 4>;       // temporal convs of the pseudo-3D blocks
using F9_311_256 = F9Cfg<3, 1, 1, 4, 8, 8, 2>;

// launchers of the variants compiled in the other translation units
int fwd9_launch_b(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream);
int fwd9_launch_c(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream);
// ... and of the instantiations with the GroupNorm-backward epilogue (g.gx set)
int fwd9_launch_d(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream);
int fwd9_launch_e(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream);
// ... and with the GroupNorm-apply prologue (g.gcoef set): Mish on the 3x3x3 variants (Family A), SiLU on the (1,3,3) ones (Family B)
int fwd9_launch_f(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream);
int fwd9_launch_g(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream);
int fwd9_launch_h(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream);
int fwd9_launch_i(const float* x, const float* packed, const float* bias, const float* residual, float* y, const F9Geom& g, size_t lds,
                  unsigned grid, void* stream);

}  // namespace diqt
