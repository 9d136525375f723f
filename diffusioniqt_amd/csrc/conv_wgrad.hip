// Weight gradient of the stride-1 3-D convolution for gfx950, version 3 (conv_wgrad3_kernel):
//   dW[co][ci][tap] = sum_v dY[v][co] * X[v + tap][ci]        M = co, N = ci, K = voxels, on v_mfma_f32_32x32x2_f32 (exact fp32)
//
// What bounded version 2 (conv_bwd_weight2_kernel, conv_mfma.hip) at 0.58 of the f32 MFMA peak: two waves per SIMD in lockstep
// sharing the matrix pipe (the older wave finishes ~20 % early and waits at the tile barrier), register-staged tiles whose
// global -> register -> LDS passes and table fills sit between two barriers with the pipe idle, and 7 accumulators per wave
// (8 LDS reads per 7 MFMAs).  This kernel is built the other way round:
//   * ONE wave per SIMD (256 threads, one workgroup per CU) with up to 14 accumulator tiles (224 AGPRs) per wave: a wave's MFMAs
//     are limited only by its own LDS reads, which are issued one k-step ahead (ping-pong operand registers);
//   * the 64-voxel X halo tile and dY tile are DOUBLE-BUFFERED in LDS and filled by LDS-DMA (`buffer_load_dwordx4 ... lds`,
//     1 KiB per wave-instruction, no VGPR staging, no per-tile tables): the pieces of tile t+1 are issued one per pair of k-steps
//     INSIDE the k-loop of tile t, so a tile boundary is `s_waitcnt vmcnt(0)` + one barrier;
//   * zero padding, ragged tiles and ragged channel counts come from the buffer descriptor's range check (an out-of-range
//     LDS-DMA lane writes zeros: tools/probes/lds_dma_probe.hip), selected per lane from tile-independent packed coordinates
//     that live in registers for the whole kernel;
//   * filter and tile extents are template parameters: every LDS offset of the fully unrolled 32 k-steps of a tile is an
//     instruction immediate (no address arithmetic in the loop);
//   * the bias gradient rides on the A operand a wave reads anyway (one v_add per k-step in the workgroups of chunk 0).
// MODE 0 (27 taps): the 4 waves are (co half) x (tap half: 13.5 taps each -- the middle tap alternates between the halves from
// k-step to k-step) over a 64 co x 32 ci block;
// MODE 1 (<= 14 taps): (co half) x (ci half), all taps per wave, over a 64 co x 64 ci block.
// Partial slabs go to the workspace in the packed layout of version 2 and are summed in a fixed order by
// conv_reduce_dw_kernel (deterministic).  Reference call sites: every nn.Conv3d backward of the U-Nets
// (/root/reference/imagen_pytorch3D.py:535-566 Block.project, imagen_video.py:352-406).
#include "common.h"
#include "conv_wgrad.h"
#include <stdlib.h>

namespace diqt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA piece: 64 lanes x 16 B from per-lane byte offsets `voff` of the buffer `rs` to LDS bytes [lds, lds + 1024).
// Inline asm because M0 (the LDS destination) must be written right BEFORE the load: with the builtin hipcc hoists the NEXT
// piece's `s_add m0` to just behind the load, where the write waits until the load has read M0 -- ~120 cycles during which
// this wave issues no MFMA (profiles/r02_wgrad_ablation.md).  The statement has no VGPR result; completion is counted by hand
// (`s_waitcnt vmcnt(0)` before the tile barrier).
__device__ __forceinline__ void w3_dma_piece(i32x4 rs, unsigned lds, unsigned voff) {
#ifdef W3_ASM_DMA
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds), "v"(voff), "s"(rs) : "memory");
#else
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((size_t)(unsigned)rs[1] << 32) | (unsigned)rs[0]), 0, rs[2], rs[3]);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(size_t)lds, 16, voff, 0, 0, 0);
#endif
}

constexpr unsigned W3_OOB = 0x80000000u;
constexpr int W3_MTV = 64;           // voxels per tile
constexpr int W3_NS = W3_MTV / 2;    // k-steps per tile (an MFMA consumes 2 voxels)
constexpr int W3_YB = W3_MTV * 64 * 4;   // dY tile: 64 voxels x 64 co

template <int KD_, int KH_, int KW_, int TD_, int TH_, int TW_, int MODE_>
struct W3Cfg {
    static constexpr int KD = KD_, KH = KH_, KW = KW_, TD = TD_, TH = TH_, TW = TW_, MODE = MODE_;
    static constexpr int T = KD * KH * KW;
    static constexpr int HD = TD + KD - 1, HH = TH + KH - 1, HWd = TW + KW - 1, HV = HD * HH * HWd;
    static constexpr int CIW = MODE ? 64 : 32;                 // input channels staged per workgroup
    static constexpr int ROWB = CIW * 4;                       // bytes per halo voxel row
    static constexpr int XB = (HV * ROWB + 4095) / 4096 * 4096;   // X halo image: whole 1-KiB DMA instructions, the same count for each of the 4 waves
    static constexpr int BUFB = XB + W3_YB;
    static constexpr int NXI = XB / 1024, NPX = NXI / 4;        // X DMA instructions per tile / per wave
    static constexpr int NPY = 4;                              // dY: 16 instructions per tile
    static constexpr int TA = MODE ? T : (T + 1) / 2;          // taps of the first tap half
    static_assert(TD * TH * TW == W3_MTV && (TW % 2) == 0, "tile: 64 voxels, even along W");
    static_assert(NPX + NPY <= W3_NS / 2, "one DMA piece per pair of k-steps");
    static_assert(TA <= 16, "accumulators live in the 256 AGPRs");
    static_assert((HV - 1) * ROWB + ROWB <= 65536, "LDS immediates are 16 bit");
};

// halo index of voxel v (v = (td * TH + th) * TW + tw) at tap (0,0,0)
template <class C> __host__ __device__ constexpr int w3_hidx(int v) {
    return ((v / (C::TW * C::TH)) * C::HH + (v / C::TW) % C::TH) * C::HWd + v % C::TW;
}
template <class C> __host__ __device__ constexpr int w3_tapoff(int tap) {
    return ((tap / (C::KW * C::KH)) * C::HH + (tap / C::KW) % C::KH) * C::HWd + tap % C::KW;
}

// One wave's whole life.  PATH 2 (MODE 1): all T taps on every k-step.  PATH 0 / 1 (MODE 0, T odd, TH = T / 2): the two tap halves
// of a co half.  An odd tap count would leave one half with an extra MFMA per k-step (14 vs 13: the lighter waves idle 7 % of
// the time), so the middle tap TH is SHARED: on even k-steps it belongs to PATH 0 (taps 0..TH | TH+1..T-1), on odd k-steps to PATH 1
// (taps 0..TH-1 | TH..T-1).  Both paths issue T MFMAs per pair of k-steps; each keeps a partial sum of tap TH, added once at the end.
template <class C, int PATH>
__device__ __forceinline__ void w3_wave(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slabs,
                                        float* __restrict__ bias_part, const W3Geom& g, char* smem, int cq, int ciq) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // ---- block roles ----
    int bx_ = blockIdx.x;
    const int cb = bx_ % g.nCoB; bx_ /= g.nCoB;
    const int cib = bx_;                                   // input-channel block (CIW wide)
    const int ci0 = cib * C::CIW, n0 = cb * 64;
    const int mtBegin = blockIdx.y * g.tilesPerSplit, mtEnd = min(mtBegin + g.tilesPerSplit, g.MT);

    // buffer descriptors as plain SGPR quads for the inline-asm LDS-DMA below: {base lo, base hi (stride 0), bytes, flags}
    const i32x4 rs_x = {(int)(unsigned)(size_t)x, (int)((size_t)x >> 32) & 0xffff, (int)g.xBytes, 0x00020000};
    const i32x4 rs_y = {(int)(unsigned)(size_t)dy, (int)((size_t)dy >> 32) & 0xffff, (int)g.yBytes, 0x00020000};
    const unsigned ldsBase = (unsigned)(size_t)(lds_void*)smem;

    // ---- tile-independent description of this lane's DMA pieces (registers for the whole kernel) ----
    unsigned posX[C::NPX], relX[C::NPX], posY[C::NPY], relY[C::NPY];
#pragma unroll
    for (int r = 0; r < C::NPX; ++r) {
        const int j = wave + 4 * r, p = j * 64 + lane;
        const int row = p / (C::CIW / 4), c4 = (p % (C::CIW / 4)) * 4;
        const int hx = row % C::HWd, hy = (row / C::HWd) % C::HH, hz = row / (C::HWd * C::HH);
        const bool ok = row < C::HV && ci0 + c4 < g.Cin;
        posX[r] = (unsigned)hz | ((unsigned)hy << 8) | ((unsigned)hx << 16) | (ok ? 0u : 1u << 24);
        relX[r] = (unsigned)(((hz * g.H + hy) * g.W + hx) * g.Cin + c4) * 4u;
    }
#pragma unroll
    for (int r = 0; r < C::NPY; ++r) {
        const int p = (wave + 4 * r) * 64 + lane;
        const int v = p >> 4, c4 = (p & 15) * 4;
        const int tw = v % C::TW, th = (v / C::TW) % C::TH, td = v / (C::TW * C::TH);
        const bool ok = n0 + c4 < g.Cout;
        posY[r] = (unsigned)td | ((unsigned)th << 8) | ((unsigned)tw << 16) | (ok ? 0u : 1u << 24);
        relY[r] = (unsigned)(((td * g.Ho + th) * g.Wo + tw) * g.Cout + c4) * 4u;
    }

    // ---- tile walk state (wave-uniform, incremental: no division per tile) ----
    int tx, ty, tz, tb;
    {
        int mt = mtBegin;
        tx = mt % g.tilesW; mt /= g.tilesW;
        ty = mt % g.tilesH; mt /= g.tilesH;
        tz = mt % g.tilesD; tb = mt / g.tilesD;
    }
    auto advance_tile = [&]() {
        if (++tx == g.tilesW) { tx = 0; if (++ty == g.tilesH) { ty = 0; if (++tz == g.tilesD) { tz = 0; ++tb; } } }
    };
    // scalars of the tile whose DMA is being issued
    int bz, by, bxx, d0, h0, w0;
    unsigned baseX, baseY;
    auto set_dma_tile = [&]() {
        d0 = tz * C::TD; h0 = ty * C::TH; w0 = tx * C::TW;
        bz = d0 - g.pd; by = h0 - g.ph; bxx = w0 - g.pw;
        baseX = (unsigned)((((tb * g.D + bz) * g.H + by) * g.W + bxx) * g.Cin + ci0) * 4u;
        baseY = (unsigned)((((tb * g.Do + d0) * g.Ho + h0) * g.Wo + w0) * g.Cout + n0) * 4u;
    };
    // The issue is UNCONDITIONAL (no branch inside the k-loop, which has to stay one basic block for the instruction scheduler)
    // and the offset arithmetic is pure VALU -- no v_cmp -> s_and -> v_cndmask chain through scalar registers, which cost ~150
    // cycles of MFMA issue per piece (profiles/r02_wgrad_ablation.md): a coordinate c is inside [0, N) iff neither c nor N-1-c is
    // negative, so the OR of all six terms carries the "outside" verdict in its sign bit, and that bit, OR-ed into the byte offset,
    // puts the lane beyond the descriptor's range (< 2^30 bytes).  `dead` = 0x80000000 for a piece that is never loaded (row /
    // channel beyond the tile) or when there is no next tile: the piece then writes zeros into the idle buffer.
    const int Dm1 = g.D - 1, Hm1 = g.H - 1, Wm1 = g.W - 1, Dom1 = g.Do - 1, Hom1 = g.Ho - 1, Wom1 = g.Wo - 1;
    auto dma_x = [&](int r, unsigned buf, unsigned dead) {   // r static; buf = LDS byte address of the destination buffer
        const unsigned p = posX[r];
        const int iz = bz + (int)(p & 255u), iy = by + (int)((p >> 8) & 255u), ix = bxx + (int)((p >> 16) & 255u);
        const unsigned m = (unsigned)(iz | iy | ix) | (unsigned)((Dm1 - iz) | (Hm1 - iy) | (Wm1 - ix)) | (p << 7) | dead;   // bit 24 of p = "never valid"
        const unsigned voff = (baseX + relX[r]) | (m & 0x80000000u);
        w3_dma_piece(rs_x, buf + (unsigned)(wave + 4 * r) * 1024u, voff);
    };
    auto dma_y = [&](int r, unsigned buf, unsigned dead) {
        const unsigned p = posY[r];
        const int od = d0 + (int)(p & 255u), oh = h0 + (int)((p >> 8) & 255u), ow = w0 + (int)((p >> 16) & 255u);
        const unsigned m = (unsigned)((Dom1 - od) | (Hom1 - oh) | (Wom1 - ow)) | (p << 7) | dead;
        const unsigned voff = (baseY + relY[r]) | (m & 0x80000000u);
        w3_dma_piece(rs_y, buf + (unsigned)(C::XB + (wave + 4 * r) * 1024), voff);
    };

    constexpr int T = C::T, TH = T / 2;
    static_assert(PATH == 2 || (T & 1), "the shared-tap split is for odd tap counts");
    // taps [E_T0, E_T0 + E_N) on even k-steps into acc[E_AO + t], [O_T0, O_T0 + O_N) on odd ones into acc[O_AO + t]
    constexpr int E_T0 = PATH == 1 ? TH + 1 : 0, E_N = PATH == 2 ? T : (PATH == 0 ? TH + 1 : T - TH - 1), E_AO = PATH == 1 ? 1 : 0;
    constexpr int O_T0 = PATH == 1 ? TH : 0, O_N = PATH == 2 ? T : (PATH == 0 ? TH : T - TH), O_AO = 0;
    constexpr int NACC = PATH == 2 ? T : TH + 1;          // accumulator i holds tap ACC_T0 + i
    constexpr int ACC_T0 = PATH == 1 ? TH : 0;
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const bool doBias = bias_part != nullptr && cib == 0 && PATH != 1 && ciq == 0;
    float bsum = 0.f;

    // diagnostic stamps (DIQT_CONV_DBG=1): taken at tile boundaries only, outside the unrolled k-loop block
    const bool dbg = g.dbg != nullptr;
    long long tK = 0, tB = 0, tStart = dbg ? (long long)__builtin_readcyclecounter() : 0, tPro = 0;
    const unsigned long long rt0 = dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    if (mtBegin < mtEnd) {
        // ---- prologue: tile 0 -> buffer 0 ----
        set_dma_tile();
#pragma unroll
        for (int r = 0; r < C::NPX; ++r) dma_x(r, ldsBase, 0u);
#pragma unroll
        for (int r = 0; r < C::NPY; ++r) dma_y(r, ldsBase, 0u);
        advance_tile();
        __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0)
        __syncthreads();

        // per-lane operand bases inside a buffer
        const int aLane = (h * 64 + cq * 32 + l31) * 4 + C::XB;
        const int bLane = (h * C::CIW + ciq * 32 + l31) * 4;
        int cur = 0;
        if (dbg) tPro = (long long)__builtin_readcyclecounter() - tStart;
        for (int mt0 = mtBegin; mt0 < mtEnd; ++mt0) {
            const long long tt0 = dbg ? (long long)__builtin_readcyclecounter() : 0;
            const bool haveNext = mt0 + 1 < mtEnd;
#ifdef W3_DEADDMA
            const unsigned deadNext = 0x80000000u;
#else
            const unsigned deadNext = haveNext ? 0u : 0x80000000u;
#endif
            const char* cbuf = smem + cur * C::BUFB;
            const unsigned nbuf = ldsBase + (unsigned)(cur ^ 1) * C::BUFB;
            set_dma_tile();                               // (the tile after the last one is never live)
            const float* ap = reinterpret_cast<const float*>(cbuf + aLane);
            const float* bp = reinterpret_cast<const float*>(cbuf + bLane);
            float a0, a1, b0[E_N], b1[O_N];               // ping-pong operand registers: even / odd k-steps
            // operands of k-step s: A = dY[2s + h][co], B[t] = X[voxel 2s + h + tap t][ci]
// An operand of tap (kz, ky, kx) at k-step S is the LDS word of tap (kz, ky, kx + 2) at step S - 1 when both steps lie in the same
// W row of the tile (the voxel pair moved 2 along W): such operands are taken over in registers (about a fifth fewer LDS reads).
#define W3_RD(S, A, Bv, T0, N, Bp, PT0, PN)                                                                          \
    do {                                                                                                             \
        A = ap[(S) * 128];                                                                                           \
        _Pragma("unroll") for (int t = 0; t < (N); ++t) {                                                            \
            const int tap = (T0) + t;                                                                                \
            const bool reuse = (S) > 0 && (2 * (S)) % C::TW != 0 && tap % C::KW + 2 < C::KW && tap + 2 >= (PT0) && tap + 2 < (PT0) + (PN);   \
            if (reuse) Bv[t] = Bp[tap + 2 - (PT0) < (PN) && tap + 2 - (PT0) >= 0 ? tap + 2 - (PT0) : 0];            \
            else Bv[t] = bp[(w3_hidx<C>(2 * (S)) + w3_tapoff<C>(tap)) * C::CIW];                                     \
        }                                                                                                            \
    } while (0)
#define W3_MM(A, Bv, AO, N, NVMEM)                                                                                   \
    do {                                                                                                             \
        _Pragma("unroll") for (int t = 0; t < (N); ++t)                                                              \
            acc[(AO) + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A, Bv[t], acc[(AO) + t], 0, 0, 0);                  \
        bsum += A;                                                                                                   \
        /* Issue order of the step: behind every MFMA one LDS read of the NEXT k-step (issued above in program order) and at    \
           most two VALU instructions (the offset arithmetic of the next DMA pieces: left alone, hipcc packs a piece's whole     \
           dependent chain into one MFMA gap and the matrix pipe waits ~90 cycles for it); the DMA piece itself behind the       \
           first MFMA of the step that carries one. */                                                                         \
        _Pragma("unroll") for (int t = 0; t < (N); ++t) {                                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                       \
            if (t == 0 && (NVMEM)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                          \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                       \
        }                                                                                                            \
    } while (0)
            W3_RD(0, a0, b0, E_T0, E_N, b1, O_T0, 0);
#pragma unroll
            for (int pr = 0; pr < W3_NS / 2; ++pr) {
                // one DMA piece of the next tile per pair of k-steps; the 4 waves run in lockstep behind the tile barrier, so the
                // two tap halves issue theirs half a pair apart (the texture path takes the pieces one after the other)
                if (PATH != 1) {
                    if (pr < C::NPX) dma_x(pr, nbuf, deadNext);
                    else if (pr < C::NPX + C::NPY) dma_y(pr - C::NPX, nbuf, deadNext);
                }
                W3_RD(2 * pr + 1, a1, b1, O_T0, O_N, b0, E_T0, E_N);
                W3_MM(a0, b0, E_AO, E_N, (PATH != 1 && pr < C::NPX + C::NPY) ? 1 : 0);
                if (PATH == 1) {
                    if (pr < C::NPX) dma_x(pr, nbuf, deadNext);
                    else if (pr < C::NPX + C::NPY) dma_y(pr - C::NPX, nbuf, deadNext);
                }
                if (pr + 1 < W3_NS / 2) W3_RD(2 * pr + 2, a0, b0, E_T0, E_N, b1, O_T0, O_N);
                W3_MM(a1, b1, O_AO, O_N, (PATH == 1 && pr < C::NPX + C::NPY) ? 1 : 0);
            }
#undef W3_RD
#undef W3_MM
            if (haveNext) advance_tile();
            const long long tt1 = dbg ? (long long)__builtin_readcyclecounter() : 0;
            __builtin_amdgcn_s_waitcnt(0x0f70);           // this wave's pieces of the next tile have landed
            __syncthreads();                              // ... everybody's have, and everybody is done reading the current buffer
            if (dbg) { const long long tt2 = (long long)__builtin_readcyclecounter(); tK += tt1 - tt0; tB += tt2 - tt1; }
            cur ^= 1;
        }
    }

    const long long tLoopEnd = dbg ? (long long)__builtin_readcyclecounter() : 0;
    // ---- bias gradient partial: sum over the two voxel parities (lane halves), one row per split-K slice ----
    if (doBias) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (h == 0 && n0 + cq * 32 + l31 < g.CoutPad) bias_part[(size_t)blockIdx.y * g.CoutPad + n0 + cq * 32 + l31] = bsum;
    }
    // ---- the shared tap: PATH 1's partial sum (acc[0]) joins PATH 0's (acc[TH]) through LDS, lane for lane ----
    if constexpr (PATH != 2) {
        float* xch = reinterpret_cast<float*>(smem) + cq * (16 * 64);      // the tile buffers are idle (last barrier of the walk)
        if (PATH == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) xch[r * 64 + lane] = acc[0][r];
        }
        __syncthreads();
        if (PATH == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[TH][r] += xch[r * 64 + lane];
        }
    }
    // ---- slab, written in the FINAL dw[co][ci][tap] order (the split-K reduce is then a pure streaming sum with contiguous stores;
    //      reducing the packed layout of version 2 scatters 4-byte stores over every line of dW: 29 of its 38 us).  The accumulators
    //      D[row = co][col = ci] (row = (r & 3) + 8 * (r >> 2) + 4 * h) of one co half at a time are transposed through LDS into
    //      [32 co][CIW ci][T taps] and leave as 16-byte stores: a co row is one contiguous run of (ci block) x T floats. ----
    {
        float* st = reinterpret_cast<float*>(smem);
        float* slab = slabs + (size_t)blockIdx.y * g.Cout * g.Cin * T;
        const int nci = min(C::CIW, g.Cin - ci0);                         // input channels of this block that exist (a multiple of 4)
        const int rowQ = nci * T / 4;                                     // float4 per co row
        for (int half = 0; half < 2; ++half) {
            __syncthreads();                                              // the staging area is free (tile buffers / exchange / previous half)
            if (cq == half) {
#pragma unroll
                for (int t = (PATH == 1 ? 1 : 0); t < NACC; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        st[(((r & 3) + 8 * (r >> 2) + 4 * h) * C::CIW + ciq * 32 + l31) * T + ACC_T0 + t] = acc[t][r];
            }
            __syncthreads();
            for (int idx = tid; idx < 32 * rowQ; idx += 256) {
                const int row = idx / rowQ, q = idx - row * rowQ;
                const int co = n0 + half * 32 + row;
                if (co < g.Cout)
                    *reinterpret_cast<float4*>(slab + ((size_t)co * g.Cin + ci0) * T + 4 * q) =
                        *reinterpret_cast<const float4*>(st + (size_t)row * C::CIW * T + 4 * q);
            }
        }
    }
    if (dbg && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0x0f70);
        unsigned long long* o = g.dbg + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        const long long tEnd = (long long)__builtin_readcyclecounter();
        o[0] = (unsigned long long)(tEnd - tStart); o[1] = (unsigned long long)tK; o[2] = (unsigned long long)tB;
        o[3] = (unsigned long long)tPro; o[4] = (unsigned long long)(tEnd - tLoopEnd);
        o[5] = __builtin_amdgcn_s_memrealtime() - rt0; o[6] = (unsigned long long)(E_N + O_N); o[7] = (unsigned long long)(mtEnd - mtBegin);
    }
}

template <class C>
__global__ __launch_bounds__(256, 1) void conv_wgrad3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ slabs, float* __restrict__ bias_part, W3Geom g) {
    extern __shared__ __attribute__((aligned(1024))) char smem3[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cq = wave & 1, q2 = wave >> 1;
    if constexpr (C::MODE == 0) {
        if (q2 == 0) w3_wave<C, 0>(x, dy, slabs, bias_part, g, smem3, cq, 0);
        else w3_wave<C, 1>(x, dy, slabs, bias_part, g, smem3, cq, 0);
    } else {
        w3_wave<C, 2>(x, dy, slabs, bias_part, g, smem3, cq, q2);
    }
}

using W3_333 = W3Cfg<3, 3, 3, 2, 4, 8, 0>;
using W3_133 = W3Cfg<1, 3, 3, 1, 8, 8, 1>;
using W3_311 = W3Cfg<3, 1, 1, 8, 2, 4, 1>;
using W3_111 = W3Cfg<1, 1, 1, 2, 4, 8, 1>;       // pointwise convs / Linear layers: dW = dY^T X over all rows, taken as 64-row tiles

template <class C> static bool w3_fill(W3Geom& g, int B, int D, int H, int W, int Cin, int Cout, int pd, int ph, int pw, int epd,
                                       int eph, int epw, int& ksplit, size_t& lds, int wgs) {
    g.B = B; g.D = D; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout;
    g.Do = D + 2 * pd + epd - C::KD + 1; g.Ho = H + 2 * ph + eph - C::KH + 1; g.Wo = W + 2 * pw + epw - C::KW + 1;
    if (g.Do <= 0 || g.Ho <= 0 || g.Wo <= 0) return false;
    g.pd = pd; g.ph = ph; g.pw = pw; g.dbg = nullptr;
    g.tilesD = (g.Do + C::TD - 1) / C::TD; g.tilesH = (g.Ho + C::TH - 1) / C::TH; g.tilesW = (g.Wo + C::TW - 1) / C::TW;
    g.nCoB = (Cout + 63) / 64; g.CoutPad = g.nCoB * 64; g.nChunks32 = (Cin + 31) / 32;
    g.nCiB = (Cin + C::CIW - 1) / C::CIW;
    const long long mt = (long long)B * g.tilesD * g.tilesH * g.tilesW;
    if (mt >= (1ll << 30)) return false;
    g.MT = (int)mt;
    const unsigned long long xb = (unsigned long long)B * D * H * W * Cin * 4ull, yb = (unsigned long long)B * g.Do * g.Ho * g.Wo * Cout * 4ull;
    if (xb >= (1ull << 30) || yb >= (1ull << 30)) return false;       // 32-bit buffer offsets with an out-of-range sentinel
    g.xBytes = (unsigned)xb; g.yBytes = (unsigned)yb;
    const int gx = g.nCoB * g.nCiB;
    ksplit = wgs / gx;                                                 // one workgroup per CU, one resident round
    if (ksplit > g.MT) ksplit = g.MT;
    if (ksplit < 1) ksplit = 1;
    g.tilesPerSplit = (g.MT + ksplit - 1) / ksplit;
    ksplit = (g.MT + g.tilesPerSplit - 1) / g.tilesPerSplit;
    lds = 2 * (size_t)C::BUFB;
    const size_t stage = (size_t)32 * C::CIW * C::T * sizeof(float);          // epilogue transpose of one co half
    if (stage > lds) lds = stage;
    return lds <= 160 * 1024;
}

unsigned long long* wgrad3_dbg_ptr = nullptr;
unsigned wgrad3_dbg_n = 0;

// which instantiation takes the filter (0 none)
static int w3_variant(int kd, int kh, int kw) {
    if (kd == 3 && kh == 3 && kw == 3) return 1;
    if (kd == 1 && kh == 3 && kw == 3) return 2;
    if (kd == 3 && kh == 1 && kw == 1) return 3;
    if (kd == 1 && kh == 1 && kw == 1) return 4;
    return 0;
}

bool wgrad3_plan(W3Geom& g, int& variant, int& ksplit, size_t& lds, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh,
                 int kw, int pd, int ph, int pw, int epd, int eph, int epw) {
    static const int mode = [] { const char* e = getenv("DIQT_BWDW_V3"); return e ? atoi(e) : 1; }();     // 0: never
    static const int wgs = [] { const char* e = getenv("DIQT_BWDW_WGS"); return e ? atoi(e) : 256; }();
    variant = mode ? w3_variant(kd, kh, kw) : 0;
    if (!variant || Cin % 4 != 0 || Cout % 4 != 0 || Cin < 16) return false;
    if (variant == 4) {
        // a pointwise filter sees rows, not a volume: any [B, D, H, W] (Linear layers arrive as one long row axis) is re-cut into
        // V / 64 "batch entries" of one 2 x 4 x 8 tile each
        static const bool pw3 = [] { const char* e = getenv("DIQT_PW_V3"); return !(e && e[0] == '0'); }();
        const long long V = (long long)B * D * H * W;
        if (!pw3 || pd || ph || pw || epd || eph || epw || V % 64 != 0 || V / 64 >= (1ll << 30)) return false;
        return w3_fill<W3_111>(g, (int)(V / 64), 2, 4, 8, Cin, Cout, 0, 0, 0, 0, 0, 0, ksplit, lds, wgs);
    }
    if (D > 255 || H > 255 || W > 255) return false;                   // packed 8-bit tile coordinates
    switch (variant) {
        case 1: return w3_fill<W3_333>(g, B, D, H, W, Cin, Cout, pd, ph, pw, epd, eph, epw, ksplit, lds, wgs);
        case 2: return w3_fill<W3_133>(g, B, D, H, W, Cin, Cout, pd, ph, pw, epd, eph, epw, ksplit, lds, wgs);
        default: return w3_fill<W3_311>(g, B, D, H, W, Cin, Cout, pd, ph, pw, epd, eph, epw, ksplit, lds, wgs);
    }
}

template <class C> static int w3_launch(const float* x, const float* dy, float* slabs, float* bias_part, const W3Geom& g, int ksplit,
                                        size_t lds, hipStream_t s) {
    auto kern = conv_wgrad3_kernel<C>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIQT_REQUIRE(e == hipSuccess, DIQT_E_LAUNCH, "conv3d_bwd_weight(v3): hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    W3Geom gg = g;
    static const bool dbg_on = [] { const char* e = getenv("DIQT_CONV_DBG"); return e && e[0] == '1'; }();
    static unsigned long long* dbuf = nullptr;
    gg.dbg = nullptr;
    if (dbg_on) {
        const size_t n = (size_t)g.nCoB * g.nCiB * ksplit * 4;
        if (!dbuf) (void)hipMalloc(&dbuf, (size_t)65536 * 8 * sizeof(unsigned long long));
        if (n <= 65536) { gg.dbg = dbuf; wgrad3_dbg_ptr = dbuf; wgrad3_dbg_n = (unsigned)n; }
    }
    hipLaunchKernelGGL(kern, dim3(g.nCoB * g.nCiB, ksplit), dim3(256), lds, s, x, dy, slabs, bias_part, gg);
    return check_launch("conv3d_bwd_weight(v3)");
}

int wgrad3_launch(const float* x, const float* dy, float* slabs, float* bias_part, const W3Geom& g, int variant, int ksplit, size_t lds,
                  void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (variant) {
        case 1: return w3_launch<W3_333>(x, dy, slabs, bias_part, g, ksplit, lds, s);
        case 2: return w3_launch<W3_133>(x, dy, slabs, bias_part, g, ksplit, lds, s);
        case 3: return w3_launch<W3_311>(x, dy, slabs, bias_part, g, ksplit, lds, s);
        case 4: return w3_launch<W3_111>(x, dy, slabs, bias_part, g, ksplit, lds, s);
    }
    set_error("conv3d_bwd_weight(v3): no variant");
    return DIQT_E_UNSUPPORTED;
}

}  // namespace diqt
