/*
 * diqt.h — C ABI of libdiqt_hip.so: the MI355X (gfx950) device kernels behind the
 * DiffusionIQT hot path (3-D conditional-diffusion U-Net forward/backward, DDPM/EDM sampler
 * step, loss, optimiser step).
 *
 * Drop-in boundary: the reference has no native layer — every FLOP is an ATen op
 * called from imagen_pytorch3D.py / imagen_video.py / elucidated_imagen.py / trainer.py.  Each entry
 * point below names the reference call site(s) (file:line, relative to the reference root) whose
 * ATen op(s) it replaces.  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only: device pointers, ints, floats; `stream` is a hipStream_t passed as void*
 *     (NULL = the null stream).  Every call is asynchronous on `stream`, holds no pointer past the
 *     completion of the work it enqueues and keeps no global mutable state.
 *   - activations are fp32, channels-last ("NDHWC"): x[b][d][h][w][c], c fastest.
 *     `rows` = b*d*h*w when an op does not care about the spatial structure.
 *   - weights cross the boundary in the reference's own layout (OIDHW for Conv3d, [out][in] for
 *     Linear); diqt_conv_pack_weight re-lays them for the MFMA kernels.
 *   - every entry returns DIQT_OK (0) or a negative DIQT_E_* code; it never throws, never exits.
 *     diqt_last_error() returns a thread-local message for the most recent failure.
 */
#ifndef DIQT_H
#define DIQT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIQT_OK             0
#define DIQT_E_SHAPE       -1   /* an extent is <= 0 or inconsistent with another          */
#define DIQT_E_ALIGN       -2   /* a pointer is NULL or not aligned as the kernel requires  */
#define DIQT_E_UNSUPPORTED -3   /* valid request the library has no kernel for              */
#define DIQT_E_LAUNCH      -4   /* HIP reported an error at launch                          */
#define DIQT_E_WORKSPACE   -5   /* workspace too small (see the *_workspace_bytes query)    */

/* activation selectors (imagen_pytorch3D.py:547 Mish; imagen_video.py:690 SiLU; :1113 GELU; :623 ReLU; :625 Sigmoid) */
#define DIQT_ACT_NONE    0
#define DIQT_ACT_MISH    1
#define DIQT_ACT_SILU    2
#define DIQT_ACT_GELU    3
#define DIQT_ACT_RELU    4
#define DIQT_ACT_SIGMOID 5

int         diqt_version(void);
const char* diqt_last_error(void);

/* Launch census (test / measurement infrastructure, no reference counterpart: the reference dispatches through ATen and can be
 * watched with torch.profiler).  Every kernel launch of the library carries a tag (the names diqt_last_error reports, e.g.
 * "conv3d_fwd_h(persistent)", "conv3d_fwd(v9)", "temporal_attention_h").  diqt_census_enable(1) zeroes the per-tag counters and
 * starts counting, (0) stops; both return the previous state.  diqt_census_count(s) = launches since then whose tag contains s
 * (NULL: all).  diqt_get_last_launch() = tag of the calling thread's most recent launch.  Thread-safe. */
int         diqt_census_enable(int on);
long long   diqt_census_count(const char* substr);
const char* diqt_get_last_launch(void);

/* ------------------------------------------------------------------------------------------------
 * Convolution (stride 1, zero padding) as an im2col-free implicit GEMM on v_mfma_f32_32x32x2_f32.
 * Replaces nn.Conv3d / nn.Linear / nn.Conv1d at: Block.project imagen_pytorch3D.py:551-553,566;
 * init_conv :1289-1291,1589; 1x1 convs :467,495,597,1388,1477; Linear :588,622-624,1310,1315;
 * pseudo-3D convs imagen_video.py:352-406,529-543.
 * Output extent per axis: O = I + 2*pad + epad - k + 1, where pad is the low-side (and, with epad = 0, the
 * high-side) zero padding and `epad` is EXTRA high-side padding (may be negative): the causal temporal
 * convolutions of imagen_video.py:399-402 / :1351-1352 are pad = k-1, epad = -(k-1).
 * ---------------------------------------------------------------------------------------------- */

/* number of floats in the packed-weight buffer for a (Cout,Cin,kd,kh,kw) filter */
size_t diqt_conv_packed_elems(int Cout, int Cin, int kd, int kh, int kw);

/* mode 0: forward packing of w[Cout][Cin][kd][kh][kw];
 * mode 1: backward-data packing (taps flipped, in/out swapped) — feed to diqt_conv3d_fwd with
 *         (Cin,Cout) swapped and pad' = k-1-pad to obtain dX from dY.                               */
int diqt_conv_pack_weight(const float* w_oidhw, float* packed, int Cout, int Cin,
                          int kd, int kh, int kw, int mode, void* stream);

/* y[B][Do][Ho][Wo][Cout] = conv(x[B][D][H][W][Cin], packed) + bias   (bias may be NULL).
 * If `residual` != NULL it is added in the epilogue (same shape as y).                              */
int diqt_conv3d_fwd(const float* x, const float* packed, const float* bias, const float* residual,
                    float* y, int B, int D, int H, int W, int Cin, int Cout,
                    int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream);

/* Mixed-precision forward conv (v_mfma_f32_32x32x16_f16 / _bf16, fp32 accumulate): the autocast path of the samplers.
 * Replaces ATen's autocast policy for conv3d / linear -- `torch.autocast` around `ElucidatedImagen.sample` and
 * `ImagenTrainer(fp16=True)` -> `Accelerator(mixed_precision='fp16')` (trainer.py:293-311): operands are cast to fp16 (bf16 = 1:
 * bf16) while the halo tile is staged, products accumulate in fp32, and with round_out = 1 the result (+ bias) is rounded once to
 * the operand type before the fp32 store, as an fp16 output tensor would be.  x, y, bias, residual stay fp32 NDHWC.
 * `packed_h`: 16-bit [ci/32][tap][co padded to 64][32] from diqt_conv_pack_weight_h (diqt_conv_packed_h_elems elements of the
 * EFFECTIVE out/in channel counts).  With the mode-1 packing the same entry point computes backward-data (dX from dY), which the
 * bf16 training path uses; fp16 gradients would need loss scaling and stay on the fp32 kernel.
 * diqt_conv3d_fwd_h_supported() == 0 (Cin % 4 != 0, a tensor >= 1 GiB, halo tile beyond the LDS): stay on diqt_conv3d_fwd.   */
size_t diqt_conv_packed_h_elems(int Cout, int Cin, int kd, int kh, int kw);
int diqt_conv_pack_weight_h(const float* w_oidhw, void* packed_h, int Cout, int Cin, int kd, int kh, int kw, int mode, int bf16,
                            void* stream);
/* `count` such packs in one launch per 64 rows: host_table rows of 8 x int64 {w (device pointer), packed_h (device pointer), Cout, Cin, kd,
 * kh, kw, mode} travel in the kernel arguments (what the captured training micro-step replays to re-derive every packed copy). */
int diqt_conv_pack_weight_h_multi(const long long* host_table, int count, int bf16, void* stream);   /* mode as in diqt_conv_pack_weight: 1 = flipped / swapped packing for backward-data */
int diqt_conv3d_fwd_h_supported(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                                int epd, int eph, int epw);
/* diqt_conv3d_fwd_h with 16-bit tensors at either end: x_half: x holds values of the operand type (fp16, or bf16 when `bf16`) instead of
 * fp32; y_half: y is stored in that type (needs round_out and no residual).  The two convs of a pseudo-3D block (per-frame k x k, then
 * temporal; imagen_video.py:352-406) pass their intermediate tensor this way: the same values the fp32 tensor would hold -- autocast
 * rounds a conv's result to the operand type -- at half the bytes.  Only where ..._io16_supported says 1 (the persistent kernel's
 * shapes, Cin and Cout multiples of 8).                                                                                        */
int diqt_conv3d_fwd_h_io16_supported(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                                     int epd, int eph, int epw, int x_half, int y_half);
/* stats (may be NULL): per-(tile, wave) column sums (sum, sum of squares) of the stored values for the consumer's GroupNorm,
 * [B][nblk][2][Cout] with nblk = diqt_conv3d_fwd_h_stats_blocks(...) (0: none for this shape); feed them to
 * diqt_groupnorm_stats_from_partials -- the statistics pass over the tensor disappears.                                        */
int diqt_conv3d_fwd_h_stats_blocks(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                                   int epd, int eph, int epw, int x_half, int y_half);
int diqt_conv3d_fwd_h_io(const void* x, const void* packed_h, const float* bias, const float* residual, void* y, int B, int D, int H,
                         int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, int bf16,
                         int round_out, int x_half, int y_half, float* stats, void* stream);
/* diqt_conv3d_fwd_h walks (tile, channel-block) units with a persistent kernel of this many workgroups (default 256, one per CU;
 * env DIQT_CONVH_WGS; DIQT_CONVH_PERSIST=0 disables it) when a launch has at least twice as many units and the halo tile fits the
 * register prefetch; n > 0 sets the count, n <= 0 only queries; returns the previous value.  Results do not depend on it.      */
int diqt_set_convh_workgroups(int n);
/* diqt_conv3d_fwd_h_io with x_half = 1 on a 3x3x3 or (1,3,3) filter (Cin % 32 == 0, Cout % 8 == 0) runs conv_f9h_kernel -- halo images by
 * LDS-DMA, one wave per SIMD, weight fragments streamed into registers -- when mode = 1 (default; env DIQT_CONV_F9H) and the tiles fill the
 * chip, for any tile count when mode = 2, never when 0.  mode >= 0 sets it, mode < 0 only queries; returns the previous value.  Results
 * agree with the other 16-bit kernels bit for bit on integer-valued data (different K order otherwise).                          */
int diqt_set_conv_f9h_mode(int mode);
int diqt_conv3d_fwd_h(const float* x, const void* packed_h, const float* bias, const float* residual, float* y, int B, int D, int H,
                      int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, int bf16,
                      int round_out, void* stream);

/* Forward conv that also emits per-tile column sums of its OUTPUT for the consumer's GroupNorm statistics / SE pooling
 * (Block -> Block and Block -> SE3D inside ResnetBlock, imagen_pytorch3D.py:568-632): stats[B][nblk][2][Cout] =
 * (sum, sum of squares) over the valid voxels of each output tile, nblk = diqt_conv3d_fwd_stats_blocks(...) (0: this shape
 * takes a path without statistics and `stats` must be NULL).  Saves one full read of the tensor per consumer.            */
/* Diagnostic: the kernel diqt_conv3d_fwd* runs for this shape -- 0 conv_fwd_kernel, 1 conv_fwd_smallcin_kernel, 2 conv1x1_fwd_kernel,
 * 3 conv_fwd8_kernel, 4 conv_fwd9_kernel (incl. its split-K form, taken when the caller passes the workspace that
 * diqt_conv3d_fwd_workspace_bytes asks for) (-1: bad shape) -- so that per-kernel timings taken around the call carry the names
 * rocprofv3 reports.                                                                                                               */
int diqt_conv3d_fwd_kernel_id(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                              int epd, int eph, int epw);
int diqt_conv3d_fwd_stats_blocks(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                                 int epd, int eph, int epw);
int diqt_conv3d_fwd_ex(const float* x, const float* packed, const float* bias, const float* residual, float* y, float* stats,
                       void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout,
                       int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream);
/* consumers of such partials */
int diqt_groupnorm_stats_from_partials(const float* partials, float* mean, float* rstd, int B, int nblk, int rows_per_batch,
                                       int C, int G, float eps, void* stream);
int diqt_channel_mean_from_partials(const float* partials, float* pooled, int B, int nblk, int rows_per_batch, int C, void* stream);

/* 'same' convolution (odd cubic filter k, padding k/2) over the f^3 sub-volume batch x[f^3][A][A][A][Cin] of ONE merged (fA)^3
 * volume (entry n = b2 + f b3 + f^2 b4, utils_mine.py:25-67), each sub-volume's halo read IN PLACE from its neighbours and zero
 * only outside the merged volume: the reference's boundary_pad (merge_sub_volumes -> F.pad -> overlapping unfold,
 * imagen_pytorch3D.py:37-46) followed by the unpadded Conv3d of Block.forward (:550-566, boundary=True), without the merged and
 * re-split copies.  Same kernels (conv_fwd8 / conv_fwd / small-Cin) and bits as running those copies through them.  stats:
 * [f^3][diqt_conv3d_fwd_neighbours_stats_blocks][2][Cout]; workspace as diqt_conv3d_fwd_ex for (B = f^3, D = H = W = A, pad = k/2).
 * The batch must stay below 1 GiB. */
int diqt_conv3d_fwd_neighbours_stats_blocks(int f, int A, int Cin, int Cout, int k);   /* rows per batch entry of `stats` (0: none) */
int diqt_conv3d_fwd_neighbours(const float* x, const float* packed, const float* bias, const float* residual, float* y, float* stats,
                               void* workspace, size_t workspace_bytes, int f, int A, int Cin, int Cout, int k, void* stream);

/* Same operator with a caller-owned workspace: launches too small to fill the chip (8^3 / 16^3 levels) slice the
 * input-channel chunks over grid.y into output slabs and a second kernel sums them (+ bias + residual) in a fixed
 * order.  diqt_conv3d_fwd_workspace_bytes() returns 0 when the shape is not split.                              */
size_t diqt_conv3d_fwd_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw,
                                       int pd, int ph, int pw, int epd, int eph, int epw);
int diqt_conv3d_fwd_ws(const float* x, const float* packed, const float* bias, const float* residual,
                       float* y, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout,
                       int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream);

/* LDS bytes the MFMA kernel needs for this geometry (> 160 KiB: use diqt_conv3d_direct_*); < 0 on bad shape */
long long diqt_conv3d_lds_bytes(int D, int H, int W, int kd, int kh, int kw, int pd, int ph, int pw,
                                int epd, int eph, int epw);

/* Which kernel diqt_conv3d_bwd_weight dispatches a shape to (profilers / bench.py: per-kernel numbers under rocprofv3's names):
 * 3 conv_wgrad3_kernel (one wave per SIMD, LDS-DMA double-buffered tiles), 2 conv_bwd_weight2_kernel, 1 conv_bwd_weight_kernel,
 * 0 a GEMM / column-sum path (1x1x1 filters, <= 4 input channels, Cout == 1), -1 bad shape. */
int diqt_conv3d_bwd_weight_kernel_id(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                                     int epd, int eph, int epw);

/* dW[Cout][Cin][kd][kh][kw] (OIDHW, overwritten) and dbias[Cout] (may be NULL) from x and dY.
 * Deterministic: split-K partial slabs in `workspace` are reduced in a fixed order.                  */
size_t diqt_conv3d_bwd_weight_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout,
                                              int kd, int kh, int kw, int pd, int ph, int pw,
                                              int epd, int eph, int epw);
int diqt_conv3d_bwd_weight(const float* x, const float* dy, float* dw_oidhw, float* dbias,
                           void* workspace, size_t workspace_bytes,
                           int B, int D, int H, int W, int Cin, int Cout,
                           int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream);

/* The weight gradient with 16-bit MFMA operands (bf16 when `bf16`, else fp16; fp32 accumulate): x and dY are rounded to that type while
 * staged, as the reference's autocast backward does under `ImagenTrainer(precision='bf16')` (trainer.py:293-311).  dbias is summed from the
 * fp32 dY.  3x3x3, (1,3,3), (3,1,1) filters, Cin % 32 == 0; ..._workspace_bytes == 0: shape not taken -- use diqt_conv3d_bwd_weight.  Deterministic.
 * Bit 2 of `bf16` (value 4): dY holds 16-bit values of the operand type too (the gradient a low-precision training step keeps in that type).
 * Bit 1 of `bf16` (value 2): x already HOLDS 16-bit values of the operand type (the GroupNorm-apply output a bf16 training step kept in
 * that type for the conv's forward and for this pass: half the bytes, the same bits the fp32 values round to).                          */
size_t diqt_conv3d_bwd_weight_h_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph,
                                                int pw, int epd, int eph, int epw);
int diqt_conv3d_bwd_weight_h(const float* x, const float* dy, float* dw_oidhw, float* dbias, void* workspace, size_t workspace_bytes,
                             int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd,
                             int eph, int epw, int bf16, void* stream);

/* Stride-1 convolution with one or two output channels (the U-Nets' final convs dim -> channels, imagen_video.py:1560,
 * imagen_pytorch3D.py:1485; GlobalContext.to_k, imagen_video.py:585-601) on the vector ALU: per halo voxel the responses of all taps
 * from ONE pass over x, then a T-point gather.  Exact fp32; OIDHW weights as they are (no packing); any Cin >= 16.
 * Filters 1x1x1, (1,3,3), (3,1,1), 3x3x3 with Cout = 1 and 1x1x1, (1,3,3) with Cout = 2 (..._supported).                    */
int diqt_conv3d_fwd_smallcout_supported(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw,
                                        int epd, int eph, int epw);
int diqt_conv3d_fwd_smallcout(const float* x, const float* w_oidhw, const float* bias, const float* residual, float* y, int B, int D,
                              int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw,
                              void* stream);

/* Direct (non-MFMA) grouped / strided convolution for the FLOP-trivial shapes: depthwise 3^3 and
 * patchify k=s=p convs of the attention blocks (imagen_pytorch3D.py:858-869, 913-924, 960-976),
 * temporal depthwise (3,1,1) PEG (imagen_video.py:1351-1352).  Weights OIDHW with I = Cin/groups.   */
int diqt_conv3d_direct_fwd(const float* x, const float* w, const float* bias, float* y,
                           int B, int D, int H, int W, int Cin, int Cout, int groups,
                           int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                           int epd, int eph, int epw, void* stream);
int diqt_conv3d_direct_bwd_data(const float* dy, const float* w, float* dx,
                                int B, int D, int H, int W, int Cin, int Cout, int groups,
                                int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                                int epd, int eph, int epw, void* stream);
int diqt_conv3d_direct_bwd_weight(const float* x, const float* dy, float* dw, float* dbias,
                                  int B, int D, int H, int W, int Cin, int Cout, int groups,
                                  int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw,
                                  int epd, int eph, int epw, void* stream);
/* Same gradients reduced over the voxel slices in a fixed order through `workspace` (no atomics: bit-reproducible). */
size_t diqt_conv3d_direct_bwd_weight_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout, int groups, int kd, int kh,
                                                     int kw, int sd, int sh, int sw, int pd, int ph, int pw, int epd, int eph, int epw);
int diqt_conv3d_direct_bwd_weight_ws(const float* x, const float* dy, float* dw, float* dbias, void* workspace,
                                     size_t workspace_bytes, int B, int D, int H, int W, int Cin, int Cout, int groups,
                                     int kd, int kh, int kw, int sd, int sh, int sw, int pd, int ph, int pw, int epd, int eph,
                                     int epw, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm + (scale+1)*x+shift + activation  — Block.forward imagen_pytorch3D.py:555-562,
 * imagen_video.py:683-699.  Statistics per (batch, group) over rows_per_batch x (C/G).
 * ---------------------------------------------------------------------------------------------- */
/* workspace: diqt_reduce_workspace_bytes(B, C) bytes, 16-byte aligned (per-(b,c) partial moments) */
size_t diqt_reduce_workspace_bytes(int B, int C);
int diqt_groupnorm_stats(const float* x, float* mean, float* rstd, void* workspace, size_t workspace_bytes,
                         int B, int rows_per_batch, int C, int G, float eps, void* stream);

/* y = act( ((x-mean)*rstd*gamma+beta) * (scale+1) + shift ); scale/shift are both NULL or point at
 * rows of `cond_stride` floats per batch element (the reference chunks one [B][2C] time embedding,
 * imagen_pytorch3D.py:603-605: scale = emb, shift = emb + C, cond_stride = 2C).                       */
int diqt_gn_act_fwd(const float* x, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, const float* scale, const float* shift,
                    int cond_stride, float* y, int B, int rows_per_batch, int C, int G, int act, void* stream);
/* The same with y stored in fp16 (bf16 when `bf16`): under autocast the consumer is a 16-bit-operand conv (diqt_conv3d_fwd_h_io,
 * x_half) that would round these values to that type while staging them.  x_half: x is itself the 16-bit output of such a conv
 * (y_half; its GroupNorm statistics come from that conv's `stats`).  Needs C % 4 == 0 and 16-byte aligned tensors.               */
int diqt_gn_act_fwd_h(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* scale,
                      const float* shift, int cond_stride, void* y_h, int B, int rows, int C, int G, int act, int bf16, int x_half,
                      void* stream);

/* dx, dgamma[C], dbeta[C], dscale[B][C], dshift[B][C] (last two NULL when scale/shift are).
 * `workspace` holds per-(b,c) partial sums: diqt_reduce_workspace_bytes(B,C).                         */
int diqt_gn_act_bwd(const float* x, const float* dy, const float* mean, const float* rstd,
                    const float* gamma, const float* beta, const float* scale, const float* shift,
                    int cond_stride, float* dx, float* dgamma, float* dbeta, float* dscale, float* dshift,
                    void* workspace, size_t workspace_bytes,
                    int B, int rows_per_batch, int C, int G, int act, void* stream);

/* The GroupNorm + activation backward split around the conv it feeds (Block = GN -> act -> conv): diqt_conv3d_fwd_gnbwd is the
 * conv's backward-data pass (x = gradient of the conv output, flipped packed weights) whose epilogue also reduces the GroupNorm
 * backward's per-channel sums over each output tile, reading the GroupNorm input gn_x at the tile's voxels:
 * partials[B][nblk][2][Cout], nblk = diqt_conv3d_fwd_gnbwd_blocks(...) (0: shape not taken; use diqt_conv3d_fwd + diqt_gn_act_bwd).
 * diqt_gn_act_bwd_from_partials then finishes the GroupNorm backward without its own reduction pass over x and dy.
 * Replaces autograd of Block.forward (imagen_pytorch3D.py:535-566, imagen_video.py:671-697).                                       */
int diqt_conv3d_fwd_gnbwd_blocks(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd,
                                 int eph, int epw);
/* Block.forward on the sampling path as one launch (GroupNorm -> (scale + 1) x + shift -> Mish / SiLU -> conv;
 * imagen_pytorch3D.py:546-566, imagen_video.py:680-697): diqt_gn_coef_from_partials (statistics from the producer's column sums) or
 * diqt_gn_coef (from existing statistics) fold the normalisation into y = act(A x + Bc), coef[2][B][C] = (A, Bc); diqt_conv3d_fwd_gn
 * takes the RAW GroupNorm input x and applies the coefficients while it stages its input tiles (arguments otherwise as
 * diqt_conv3d_fwd_ex).  diqt_conv3d_fwd_gn_supported = 0: use diqt_gn_act_fwd + diqt_conv3d_fwd_ex.  act: DIQT_ACT_MISH on 3x3x3
 * filters, DIQT_ACT_SILU on (1,3,3) filters.                                                                                       */
int diqt_gn_coef_from_partials(const float* partials, int nblk, int rows, const float* gamma, const float* beta, const float* scale,
                               const float* shift, int cond_stride, float* mean, float* rstd, float* coef, int B, int C, int G,
                               float eps, void* stream);
int diqt_groupnorm_stats_coef(const float* x, const float* gamma, const float* beta, const float* scale, const float* shift,
                              int cond_stride, float* mean, float* rstd, float* coef, void* workspace, size_t workspace_bytes, int B,
                              int rows, int C, int G, float eps, void* stream);      /* diqt_groupnorm_stats + the coefficients */
int diqt_gn_coef(const float* mean, const float* rstd, const float* gamma, const float* beta, const float* scale, const float* shift,
                 int cond_stride, float* coef, int B, int C, int G, void* stream);
int diqt_conv3d_fwd_gn_supported(int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd, int ph, int pw, int epd,
                                 int eph, int epw, int act);
int diqt_conv3d_fwd_gn(const float* x, const float* packed, const float* bias, const float* residual, float* y, float* stats,
                       void* workspace, size_t workspace_bytes, const float* coef, int act, int B, int D, int H, int W, int Cin, int Cout,
                       int kd, int kh, int kw, int pd, int ph, int pw, int epd, int eph, int epw, void* stream);

/* Process-wide switch of that fusion (default: off unless DIQT_GNBWD_FUSE=1 in the environment at the first query).  The setter
 * returns the previous value; with the switch off diqt_conv3d_fwd_gnbwd_blocks answers 0 for every shape.  Host-side state only
 * (no reference counterpart: both settings compute autograd of Block.forward, imagen_pytorch3D.py:535-566).                        */
int diqt_get_gnbwd_fuse(void);
int diqt_set_gnbwd_fuse(int on);
int diqt_conv3d_fwd_gnbwd(const float* x, const float* packed, float* y, float* partials, const float* gn_x, const float* mean,
                          const float* rstd, const float* gamma, const float* beta, const float* scale, const float* shift,
                          int cond_stride, int G, int act, int B, int D, int H, int W, int Cin, int Cout, int kd, int kh, int kw, int pd,
                          int ph, int pw, int epd, int eph, int epw, void* stream);
/* As diqt_gn_act_bwd (partials == NULL) / diqt_gn_act_bwd_from_partials, plus dx_add (optional, same shape as x): dx = GroupNorm
 * backward + dx_add -- the gradient that reaches x through its second consumer (ResnetBlock: x feeds block1 AND the residual branch,
 * imagen_pytorch3D.py:601-614, imagen_video.py:745-770), so the framework's separate sum over the two branches disappears.          */
int diqt_gn_act_bwd_ex(const float* x, const float* dy, const float* partials, int nblk, const float* dx_add, const float* mean,
                       const float* rstd, const float* gamma, const float* beta, const float* scale, const float* shift,
                       int cond_stride, float* dx, float* dgamma, float* dbeta, float* dscale, float* dshift, void* workspace,
                       size_t workspace_bytes, int B, int rows, int C, int G, int act, void* stream);
/* diqt_gn_act_bwd_ex with x / dy read and / or dx written in a 16-bit type (x_type, dx_type, dy_type: 0 fp32, 1 fp16, 2 bf16): the block1
 * output of a ResnetBlock and the gradient flowing back into it during a low-precision training step (both only meet 16-bit-operand
 * kernels); dy: the output of the backward-data conv in front of this pass, which autocast's backward holds in the operand type. */
int diqt_gn_act_bwd_h(const void* x, const void* dy, const float* partials, int nblk, const float* dx_add, const float* mean,
                      const float* rstd, const float* gamma, const float* beta, const float* scale, const float* shift, int cond_stride,
                      void* dx, float* dgamma, float* dbeta, float* dscale, float* dshift, void* workspace, size_t workspace_bytes, int B,
                      int rows, int C, int G, int act, int x_type, int dx_type, int dy_type, void* stream);
int diqt_gn_act_bwd_from_partials(const float* x, const float* dy, const float* partials, int nblk, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, const float* scale, const float* shift, int cond_stride,
                                  float* dx, float* dgamma, float* dbeta, float* dscale, float* dshift, void* workspace,
                                  size_t workspace_bytes, int B, int rows, int C, int G, int act, void* stream);

/* Per-position LayerNorm over the channel axis, biased variance; gain `g` and optional bias `b` (NULL for the
 * gain-only ChanLayerNorm imagen_pytorch3D.py:361-382 / LayerNorm imagen_video.py:172-200; non-NULL for
 * nn.LayerNorm at imagen_video.py:444,1306).                                                           */
int diqt_chan_layernorm_fwd(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd,
                            int rows, int C, float eps, void* stream);
/* The same with `+ residual` in the same pass: Residual(Attention) / TransformerBlock `attn(x) + x`, whose to_out ends in a LayerNorm
 * (imagen_video.py:217-224, 483-525, 1004-1029); residual may be NULL.                                                          */
int diqt_chan_layernorm_fwd_res(const float* x, const float* g, const float* b, const float* residual, float* y, float* mean,
                                float* rstd, int rows, int C, float eps, void* stream);

/* Depthwise temporal conv of the pseudo-3D U-Net's TemporalPEG -- nn.Conv3d(C, C, (3,1,1), groups=C) after a causal (2,0) or symmetric
 * (1,1) frame pad, wrapped in a Residual (/root/reference/imagen_video.py:1340-1362): x[B][F][P][C] channels-last, w[C][kt], kt = 3,
 * `left` frames of zero padding in front.  y = conv(x) + bias (+ residual).  flip = 1 with left' = kt - 1 - left is the gradient
 * w.r.t. x.  The weight gradient comes tap-major with the bias gradient as its last row: dwb[kt + 1][C].                          */
int diqt_dwconv_temporal_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, int B, int F, int P,
                             int C, int kt, int left, int flip, void* stream);
size_t diqt_dwconv_temporal_bwd_weight_workspace_bytes(int B, int F, int P, int C, int kt);
int diqt_dwconv_temporal_bwd_weight(const float* x, const float* dy, float* dwb, void* workspace, size_t workspace_bytes, int B, int F,
                                    int P, int C, int kt, int left, void* stream);

/* dg[C], db[C] (db may be NULL) are reduced through `workspace` (diqt_reduce_workspace_bytes(1, C)) */
int diqt_chan_layernorm_bwd(const float* x, const float* dy, const float* g, const float* mean,
                            const float* rstd, float* dx, float* dg, float* db, void* workspace,
                            size_t workspace_bytes, int rows, int C, void* stream);
/* ... plus dx_add (optional, same shape as x): dx = LayerNorm backward + dx_add, the residual branch's gradient of `fn(LN(x)) + x`
 * (Attention / ChanFeedForward blocks, imagen_video.py:410-525, 994-1029) summed in the same pass.                                   */
int diqt_chan_layernorm_bwd_ex(const float* x, const float* dy, const float* dx_add, const float* g, const float* mean,
                               const float* rstd, float* dx, float* dg, float* db, void* workspace,
                               size_t workspace_bytes, int rows, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Elementwise activations (n floats) — nn.Mish / SiLU / GELU / ReLU / Sigmoid call sites above.
 * ---------------------------------------------------------------------------------------------- */
int diqt_act_fwd(const float* x, float* y, size_t n, int act, void* stream);
int diqt_act_bwd(const float* x, const float* dy, float* dx, size_t n, int act, void* stream);

/* Skinny nn.Linear for the time-conditioning MLPs (imagen_pytorch3D.py:589-606 to_time_hiddens / to_time_cond /
 * to_time_tokens, :554-557 ResnetBlock.time_mlp; imagen_video.py:1226-1262): y[M][N] = x[M][K] W[N][K]^T + bias,
 * 0 < M <= 64 (batch rows).  W is the nn.Linear [out][in] layout.  bias may be NULL.                    */
int diqt_linear_small_fwd(const float* x, const float* W, const float* bias, float* y, int M, int K, int N, void* stream);
/* dx[M][K], dw[N][K], db[N] (each may be NULL; db needs dw).  dx goes through `workspace`.             */
size_t diqt_linear_small_workspace_bytes(int M, int K, int N);
int diqt_linear_small_bwd(const float* x, const float* W, const float* dy, float* dx, float* dw, float* db,
                          void* workspace, size_t workspace_bytes, int M, int K, int N, void* stream);

/* LearnedSinusoidalPosEmb (imagen_pytorch3D.py:518-533): out[b] = [t, sin(2 pi t w), cos(2 pi t w)] */
int diqt_learned_sinu_fwd(const float* t, const float* w, float* out, int B, int half, void* stream);
/* dw[half] (overwritten) from dout[B][2*half+1] */
int diqt_learned_sinu_bwd(const float* t, const float* w, const float* dout, float* dw, int B, int half, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Squeeze-excite + residual — SE3D imagen_pytorch3D.py:617-632 and `h + res_conv(x)` :610-612.
 * ---------------------------------------------------------------------------------------------- */
/* pooled[b][c] = mean over rows_per_batch of x[b][.][c]  (AdaptiveAvgPool3d(1) :620,630) */
int diqt_channel_mean(const float* x, float* pooled, void* workspace, size_t workspace_bytes,
                      int B, int rows_per_batch, int C, void* stream);
/* The whole pooling of GlobalContext in one pass over x (sampling path): pooled[b][c] = sum_n softmax_n(x[b][n][:] . w)[n] * x[b][n][c]
 * -- to_k (a 1x1 conv to one channel; its bias drops out of the soft-max), softmax over all positions and the weighted sum
 * (imagen_video.py:957-982).  Online soft-max, fixed-order combines.  C in {64, 128, 256}; workspace: diqt_reduce_workspace_bytes(B, C). */
int diqt_softmax_pool_supported(int B, int rows, int C);
int diqt_softmax_pool(const float* x, const float* w, float* pooled, void* workspace, size_t workspace_bytes, int B, int rows, int C,
                      void* stream);
/* out[b][c] = sum_rows w[b][row] * x[b][row][c] — the attention-weighted pooling of GlobalContext
 * (imagen_video.py:975-979: einsum('b i n, b c n -> b c i', softmax(context), x)); workspace: diqt_reduce_workspace_bytes(B, C). */
int diqt_weighted_colsum(const float* x, const float* w, float* out, void* workspace, size_t workspace_bytes,
                         int B, int rows, int C, void* stream);

/* y = h*gate[b][c] + res + alpha*addc[b][c]   (res, addc may be NULL).  The same entry yields the
 * backward dh = dy*gate + (1/rows)*dpooled[b][c] of the gate AND of the mean pool in one pass.        */
int diqt_gate_residual_fwd(const float* h, const float* gate, const float* res, const float* addc, float alpha,
                           float* y, int B, int rows_per_batch, int C, void* stream);
/* The same with res only, plus per-workgroup column sums of y for the consumer's GroupNorm statistics (a ResnetBlock's output feeds
 * the next block's first GroupNorm, imagen_pytorch3D.py:568-614): stats[B][nblk][2 (sum, sum of squares)][C],
 * nblk = diqt_gate_residual_stats_blocks(rows, C) (0: C is not a multiple of 4 dividing 1024 -- use the plain entry point).
 * Feed them to diqt_groupnorm_stats_from_partials; saves one full read of y.                                                  */
int diqt_gate_residual_stats_blocks(int rows, int C);
int diqt_gate_residual_fwd_stats(const float* h, const float* gate, const float* res, float* y, float* stats, int B, int rows, int C,
                                 void* stream);
/* dgate[b][c] = sum_rows dy*h */
int diqt_gate_residual_bwd(const float* h, const float* dy, float* dgate, void* workspace, size_t workspace_bytes,
                           int B, int rows_per_batch, int C, void* stream);
/* The SE gate with the conv output h and the gradient flowing back into it in a 16-bit type (h_type / y_type: 0 fp32, 1 fp16, 2 bf16): a
 * ResnetBlock's block2 output under autocast sampling / low-precision training exists in the operand type only -- autocast rounds a conv
 * result to it anyway (imagen_pytorch3D.py:601-632).  _fwd_h: y = h gate (+ res) (+ alpha addc[b][c]); with res = NULL and addc it is also
 * the backward dh = dy gate + dpooled / rows, written in y_type.                                                                       */
int diqt_gate_residual_fwd_h(const void* h, const float* gate, const float* res, const float* addc, float alpha, void* y, int B, int rows,
                             int C, int h_type, int y_type, void* stream);
int diqt_gate_residual_fwd_stats_h(const void* h, const float* gate, const float* res, float* y, float* stats, int B, int rows, int C,
                                   int h_type, void* stream);
int diqt_gate_residual_bwd_h(const void* h, const float* dy, float* dgate, void* workspace, size_t workspace_bytes, int B, int rows, int C,
                             int h_type, void* stream);
/* SE3D.fc (imagen_pytorch3D.py:621-626,631): hidden = relu(pooled w1^T), gate = sigmoid(hidden w2^T);
 * w1[Cr][C], w2[C][Cr], no biases.  bwd: dpooled, dw1, dw2 from dgate; scratch >= B*(C+Cr) floats.     */
int diqt_se_mlp_fwd(const float* pooled, const float* w1, const float* w2, float* hidden, float* gate,
                    int B, int C, int Cr, void* stream);
/* SE3D squeeze + excitation in ONE launch, one workgroup per batch entry: the channel means are finished from the producer's
 * per-tile column sums `partials` [B][nblk][2][C] (diqt_conv3d_fwd_ex) -- or read from `pooled` when partials == NULL -- then
 * fc1 -> ReLU -> fc2 -> sigmoid.  Writes pooled (when computed here), hidden [B][Cr] and gate [B][C].                         */
int diqt_se_pool_mlp_fwd(const float* partials, int nblk, int rows, float* pooled, const float* w1, const float* w2, float* hidden,
                         float* gate, int B, int C, int Cr, void* stream);
int diqt_se_mlp_bwd(const float* pooled, const float* w1, const float* w2, const float* hidden, const float* gate,
                    const float* dgate, float* dpooled, float* dw1, float* dw2, float* scratch,
                    int B, int C, int Cr, void* stream);
/* x[b][.][c] += v[b][c] broadcast (backward of the mean pool, scaled by caller) and friends */
int diqt_add_channel_broadcast(float* x, const float* v, float alpha, int B, int rows_per_batch, int C,
                               void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data movement — Downsample rearrange imagen_pytorch3D.py:494, PixelShuffle3D :427-439,
 * torch.cat on channels :1576,1653, sub-volume split/merge utils_mine.py:25-67, halo :37-46.
 * ---------------------------------------------------------------------------------------------- */
/* x[B][2D][2H][2W][C] -> y[B][D][H][W][C*8], out channel = c*8 + s1*4 + s2*2 + s3 */
int diqt_space_to_depth2(const float* x, float* y, int B, int D, int H, int W, int C, void* stream);
/* exact inverse (x[B][D][H][W][C*8] -> y[B][2D][2H][2W][C]); also PixelShuffle3D(2) forward */
int diqt_depth_to_space2(const float* x, float* y, int B, int D, int H, int W, int C, void* stream);
/* generic factor-(sd,sh,sw) version, each factor 1 or 2 (2-D per-frame shuffles of imagen_video.py:564-600:
 * Rearrange 'b c f (h p1) (w p2) -> b (c p1 p2) f h w' and nn.PixelShuffle(2)); D,H,W are the coarse extents */
int diqt_space_to_depth_nd(const float* x, float* y, int B, int D, int H, int W, int C, int sd, int sh, int sw, void* stream);
int diqt_depth_to_space_nd(const float* x, float* y, int B, int D, int H, int W, int C, int sd, int sh, int sw, void* stream);
/* x[A][M][N][C] -> y[A][N][M][C]  ('b c f h w' <-> '(b h w) f c' token views, imagen_video.py:1354) */
int diqt_transpose_mid(const float* x, float* y, int A, int M, int N, int C, void* stream);
/* F.interpolate(mode='nearest') on channels-last volumes (resize_video_to imagen_video.py:137-158) */
int diqt_nearest_resize(const float* x, float* y, int B, int D, int H, int W, int C, int Do, int Ho, int Wo, void* stream);
/* its gradient for whole-number up-scaling factors (UpsampleCombiner's resize_video_to of the up-path feature maps,
 * imagen_video.py:1085-1117): dx[b][d][h][w][c] = sum of the Do/D x Ho/H x Wo/W copies in dy                                       */
int diqt_nearest_resize_bwd(const float* dy, float* dx, int B, int D, int H, int W, int C, int Do, int Ho, int Wo, void* stream);
/* F.normalize(dim = -1) over rows of d floats (the l2norm of cosine-sim attention, imagen_video.py:118-119, 484-486, 828-829): rows of
 * x are x_stride floats apart (so the k half of a [.., 2d] k|v row can be normalised in place of a copy), rows of y / dy / dx
 * y_stride / dx_stride apart; inv[rows] = 1 / max(||x||, 1e-12) is kept for the backward dx = (dy - y (y . dy)) inv.               */
int diqt_l2norm_rows_fwd(const float* x, float* y, float* inv, size_t rows, int d, int x_stride, int y_stride, void* stream);
int diqt_l2norm_rows_bwd(const float* y, const float* dy, const float* inv, float* dx, size_t rows, int d, int y_stride, int dx_stride,
                         void* stream);
/* y[rows][Ca+Cb] = cat(a[rows][Ca], b[rows][Cb]) ; and the inverse split */
int diqt_concat_channels(const float* a, int Ca, const float* b, int Cb, float* y, size_t rows, void* stream);
int diqt_split_channels(const float* y, float* a, int Ca, float* b, int Cb, size_t rows, void* stream);
/* y = cat(sa * a, sb * b) over the channel axis and its adjoint (a = sa * y[:, :Ca], b = sb * y[:, Ca:]; either output may be NULL):
 * the scaled skip connections cat(x, skip * 2^-0.5) of the U-Nets (imagen_video.py:1743, imagen_pytorch3D.py:1631) in one pass.
 * float4 lanes when both channel counts are multiples of 4.                                                                  */
int diqt_concat_channels_scaled(const float* a, int Ca, const float* b, int Cb, float sa, float sb, float* y, size_t rows, void* stream);
/* The same pass ALSO writing the column sums (sum, sum of squares) of y per 256-row block, stats[B][nblk][2][Ca + Cb] with nblk =
 * diqt_concat_channels_stats_blocks (0: shape not taken): the concatenated skip connection feeds a ResnetBlock's GroupNorm
 * (imagen_video.py:1743-1747, imagen_pytorch3D.py:1631-1633), whose statistics then come from diqt_groupnorm_stats_from_partials. */
int diqt_concat_channels_stats_blocks(int Ca, int Cb, int rows_per_batch);
int diqt_concat_channels_stats(const float* a, int Ca, const float* b, int Cb, float sa, float sb, float* y, int B, int rows_per_batch,
                               float* stats, void* stream);
int diqt_split_channels_scaled(const float* y, float* a, int Ca, float* b, int Cb, float sa, float sb, size_t rows, void* stream);
/* trilinear x`scale` up-sampling with align_corners=True (nn.Upsample imagen_pytorch3D.py:954) and its
 * adjoint (dx must be zeroed by the caller; accumulated with float atomics)                          */
int diqt_trilinear_up_fwd(const float* x, float* y, int B, int D, int H, int W, int C, int scale, void* stream);
int diqt_trilinear_up_bwd(const float* dy, float* dx, int B, int D, int H, int W, int C, int scale, void* stream);
/* sub-volume batch <-> merged volume, NDHWC; sub-volume n = b2 + f*b3 + f*f*b4 (utils_mine.py:41).
 * halo>0 gathers (A+2*halo)^3 blocks from the zero-padded merged volume (boundary_pad).             */
int diqt_subvolume_gather(const float* vol, float* sub, int f, int A, int C, int halo, void* stream);
int diqt_subvolume_scatter(const float* sub, float* vol, int f, int A, int C, int halo, int accumulate,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * Diffusion elementwise steps.
 * ---------------------------------------------------------------------------------------------- */
/* x_t = alpha[b]*x0 + sigma[b]*noise   — q_sample imagen_pytorch3D.py:311-322 */
int diqt_q_sample(const float* x0, const float* noise, const float* alpha, const float* sigma,
                  float* xt, int B, size_t per_batch, void* stream);
/* DDPM ancestral step — p_mean_variance/p_sample imagen_pytorch3D.py:1996-2055, q_posterior :290-309.
 * x0 = clamp(pred) [clamp_mode 0: min=lo ; 1: [lo,hi]]; x_next = ca[b]*x_t + cb[b]*x0 + cn[b]*noise,
 * where the caller precomputes ca = alpha_next*(1-c)/alpha, cb = alpha_next*c, cn = nonzero*exp(.5*logvar). */
int diqt_ddpm_step(const float* x_t, const float* pred, const float* noise,
                   const float* ca, const float* cb, const float* cn, float lo, float hi, int clamp_mode,
                   float* x_next, float* x0_out, int B, size_t per_batch, void* stream);
/* generic per-batch affine combination used by the EDM Heun sampler (elucidated_imagen.py:476-516):
 * out = clamp( c0[b]*a + c1[b]*b_ + c2[b]*c_ ) ; b_/c_ may be NULL; clamp_mode 0 none, 1 min=lo, 2 [lo,hi] */
int diqt_axpby3(const float* a, const float* b_, const float* c_, const float* c0, const float* c1,
                const float* c2, float lo, float hi, int clamp_mode, float* out,
                int B, size_t per_batch, void* stream);
/* loss = mean_b mean_i (clamp_min(pred,lo) - target)^2, per-batch weights w[b] (NULL = 1)
 * — p_losses imagen_pytorch3D.py:2361-2364; the clamped prediction the reference returns is written to
 * pred_clamped (may alias pred for the reference's in-place form, or be NULL).
 * loss_out: single float (atomic-free two-stage reduce through `partials`, >= 1024 floats).          */
/* diqt_mse_clamp_fwd / _bwd with the per-element loss selected by `kind`: 0 squared error (F.mse_loss), 1 absolute error (F.l1_loss),
 * 2 Huber with beta 1 (F.smooth_l1_loss) -- Imagen(loss_type = 'l2' | 'l1' | 'huber'), imagen_pytorch3D.py:1785-1790, 2370.        */
int diqt_loss_clamp_fwd(const float* pred, float* pred_clamped, const float* target, const float* w, float lo, int do_clamp, int kind,
                        float* partials, float* loss_out, int B, size_t per, void* stream);
int diqt_loss_clamp_bwd(const float* pred, const float* target, const float* w, float lo, int do_clamp, int kind, float gscale,
                        float* dpred, int B, size_t per, void* stream);
int diqt_mse_clamp_fwd(const float* pred, float* pred_clamped, const float* target, const float* w, float lo,
                       int do_clamp, float* partials, float* loss_out, int B, size_t per_batch, void* stream);
/* dpred = gscale * 2*(pred-target)*w[b]/(B*per_batch), zero where pred was clamped (pred <= lo)      */
int diqt_mse_clamp_bwd(const float* pred_clamped, const float* target, const float* w, float lo,
                       int do_clamp, float gscale, float* dpred, int B, size_t per_batch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimiser — Adam (trainer.py:352-359, step at :1056) fused with zero_grad, and the EMA lerp
 * (ema_pytorch, trainer.py:1059-1061), over one flat fp32 arena.
 * ---------------------------------------------------------------------------------------------- */
int diqt_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay,
                   float bias_correction1, float bias_correction2, int zero_grad, void* stream);
/* Global gradient-norm clipping (ImagenTrainer(max_grad_norm=...): accelerator.clip_grad_norm_ -> torch.nn.utils.clip_grad_norm_,
 * trainer.py:1054) over the flat gradient arena.  diqt_grad_norm_clip writes out2[0] = L2 norm of grad[0..n) (fixed-order, fp64
 * combine) and out2[1] = min(1, max_norm / (norm + 1e-6)); `workspace`: diqt_grad_norm_workspace_bytes() bytes, 8-byte aligned.
 * diqt_adam_step_scaled is diqt_adam_step with every gradient multiplied by *grad_scale (a device scalar, e.g. out2 + 1) first. */
size_t diqt_grad_norm_workspace_bytes(void);
int diqt_grad_norm_clip(const float* grad, size_t n, float max_norm, void* workspace, float* out2, void* stream);
int diqt_adam_step_scaled(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                          float lr, float beta1, float beta2, float eps, float weight_decay,
                          float bias_correction1, float bias_correction2, int zero_grad, const float* grad_scale, void* stream);
/* Dynamic thresholding of the predicted x0 (imagen_pytorch3D.py:2006-2021; elucidated_imagen.py:340-358):
 * out[b] = torch.quantile(|x[b, :]|, q) with linear interpolation.  The caller passes the fp32 rank split the way
 * torch does: rank = q * (per - 1) in fp32, k_lo = floor(rank), weight = rank - k_lo.                              */
int diqt_abs_quantile(const float* x, float* out, int B, size_t per_batch, unsigned k_lo, float weight, void* stream);
/* out = clamp(x0, -s[b], s[b]) / s[b] */
int diqt_dynamic_threshold(const float* x0, const float* s, float* out, int B, size_t per_batch, void* stream);
/* Inpainting blend (imagen_pytorch3D.py:2121-2123): out = mask != 0 ? y : x  (mask as 0/1 floats, n elements) */
int diqt_mask_blend(const float* x, const float* y, const float* mask, float* out, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-volume inference (test_all.py:182-300 with data.py:138-202): sliding-window patches of a volume resident in HBM.
 * `idx` = int32 [n][3] patch origins; volumes are fp32 [D][H][W].
 * ---------------------------------------------------------------------------------------------- */
/* out[n][P][P][P] = (vol[origin + ijk] - mean) / std (out may be NULL); nonzero[n] = non-zero RAW voxels (may be NULL):
 * supervisedIQT_INF.__getitem__ / normalize (data.py:168-199) incl. its 5 % non-zero rejection count.               */
int diqt_patch_gather(const float* vol, const int* idx, float* out, int* nonzero, int n_patches, int D, int H, int W, int P,
                      float mean, float stdv, void* stream);
/* pred[origin + ijk] = patches[n][ijk] inside the crop margins[n] = {lo_i, hi_i, lo_j, hi_j, lo_k, hi_k}
 * (the overlap//2 crop with edge special cases, test_all.py:235-298).  Patches are written in index order per launch;
 * overlapping writes of one launch must carry equal margins-free regions (the caller launches overlapping blocks one by one). */
int diqt_patch_scatter(const float* patches, const int* idx, const int* margins, float* pred, int n_patches, int D, int H, int W,
                       int P, void* stream);
/* pred[i] = min_val where (vol[i] - mean) / std == min_val (test_all.py:300) */
int diqt_background_reset(float* pred, const float* vol, size_t n, float mean, float stdv, float min_val, void* stream);
/* out[0] = min(x[0..n)); workspace: 1024 floats */
int diqt_min_value(const float* x, size_t n, float* workspace_1024, float* out, void* stream);

/* ---- training data path + validation metrics on the device ----------------------------------------------
 * data.py:88-137 supervisedIQT.__getitem__: crop a P^3 patch pair out of HBM-resident [V][D][H][W] low-res / high-res volume
 * stacks at sel[n] = {volume, i0, j0, k0} and normalise it: mode 0 (v - mean) / std, mode 1 2*((v - min)/(max - min) - 0.5)
 * with the patch's own extrema (data.py:82-86).  Outputs are [n][P][P][P].  The workspace is only read in mode 1. */
size_t diqt_patch_pair_crop_workspace_bytes(int n_patches, int P);
int diqt_patch_pair_crop(const float* lr_vols, const float* hr_vols, const int* sel, float* lr_out, float* hr_out, void* workspace,
                         size_t workspace_bytes, int n_patches, int V, int D, int H, int W, int P, int mode, float mean, float stdv,
                         void* stream);
/* out2 = {min, max} of x[0..n).  workspace: 8 KiB. */
int diqt_minmax(const float* x, size_t n, void* workspace_8k, float* out2, void* stream);
/* metrics.py:19-23 PSNR -> torchmetrics 0.9.0 peak_signal_noise_ratio(data_range): out2 = {mse, 10 log10(range^2 / mse)} of
 * the tensors after the optional min-max normalisation stats4 = {pred min, pred max, target min, target max} (device, or NULL). */
int diqt_psnr(const float* pred, const float* target, size_t n, const float* stats4, float data_range, void* workspace_8k,
              float* out2, void* stream);
/* metrics.py:25-31 SSIM -> torchmetrics 0.9.0 StructuralSimilarityIndexMeasure on N volumes [D][H][W] (N = batch * channels):
 * K-tap separable Gaussian (taps: HOST pointer, K odd <= 11), constants (k1 range)^2 / (k2 range)^2, mean of the SSIM map over
 * the windows that survive torchmetrics' reflect-pad + crop (= the windows fully inside the volume).  out[0] = SSIM. */
size_t diqt_ssim3d_workspace_bytes(int N, int D, int H, int W, int K);
int diqt_ssim3d(const float* pred, const float* target, int N, int D, int H, int W, const float* taps, int K, const float* stats4,
                float data_range, float k1, float k2, void* workspace, size_t workspace_bytes, float* out, void* stream);

/* Gradient accumulation (accelerate's accumulate()/DDP no_sync, trainer.py:300,1118): the per-parameter gradients of one
 * micro-step are added into the flat gradient arena in ONE launch.  table[t] = {src device pointer, dst element offset,
 * element count} (3 x int64, device memory); every tensor gets `blocks_per_tensor` workgroups.                    */
int diqt_multi_accumulate(float* dst, const long long* table, int count, int blocks_per_tensor, void* stream);
/* The same with the table in HOST memory: it travels in the kernel arguments (120 rows per launch), so there is no table upload and the
 * launches can be captured into a hipGraph as they stand (the trainer's captured micro-step).                                   */
int diqt_multi_accumulate_host(float* dst, const long long* host_table, int count, int blocks_per_tensor, void* stream);
int diqt_ema_lerp(float* ema, const float* param, size_t n, float one_minus_decay, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Attention over flattened volumetric tokens.
 * ---------------------------------------------------------------------------------------------- */
/* softmax over the middle axis of x[outer][n][inner] (inner = 1: row softmax), scaled by `scale`
 * after normalisation (LinearAttention imagen_pytorch3D.py:1003-1006; Attention imagen_video.py:511) */
int diqt_softmax_fwd(const float* x, float* y, size_t outer, int n, int inner, float scale, void* stream);
int diqt_softmax_bwd(const float* y, const float* dy, float* dx, size_t outer, int n, int inner, float scale,
                     void* stream);
/* Multi-query attention probabilities (imagen_video.Attention :490-518): sim[G][n][h][M] holds scores of n queries x h
 * heads against M = n_extra + n_self keys ordered [context..., null, self...]; adds rel[(i-j+n_self-1)][h] (T5-style
 * DynamicPositionBias table, may be NULL) on self keys and null_bias[h] (may be NULL) on the null key (the last extra
 * key), masks self keys j > i when causal, and soft-maxes over M in place layout.  bwd also accumulates drel / dnull
 * (zeroed by the caller; float atomics).                                                                         */
int diqt_attn_softmax_fwd(const float* sim, const float* rel, const float* null_bias, float* p,
                          int G, int n, int h, int n_extra, int n_self, int causal, void* stream);
int diqt_attn_softmax_bwd(const float* p, const float* dp, float* dsim, float* drel, float* dnull_bias,
                          int G, int n, int h, int n_extra, int n_self, int causal, void* stream);
/* Same gradients with the relative-bias / null-bias sums reduced in a fixed order through `workspace`
 * (diqt_attn_softmax_bwd_workspace_bytes; 0 = table too large, the call falls back to the atomic kernel). */
size_t diqt_attn_softmax_bwd_workspace_bytes(int G, int n, int h, int n_extra, int n_self);
int diqt_attn_softmax_bwd_ws(const float* p, const float* dp, float* dsim, float* drel, float* dnull_bias,
                             void* workspace, size_t workspace_bytes, int G, int n, int h, int n_extra, int n_self,
                             int causal, void* stream);
/* Fused multi-query attention forward for the sampling path (imagen_video.py:410-525 Attention.forward): q[G][n][h][d],
 * kv[G][n_extra + n_self][2d] (k | v per key row; extra keys first, learned null key last of them), optional relative
 * position table rel[2n-1][h] on the self keys with null_bias[h] on the null key, optional causal mask ->
 * out[G][n][h*d] = softmax(scale q.k + bias) v.  The [G, n*h, keys] score tensor is never materialised.  d = 32 or 64. */
int diqt_mqa_attention_fwd(const float* q, const float* kv, const float* rel, const float* null_bias, float* out, int G,
                           int n, int h, int d, int n_extra, int n_self, int causal, float scale, void* stream);
/* Training path of the same Attention.forward (imagen_video.py:483-520 `sim = einsum(...)`, `sim.softmax`, `einsum(attn, v)` and
 * their autograd backward): the forward additionally writes lse[G][n*h] (row log-sum-exp), and the backward recomputes the
 * probabilities from it instead of reading a stored [G, n*h, keys] tensor:
 *   dq[G][n*h][d], dkv[G][n_extra + n_self][2d] (every row written: the extra rows carry the gradient of the null / context
 *   keys), drel[2 n_self - 1][h] and dnull[h] (required iff rel / null_bias are given).
 * Deterministic (no atomics; bias tables are summed per wave, then in a fixed order).  workspace: ..._bwd_workspace_bytes. */
/* The temporal attentions of the pseudo-3D U-Net -- EinopsToAndFrom('b c f h w', '(b h w) f c', Attention), imagen_video.py:1351-1354,
 * 410-525 -- on the channels-last tensors as they stand: q[B][F][P][h d], kv[B][F][P][2 d] (k | v of frame f at pixel p), out like q.
 * A sequence is (b, p), its tokens the F frames; one extra key / value nullkv[2 d] in front (the learned null row; null_bias[h] is
 * added to its score when rel is given).  Replaces the two mid-axis transposes and the null-row concatenation around
 * diqt_mqa_attention_fwd; same kernel, bit-identical results.                                                                      */
int diqt_mqa_attention_fwd_frames(const float* q, const float* kv, const float* nullkv, const float* rel, const float* null_bias,
                                  float* out, int B, int F, int P, int h, int d, int causal, float scale, void* stream);
/* The whole temporal attention block of the pseudo-3D U-Net as one kernel (sampling path under autocast): per sequence (b, p) of F frames
 *     y = LayerNorm(Attention(LayerNorm(x; norm_g)) W_o; out_g) + x
 * -- Residual(EinopsToAndFrom('b c f h w', '(b h w) f c', Attention(dim, causal, rel_pos_bias))), imagen_video.py:1351-1354, 410-525 --
 * on x[B][F][P][C] fp32 as it stands.  16-bit MFMA operands (fp16, or bf16 when `bf16`), fp32 accumulation, soft-max and LayerNorms;
 * q, k, v, scores and head outputs never leave the CU.  Weights, 16-bit: wq_h[h d][C] = scale * to_q.weight, wkv_h[2 d][C] =
 * to_kv.weight, wo_h[h][C][d] = to_out.weight[c][head d] with the 64 channels of a head in the accumulator order
 * pos(d) = 16 (d >> 4) + 8 ((d >> 2) & 1) + 4 ((d >> 3) & 1) + (d & 3).  null_kv[2][d], rel[2F-1][h] / null_bias[h] (both or neither).
 * round_out: round the to_out product to the operand type before the LayerNorm (what autocast's Linear returns).
 * Shapes: d = 64, h in {4, 8}, F in {32, 64}, C in {64, 128, 256} (diqt_temporal_attention_h_supported).                              */
int diqt_temporal_attention_h_supported(int B, int F, int P, int C, int h, int d);
int diqt_temporal_attention_h(const float* x, const float* norm_g, const void* wq_h, const void* wkv_h, const void* wo_h,
                              const float* out_g, const float* null_kv, const float* rel, const float* null_bias, float* y,
                              int B, int F, int P, int C, int h, int d, int causal, float eps, int bf16, int round_out, void* stream);
int diqt_mqa_attention_fwd_lse(const float* q, const float* kv, const float* rel, const float* null_bias, float* out, float* lse,
                               int G, int n, int h, int d, int n_extra, int n_self, int causal, float scale, void* stream);
size_t diqt_mqa_attention_bwd_workspace_bytes(int G, int n, int h, int d, int n_extra, int n_self, int has_rel);
int diqt_mqa_attention_bwd(const float* q, const float* kv, const float* rel, const float* null_bias, const float* out,
                           const float* dout, const float* lse, float* dq, float* dkv, float* drel, float* dnull,
                           void* workspace, size_t workspace_bytes, int G, int n, int h, int d, int n_extra, int n_self,
                           int causal, float scale, void* stream);

/* Mixed-precision variant for torch.autocast (the reference's `einsum('b h i d, b j d -> b h i j')` and `einsum(attn, v)` run in
 * fp16 / bf16 there, its soft-max in fp32; imagen_video.py:483-520): q k^T and p v on v_mfma_f32_32x32x16_{f16,bf16} with fp32
 * accumulation and fp32 soft-max statistics; q and out stay fp32 in HBM, `kv_h` is the 16-bit copy of kv (diqt_cast_to_h: every
 * workgroup streams all keys, so the copy halves the dominant traffic); round_out = 1 rounds the result once to the operand type.
 * Otherwise the arguments and layouts of diqt_mqa_attention_fwd.                                      */
int diqt_mqa_attention_fwd_h(const float* q, const void* kv_h, const float* rel, const float* null_bias, float* out, int G, int n,
                             int h, int d, int n_extra, int n_self, int causal, float scale, int bf16, int round_out, void* stream);
/* y[i] = (fp16 | bf16) x[i]: the 16-bit copy of the K|V rows that diqt_mqa_attention_fwd_h streams (n even).                   */
int diqt_cast_to_h(const float* x, void* y_h, size_t n, int bf16, void* stream);


/* batched fp32 MFMA GEMM: C[g] = alpha * op(A[g]) * op(B[g]) (+ beta*C[g]); row-major, strides in floats */
int diqt_bgemm(const float* A, const float* Bm, float* C, int batch, int M, int N, int K,
               int transA, int transB, long long strideA, long long strideB, long long strideC,
               int lda, int ldb, int ldc, float alpha, float beta, void* stream);
/* Same GEMM with a caller-owned workspace: problems with few output tiles and a long K (dK / dV of the attentions, 1- and
 * 5-row products) are split along K into slabs that a second kernel sums in a fixed order.
 * diqt_bgemm_workspace_bytes() returns 0 when the shape is not split.                                              */
size_t diqt_bgemm_workspace_bytes(int batch, int M, int N, int K);
int diqt_bgemm_ws(const float* A, const float* Bm, float* C, void* workspace, size_t workspace_bytes, int batch, int M, int N,
                  int K, int transA, int transB, long long strideA, long long strideB, long long strideC, int lda, int ldb,
                  int ldc, float alpha, float beta, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIQT_H */
