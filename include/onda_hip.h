/*
 * onda_hip.h -- C ABI of libonda_hip.so, the MI355X (gfx950) kernels behind the OnDA
 * adaptation hot path.
 *
 * The reference (theo2021/OnDA) has no FFI: every entry point below replaces the ATen /
 * cuDNN work that one of its Python call sites triggers implicitly (file:line cited per
 * function, relative to the reference root).  Conventions (SURVEY 8b "Error / ownership /
 * threading"):
 *   - plain pointers and sizes only; all device buffers, including workspaces, are owned
 *     by the caller (torch's caching allocator) -- nothing here allocates or frees;
 *   - every call takes the hipStream_t to launch on, holds no thread-local state and is
 *     re-entrant (backward runs on torch's autograd thread);
 *   - return value: 0 on success, a negative ONDA_E* code for bad arguments, a positive
 *     hipError_t if the launch failed.  The Python wrapper raises RuntimeError on != 0;
 *   - activations are NHWC ("pixel-major rows": row m = (b*H + h)*W + w holds C channels,
 *     row stride `ld*` floats), fp32 throughout; labels are int64 with 255 = ignore.
 */
#ifndef ONDA_HIP_H
#define ONDA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* onda_stream_t; /* hipStream_t */

#define ONDA_OK 0
#define ONDA_EINVAL (-1)  /* unsupported shape / null pointer */
#define ONDA_EALIGN (-2)  /* pointer or leading dimension not 16-byte aligned */

/* Geometry of one convolution.  Replaces nn.Conv2d as used by
 * framework/model/deeplabv2.py:22-44 (bottleneck 1x1 / dilated 3x3), :129-184 (ASPP),
 * :201-208 (head), :283 (stem, through onda_stem_im2col). */
typedef struct OndaConv {
  int B, Hi, Wi, Cin;   /* input grid; Cin multiple of 32 (stem/head are padded by the packers) */
  int Ho, Wo, Cout;     /* output grid; Cout multiple of 4 */
  int kh, kw, stride, dil, pad;
  int ldx;              /* input row stride (floats), >= Cin, multiple of 4 */
  int ldy;              /* output row stride (floats): > Cout when writing a concat slice */
  int ldr;              /* residual row stride */
  int out_os;           /* output pixel (ho,wo) is stored at (ho*out_os, wo*out_os) of an   */
  int Hf, Wf;           /* [B,Hf,Wf] grid (out_os=1,Hf=Ho,Wf=Wo normally; 2 for the dgrad of */
                        /* the stride-2 1x1 convs, deeplabv2.py:22-24,351-357)              */
  int relu;
  const int32_t* run_if; /* NULL, or a device int: the launch does nothing when *run_if == 0 (a predicate decided on the
                          * device, onda_switch_step; honoured by the pre-split kernels of csrc/conv_l2.hip, EINVAL elsewhere) */
  int64_t stat_split;    /* 0, or the GEMM row at which a second ROW GROUP starts (two micro-batches in one launch, each with
                          * its own BatchNorm batch statistics: prototypes.py:418-450 runs the student on the source-replay and
                          * on the target batch): the schedule then keeps the tile that straddles the boundary out of the
                          * stream-K remainder, so that onda_bn_finalize_l2 can split its statistics row (csrc/norm_l2.hip) */
  int32_t plain_schedule; /* != 0: one tile per workgroup, no stream-K remainder (a launch that shares the GPU with launches of
                           * other streams: its short last round is filled by them, the remainder's partial tiles and fix-up
                           * launch are not worth their traffic) */
  const int32_t* pix_table; /* onda_conv2d_wgrad_l2 only: NULL, or the caller's table of this geometry's input pixels      */
  int64_t pix_stride;       /* (onda_conv2d_wgrad_l2_table) and its int32 entries per tap                                  */
} OndaConv;

/* Number of float partials conv_fwd writes when `stats` != NULL: tiles_m * 2 * Cout, where
 * tiles_m = onda_conv_tiles_m(M). */
int onda_conv_tiles_m(int M);
/* ... per problem: onda_conv2d_fwd runs (M, Cout) problems that fill the chip on 256 x 128 tiles (round 6) and writes one
 * statistics row per tile row of the kernel it chose: size `stats` with onda_conv_tiles_mc(M, Cout) */
int onda_conv_tiles_mc(int M, int Cout);

/* y = epilogue(conv(x, w)).  w is packed [Cout][kh*kw][Cin] (onda_pack_weight_fwd).
 * Epilogue, in order: per-channel sum / sum-of-squares partials of the raw result into
 * `stats` (BatchNorm batch statistics, deeplabv2.py:25,40,45 in train mode); v*scale[n]
 * + shift[n] (folded eval-mode BatchNorm or conv bias); + residual (deeplabv2.py:65);
 * ReLU (:57,60,66).  scale, shift, residual, stats may each be NULL.  The same entry runs
 * data gradients: pass dy as x and weights packed by onda_pack_weight_dgrad.
 * ws: NULL, or onda_conv_ws_floats() floats of scratch; with it, shapes whose tile count
 * leaves the last round of resident workgroups mostly idle run "stream-K" balanced (the tile x
 * K-step space is cut evenly over the resident workgroups, partial tiles are combined in a
 * fixed order by a second launch -- results stay bitwise reproducible). */
int64_t onda_conv_ws_floats(void);
int onda_conv2d_fwd(const float* x, const float* w, float* y, const float* scale, const float* shift,
                    const float* residual, float* stats, float* ws, const OndaConv* c, onda_stream_t s);

/* ---- "f16x2": the same convolution with TWO f16 limbs per operand and a per-tensor power-of-two
 * scale (csrc/conv_h2.hip): a1*b1 + a1*b2 + a2*b1 on v_mfma_f32_16x16x32_f16 at fp32-GEMM accuracy (3e-7 relative
 * L2 against fp64).
 * A tensor's scale travels as `amax`: ONDA_AMAX_FLOATS device floats whose maximum is max|x| (producers spread
 * their atomicMax over ONDA_AMAX_SLOTS slots, one cache line apart); every consumer derives 2^e (max * 2^e in
 * [2^14, 2^15)) from it in-kernel.  amax buffers must be ZERO before their producer runs; producers are
 * onda_absmax below, or -- fused, no extra pass -- onda_bn_apply / onda_bn_bwd / onda_conv2d_fwd_l2 (their
 * `amax` / `yamax` argument, may be NULL).  onda_absmax: x[rows][ld] with C valid channels, C and ld multiples of
 * 4; a flat tensor (rows == 1) may have any length. */
#define ONDA_AMAX_SLOTS 64     /* slots, one 128-byte line apart */
#define ONDA_AMAX_FLOATS 2048  /* floats per amax buffer */
int onda_absmax(const float* x, int64_t rows, int C, int ld, float* amax, onda_stream_t s);
/* OIHW fp32 weights -> dst[rows_pad][Kp / 32][2][32] f16 LIMB ROWS of w * 2^e(amax) (the operand format of the pre-split
 * kernels, below: per block of 32 K indices a row holds the 32 first limbs, then the 32 second limbs), rows = Cout and
 * K = taps*Cin (dgrad=0: row n, k = tap*Cin + c; Kp a multiple of 32) or rows = Cin and K = taps*Cout_pad (dgrad=1: row c,
 * k = tap'*Cout_pad + n, taps flipped -- the data-gradient operand) */
int onda_pack_weight_h2(const float* w_oihw, void* dst, int Cout, int Cin, int taps, int rows_pad, int Kp, int dgrad,
                        int Cout_pad, const float* amax, onda_stream_t s);
/* All conv weights of a model at once: a device table of OIHW tensors -> per tensor max|w| into `amax` (zeroed) and the
 * forward limb rows fwd[Cout][taps*Cin/32][2][32] (+ the data-gradient rows dgrad[Cin][taps*Cout/32][2][32], taps flipped, unless
 * dgrad is NULL); two launches for the whole table instead of two or three per tensor. */
typedef struct {
  const float* w;
  void* fwd;
  void* dgrad;
  float* amax;
  int Cout, Cin, taps;
  int first_block; /* sum of onda_pack_blocks() of the entries before this one */
} OndaPackEntry;
int onda_pack_blocks(int Cout, int Cin, int taps); /* workgroups one entry takes in the two launches */
int onda_pack_weights_h2_multi(const OndaPackEntry* table, int n, int64_t total_blocks, onda_stream_t s);
/* ---- "f16x2" with BOTH operands pre-split (csrc/conv_l2.hip): activations travel as LIMB ROWS
 *   xl[rows][ldx / 32][2][32] f16:  first limbs l1 = f16(x * 2^e), second limbs l2 = f16((x * 2^e - l1) * 2^11),  e from
 *   xamax (as above); element (row r, channel c) has its first limb at f16 index r * 2*ldx + (c/32)*64 + c%32 and its second
 *   limb 32 further: the 128 bytes a K-step of 32 channels reads of a row are ONE cache line (round 5; until then two planes
 *   xl[2][rows][ldx] = two half lines per row and K-step, and the kernels' K loops were bound by line requests per CU:
 *   tools/micro/dma_rate.hip).  ldx is a multiple of 32.  The `*plane` arguments of the entry points below date from the
 *   two-plane format and are ignored.
 * 4 bytes per element like fp32, written by the kernel that produces the tensor or by onda_split_h2 from an fp32
 * tensor whose max|x| is known.  The conv kernel then moves both operands HBM/L2 -> LDS by LDS-DMA only (no VALU,
 * no LDS stores in the K loop), 256 x 128 tiles on a 3-stage ring filled two K-steps ahead.
 * Replaces the same F.conv2d call sites as onda_conv2d_fwd (deeplabv2.py:53-68, :243-257); c->ldx counts f16
 * elements (channels) of a row.  onda_split_h2: ldo (and the Kp of onda_stem_im2col_l2) must be a multiple of 32 --
 * whole 32-channel blocks per row; anything else is ONDA_EINVAL. */
int onda_split_h2(const float* x, int64_t rows, int C, int ldx, void* dst, int ldo, int64_t plane, const float* amax,
                  onda_stream_t s);
/* onda_conv2d_fwd_l2 whose OUTPUT is limb planes as well: eval-mode conv + folded BatchNorm (scale, shift) [+ residual
 * given as limb planes] [+ ReLU, c->relu] -> out[2][M][c->ldy] f16, the operand format of the next convolution -- no fp32
 * tensor and no split pass in between (deeplabv2.py:53-68 in eval mode: static / dynamic model, evaluation).
 * The planes' scale must exist before the first element is stored, so it comes from an a-priori bound
 *   max|y| <= max|x| * kb[0] + kb[1] (+ max|residual|),  kb[0] = max_c |scale_c| * sum_k |w_ck|,  kb[1] = max_c |shift_c|
 * (device floats, computed once per weight version by the caller).  The kernel writes the bound to out_bound (an amax
 * buffer: it defines the scale of `out` for every consumer) and the TRUE max|y| to out_amax (zeroed amax buffer): the
 * next layer's bound starts from the true maximum (`xtrue` of its own call), so bounds do not compound.  `xtrue` = the
 * amax buffer holding the true max|x| of the input (= xamax for planes whose scale comes from their true maximum).
 * The output is dense ([B,Ho,Wo] rows of c->ldy f16); c->ldr is the residual planes' row length. */
typedef struct OndaLimbOut {
  void* out;
  int64_t out_plane;      /* f16 elements between the two output planes */
  float* out_bound;       /* amax buffer, receives the bound (scale of `out`) */
  float* out_amax;        /* amax buffer, zeroed: true max|y| (may be NULL) */
  const float* kb;        /* {kb[0], kb[1]} */
  const float* xtrue;     /* true max|x| of the input */
  const void* res;        /* residual limb planes or NULL */
  int64_t res_plane;
  const float* res_amax;  /* the residual planes' scale-defining amax buffer */
  const float* res_true;  /* true max|residual| (NULL: res_amax) */
} OndaLimbOut;
int onda_conv2d_fwd_l2_limbs(const void* xl, int64_t xplane, const float* xamax, const void* w2, const float* wamax,
                             const float* scale, const float* shift, const OndaLimbOut* lo, float* ws, const OndaConv* c,
                             onda_stream_t s);

/* Diagnostics: a device buffer of 4096 x 8 uint64 into which onda_conv2d_wgrad_l2's workgroups write s_memtime at the
 * start, after their set-up, after the K loop and after the slab store (slot 5: live K-steps).  NULL switches it off. */
void onda_debug_stamps(void* buffer);

/* Stem patches (onda_stem_im2col) written directly as limb planes dst[2][B*Ho*Wo][Kp] f16 (plane = f16 elements between
 * the two planes): the patch matrix holds image values and zeros, so its max|x| is the image's (xamax, from onda_absmax
 * over the image).  Replaces the stem's F.conv2d input side (deeplabv2.py:283) in "f16x2" pre-split mode. */
int onda_stem_im2col_l2(const float* x_nchw, const float* xamax, void* dst, int64_t plane, int B, int H, int W, int Ho, int Wo,
                        int Kp, onda_stream_t s);
int onda_conv_l2_variant(int64_t M, int Cout);  /* base tile shape of an (M, Cout) problem: 0 256x128, 1 128x128, 2 256x64 */
/* device kernel launched for a problem: 0 / 1 / 2 = conv_l2_kernel<4,2> / <2,2> / <4,1>, 3 = conv_l2x_kernel<4,2> (256x128 tiles,
 * at most 32 K-steps per tile: the continuous K-step stream); bench.py names its per-kernel figures after this.  Short K loops
 * (1x1 convolutions) of a 256x128 problem run as 128x128 tiles on two workgroups per CU where that measured faster (few K-steps,
 * or many column tiles: csrc/conv_l2.hip, l2_variant_k) */
int onda_conv_l2_kernel_id(int64_t M, int Cout, int taps, int Cin);
int onda_conv_l2_tiles_m(int64_t M, int Cout, int taps, int Cin);  /* rows of the `stats` partials the conv writes for this problem */
/* the same for a launch with a row-group boundary (OndaConv.stat_split) and / or OndaConv.plain_schedule; *tile_rows (optional)
 * receives the GEMM rows one partial row covers */
int onda_conv_l2_tiles_m_split(int64_t M, int Cout, int taps, int Cin, int64_t stat_split, int plain_schedule, int* tile_rows);
/* Measurement helpers (bench.py `roofline.pipe`): the share of a problem's K-steps the kernels really issue -- whole tiles of
 * onda_conv2d_fwd_l2 skip filter taps that only see padding (dilated ASPP branches, deeplabv2.py:147-167), the weight
 * gradient skips 32-pixel steps whose rows are all padding for its tap; 1.0 = nothing skipped.  Host arithmetic only. */
double onda_conv_l2_live_fraction(const OndaConv* c, int with_stats);
double onda_conv_wgrad_l2_live_fraction(const OndaConv* c, int splitk);
/* stats_rows: 2 = stats[tile][sum, sumsq][Cout] as onda_conv2d_fwd; 4 = also the per-channel min and max of the raw
 * output tile, from which onda_bn_finalize_l2 bounds max|BatchNorm output| before the apply pass writes limb planes */
int onda_conv2d_fwd_l2(const void* xl, int64_t xplane, const float* xamax, const void* w2, const float* wamax, float* y,
                       const float* scale, const float* shift, const float* residual, float* stats, int stats_rows, float* ws,
                       float* yamax, const OndaConv* c, onda_stream_t s);

/* ---- BatchNorm whose OUTPUT is a conv operand, written as limb planes (csrc/norm_l2.hip); same arithmetic and
 * reference call sites as onda_bn_finalize / onda_bn_apply / onda_bn_bwd (deeplabv2.py:53-68).
 * finalize: partials[tiles][4][C] (sum, sumsq, min, max) -> mean, invstd, running statistics, xhat_amax[C] =
 *   max|(x-mean)*invstd| per channel, and an upper bound of max|out| folded into out_amax (zeroed; res_amax = the amax
 *   of the residual operand or NULL).  apply: out[2][M][C] limb planes of [relu]((x-mean)*invstd*gamma+beta [+res]),
 *   the residual read from ITS limb planes; relu_mask (optional, M*C/8 bytes): bit (m*C+c) & 7 of byte (m*C+c) >> 3 =
 *   [out > 0], read by the backward passes instead of out's first limb (1/8 byte per element instead of 2).
 *   bwd: g = dout*[out>0] (relu_mask, or the sign of out's first limb when relu_mask is NULL), dres = g (fp32, optional),
 *   dx[2][M][C] limb planes of gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)), scaled by a bound folded into dx_amax.
 *   ROW GROUPS (split > 0): rows [0, split) and [split, M) of the same tensor are two micro-batches normalised with their
 *   own batch statistics in one pass (the student's source-replay and target batches, prototypes.py:418-450, with the
 *   BN_POLICY "freeze" of adaptation_model.py:29-36: run_group = the group whose statistics move the running buffers, -1 =
 *   none); mean / invstd / xhat_amax are then [2][C].  finalize splits the one partial row that straddles the boundary
 *   with the conv output itself (y, row stride ldy; tile_rows = GEMM rows per partial row, onda_conv_l2_tiles_m_split). */
int onda_bn_finalize_l2(const float* partials, int tiles, int C, int64_t count, float eps, float* mean, float* invstd,
                        float* running_mean, float* running_var, int64_t* nbt, float momentum, const float* gamma,
                        const float* beta, const float* res_amax, int relu, float* xhat_amax, float* out_amax, int64_t split,
                        int tile_rows, int run_group, const float* y, int ldy, onda_stream_t s);
int onda_bn_apply_l2(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     const void* res, int64_t res_plane, const float* res_amax, void* out, int64_t out_plane,
                     const float* out_amax, int64_t M, int C, int relu, uint8_t* relu_mask, int64_t split, onda_stream_t s);
/* finalize + apply in ONE launch (round 6): BatchNorm's statistics are a grid-wide dependency between a pass of a few
 * workgroups and one of thousands; as two launches the small one costs more in launch latency than in work.  Here the first
 * (C / 16) * groups workgroups of the apply launch finalize, publish (release fence + atomic count) and every workgroup waits
 * for the count (workgroups are dispatched in ascending order: the ones waited for never wait themselves; the wait is bounded).
 * Same arguments and results as the two calls (x dense [M][C]); out_amax = ONDA_AMAX_FLOATS ZEROED floats as before, whose
 * floats 1..3 carry the hand-over's counter (zero before the call, not reusable after it).  onda_bn_bwd_l2's fuse_sums != 0
 * does the same for the backward pass's small reduction (two launches instead of three; counter in dx_amax). */
int onda_bn_train_l2(const float* x, const float* partials, int tiles, float eps, float* mean, float* invstd, float* running_mean,
                     float* running_var, int64_t* nbt, float momentum, const float* gamma, const float* beta, const void* res,
                     const float* res_amax, int relu, float* xhat_amax, void* out, float* out_amax, int64_t M, int C,
                     uint8_t* relu_mask, int64_t split, int tile_rows, int run_group, onda_stream_t s);
int64_t onda_bn_bwd_l2_ws(int64_t M, int C);
int onda_bn_bwd_l2(const float* dout, const void* out, int64_t out_plane, const float* x, const float* mean, const float* invstd,
                   const float* gamma, const float* xhat_amax, void* dx, int64_t dx_plane, float* dx_amax, float* dres, float* ws,
                   int64_t M, int C, int relu, const uint8_t* relu_mask, int64_t split, int fuse_sums, onda_stream_t s);

/* onda_conv2d_wgrad slabs with both operands pre-split: [pixel][channel] limb planes in, LDS-DMA + transposed LDS reads
 * (ds_read_b64_tr_b16), tiles of 256 x 128 or 128 x 128 (output x input channels) per tap and pixel range.
 * The 256 x 128 kernel reads the input pixel of every (tap, output pixel) from a table per convolution GEOMETRY (input and
 * output size, kernel, stride, dilation, padding; 4 bytes per tap and output pixel) that the CALLER owns, like every other
 * buffer: onda_conv2d_wgrad_l2_table_stride(c) = int32 entries per tap at c->B images (0: this problem needs no table --
 * 1 x 1 stride-1 convolutions, the 128 x 128 tile), the table holds kh*kw times that; onda_conv2d_wgrad_l2_table(c, table, s)
 * fills it with one launch on `s`; c->pix_table / c->pix_stride hand it to onda_conv2d_wgrad_l2.  A table built for a larger
 * batch of the same geometry serves a smaller one.  Without a table (pix_table NULL) the kernel works the pixels out in its
 * K loop (the loop of rounds 2-4: same results, 15-25 % slower).  No call of this library allocates or synchronises.
 * (Round 6 gave the exact-fp32 weight gradient the same table: 59.9 -> 61.6 ms per pass, and 66 with the entries fetched a
 * K-step ahead -- its divisions were never what it waits for; not kept.) */
int onda_conv_wgrad_l2_variant(int Cout, int Cin);
int64_t onda_conv2d_wgrad_l2_table_stride(const OndaConv* c);
int onda_conv2d_wgrad_l2_table(const OndaConv* c, int32_t* table, onda_stream_t s);
int onda_conv2d_wgrad_l2(const void* xl, int64_t xplane, const float* xamax, const void* dyl, int64_t dyplane, const float* dyamax,
                         float* slabs, int lddy, int splitk, const OndaConv* c, onda_stream_t s);

/* Weight gradient, split over `splitk` pixel ranges: slabs[ks][Cout][kh*kw][Cin] partial
 * sums of dy[m][n] * x[pix(m,tap)][c]  (autograd of F.conv2d w.r.t. weight).  Then
 * onda_wgrad_reduce sums the slabs in a fixed order (deterministic) into the OIHW
 * gradient dw[Cout_real][Cin_real][kh][kw] (flat_k: stem layout where the packed K index is
 * tap*Cin_real + c); accumulate != 0 adds into dw (gradient accumulation over the source and
 * target backward passes of one step without a separate add pass). */
int onda_conv2d_wgrad(const float* x, const float* dy, float* slabs, int lddy, int splitk,
                      const OndaConv* c, onda_stream_t s);
int onda_wgrad_reduce(const float* slabs, float* dw, int splitk, int Cout, int taps, int Cin,
                      int Cout_real, int Cin_real, int flat_k, int accumulate, onda_stream_t s);

/* OIHW -> kernel layouts.  fwd: dst[n][tap*Cin_real + c] rows of length Kp (zero padded),
 * Cout_pad rows.  dgrad: dst[c][taps-1-tap][n] with rows of Cout_pad, for stride-1 convs
 * the data gradient is then a plain convolution of dy with dst. */
int onda_pack_weight_fwd(const float* w_oihw, float* dst, int Cout, int Cin, int taps, int Cout_pad, int Kp,
                         onda_stream_t s);
int onda_pack_weight_dgrad(const float* w_oihw, float* dst, int Cout, int Cin, int taps, int Cout_pad,
                           onda_stream_t s);

/* Stem patches: col[m][ (r*7+s)*3 + c ] (zero padded to Kp) from the NCHW image
 * f32[B,3,H,W] for the 7x7 stride-2 pad-3 stem (deeplabv2.py:283). */
int onda_stem_im2col(const float* x_nchw, float* col, int B, int H, int W, int Ho, int Wo, int Kp, onda_stream_t s);

/* ---- BatchNorm2d (eps 1e-5, affine frozen) ------------------------------------------------
 * train-mode semantics of torch.nn.BatchNorm2d as the reference toggles it
 * (adaptation_model.py:29-36, prototypes.py:104-110): */
/* partials [tiles][2][C] -> mean[C], invstd[C]; if running_mean != NULL also the momentum
 * update with the unbiased variance and num_batches_tracked += 1. */
int onda_bn_finalize(const float* partials, int tiles, int C, int64_t count, float eps, float* mean, float* invstd,
                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                     onda_stream_t s);
/* per-channel sum / sumsq partials of x[M][C] (when the producer could not emit them) */
int onda_bn_stats(const float* x, int64_t M, int C, int ldx, float* partials, int* tiles_out, onda_stream_t s);
/* out = [relu]( (x-mean)*invstd*gamma + beta [+ residual] ) */
int onda_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  const float* residual, float* out, int64_t M, int C, int relu, float* amax, onda_stream_t s);
/* eval-mode fold: scale = gamma/sqrt(var+eps), shift = beta - mean*scale (for conv_fwd's epilogue) */
int onda_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                 float eps, float* scale, float* shift, int C, onda_stream_t s);
/* backward of bn_apply (+ReLU, + residual fan-out): g = dout * (out > 0 if relu);
 * dres = g (if dres != NULL); dx = gamma*invstd*(g - mean_m(g) - xhat*mean_m(g*xhat)).
 * ws: float workspace of onda_bn_bwd_ws(M, C) floats. */
int64_t onda_bn_bwd_ws(int64_t M, int C);
int onda_bn_bwd(const float* dout, const float* out, const float* x, const float* mean, const float* invstd,
                const float* gamma, float* dx, float* dres, float* ws, int64_t M, int C, int relu, float* amax, onda_stream_t s);

/* ---- GroupNorm(32 groups, eps 1e-5, trainable affine) [+ReLU] [* Dropout2d mask] -----------
 * deeplabv2.py:141,163,182 and :203,250 (feat is taken after the dropout). x[B][HW][C]
 * rows of stride ldx; out rows of stride ldo (concat slice).  chmul: f32[B][C] or NULL. */
int64_t onda_gn_ws(int B, int64_t HW, int C);
int onda_gn_fwd(const float* x, int ldx, const float* gamma, const float* beta, const float* chmul, float* out, int ldo,
                float* mean, float* rstd, float* ws, int B, int64_t HW, int C, int groups, float eps, int relu,
                float* amax /* NULL, or the running max|out| of a tensor that feeds a convolution (f16x2 section) */,
                onda_stream_t s);
int onda_gn_bwd(const float* dout, int lddo, const float* out, int ldo, const float* x, int ldx, const float* gamma,
                const float* chmul, const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta,
                float* ws, int B, int64_t HW, int C, int groups, int relu, onda_stream_t s);

/* ---- MaxPool 3x3 s2 p1 ceil_mode (deeplabv2.py:289-291), NHWC; idx = winning window slot -- */
int onda_maxpool_fwd(const float* x, float* y, uint8_t* idx, int B, int Hi, int Wi, int C, int Ho, int Wo,
                     onda_stream_t s);
/* the same with the result written as limb rows yl[B*Ho*Wo][C/32][2][32] f16 (the operand format of the convolutions that
 * read the pooled stem output; scale from xamax = max|x|, which a maximum over windows cannot pass): no fp32 copy, no split pass */
int onda_maxpool_fwd_limbs(const float* x, const float* xamax, void* yl, uint8_t* idx, int B, int Hi, int Wi, int C, int Ho, int Wo,
                           onda_stream_t s);
int onda_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int Hi, int Wi, int C, int Ho, int Wo,
                     onda_stream_t s);

/* ---- SE block (deeplabv2.py:99-114): pooled mean, two linears, sigmoid, channel scale ------ */
int64_t onda_colsum_ws(int B, int64_t HW, int C);
/* out[b][c] = alpha * sum_px x[b][px][c] (* y[b][px][c] if y != NULL) */
int onda_colsum(const float* x, int ldx, const float* y, int ldy, float* out, float alpha, float* ws, int B,
                int64_t HW, int C, onda_stream_t s);
int onda_se_fc_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2,
                   float* hidden, float* gate, int B, int C, int R, onda_stream_t s);
/* ws: B*R floats.  dpooled is returned multiplied by dpooled_scale (1/HW of the mean pool). */
int onda_se_fc_bwd(const float* dgate, const float* pooled, const float* hidden, const float* gate, const float* w1,
                   const float* w2, float* dw1, float* db1, float* dw2, float* db2, float* dpooled, float* ws,
                   float dpooled_scale, int B, int C, int R, onda_stream_t s);
/* out[b][px][c] = x[b][px][c] * gate[b][c] (+ add[b][c] if add != NULL) */
int onda_chan_scale(const float* x, const float* gate, const float* add, float* out, int B, int64_t HW, int C,
                    onda_stream_t s);
/* the SE gate applied and the result written as the NEXT conv's operand: out[2][B*HW][C] f16 limb planes of x * gate scaled by
 * the scale of x_amax (= max|x|: a sigmoid gate cannot raise it) -- no fp32 copy, no max pass, no split pass */
int onda_chan_scale_limbs(const float* x, const float* gate, void* out, int64_t out_plane, const float* x_amax, int B, int64_t HW,
                          int C, onda_stream_t s);

/* ---- bilinear upsample, align_corners=True (adaptation_model.py:94-98) -------------------
 * logits NHWC [B,h,w] rows of stride ldl holding K classes -> NCHW f32[B,K,H,W]. */
int onda_upsample_fwd(const float* logits, int ldl, float* out_nchw, int B, int h, int w, int K, int H, int W,
                      onda_stream_t s);
/* gradient w.r.t. the low-res logits (gather form, deterministic); dlogits rows of stride ldl */
int onda_upsample_bwd(const float* dout_nchw, float* dlogits, int ldl, int B, int h, int w, int K, int H, int W,
                      onda_stream_t s);
/* fused supervised-loss head: loss_calc(interp(logits), label) (segmentation.py:70-80 -> func.py:35-42 -> loss.py:16-45) without
 * the upsampled tensor.  labels u8[B,H,W] (values >= K, i.e. 255, are ignored); result[0] = mean CE over the kept pixels (NaN
 * when none is kept, as the reference), result[1] = their number; ws: onda_upsample_ce_ws floats.  bwd: dlogits[B,h,w] rows of
 * stride ldl (columns >= K zeroed) = gscale[0] (or 1) * w_ce * d result[0] / d logits: two separable gather passes, deterministic. */
int64_t onda_upsample_ce_ws(int B, int H, int W);
int onda_upsample_ce_fwd(const float* logits, int ldl, const uint8_t* labels, float* result, float* ws, int B, int h, int w, int K,
                         int H, int W, onda_stream_t s);
int64_t onda_upsample_ce_bwd_ws(int B, int w, int K, int H);  /* floats of `ws` for the backward pass */
int onda_upsample_ce_bwd(const float* logits, int ldl, const uint8_t* labels, const float* result, const float* gscale, float w_ce,
                         float* dlogits, float* ws, int B, int h, int w, int K, int H, int W, onda_stream_t s);
/* fused evaluation tail: upsample -> (softmax) -> argmax class map u8[B,H,W]
 * (adaptation_model.py:145-153) without materialising the upsampled tensor */
int onda_upsample_argmax(const float* logits, int ldl, uint8_t* cls, int B, int h, int w, int K, int H, int W,
                         onda_stream_t s);

/* the whole evaluation tail of da_model.evaluate (adaptation_model.py:143-158): upsample ->
 * argmax -> confusion matrix hist[K*K] += (row = ground truth u8[B,H,W] in [0,K), column =
 * prediction), replacing the per-image .cpu().numpy() + np.bincount of func.py:77-79.  hist is
 * accumulated (zero it first); cls (optional) also receives the class map. */
int onda_upsample_argmax_hist(const float* logits, int ldl, const uint8_t* labels, int64_t* hist, uint8_t* cls, int B,
                              int h, int w, int K, int H, int W, onda_stream_t s);

/* ---- per-pixel softmax statistics --------------------------------------------------------
 * probs (optional, rows of stride ldp) = softmax(logits row); argmax (optional int32);
 * result[0] = mean over pixels of the max probability (the "prior ..." / "model" monitor
 * values of prototypes.py:289, prototypes_hybrid_switch.py:55,62).  ws: N/256+1 floats. */
int onda_softmax_stats(const float* logits, int ldl, float* probs, int ldp, int32_t* argmax, float* result,
                       float* ws, int64_t N, int K, onda_stream_t s);

/* ---- target losses (loss.py:16-45 hard CE, :88-112 hard RCE, prototypes.py:29-39 MRKLD) ---
 * fwd: result[0..3] = {ce, rce, mrkld, n_valid}; bwd: dlogits = g * d(w_ce*ce + w_rce*rce +
 * w_reg*mrkld)/dlogits, using the normalisers left in `result` (8 floats: ce, rce, mrkld,
 * n_valid, n_mask) by fwd; columns K..ldl-1 of dlogits are zeroed.  ws: 8*(N/256+1). */
int onda_seg_loss_fwd(const float* logits, int ldl, const int64_t* labels, float* result, float* ws, int64_t N, int K,
                      onda_stream_t s);
int onda_seg_loss_bwd(const float* logits, int ldl, const int64_t* labels, const float* result, const float* gscale,
                      float w_ce, float w_rce, float w_reg, float* dlogits, int64_t N, int K, onda_stream_t s);

/* ---- prototypes (framework/domain_adaptation/methods/prototype_handler.py) ----------------
 * sigma (:53-60) from the state; then per pixel (:111-166): D[k] = || (f - p_k) / sigma ||
 * (mahalanobis=1) or || f - p_k ||, minus its minimum; P = softmax(-D/tau); P *= prior;
 * P /= sum P; label = argmax P, 255 if max P < thresh.  One pass emits the hard labels, the
 * soft map and the three monitor sums result[0..2] = mean max softmax(-D/tau), mean max P,
 * mean max prior.  prior may be NULL.  ws: 3*onda_proto_assign_blocks(N) floats.  C == 256.
 * The feature <-> prototype contraction runs on the matrix cores (v_mfma_f32_32x32x2_f32, fp32 operands):
 * D^2 = |f/s|^2 - 2 (f/s).(p/s) + |p/s|^2; pixels whose decision is closer than 1e-3 (two largest P, or max P
 * against thresh) are redone in the direct form above, so labels are those of the direct form. */
int onda_proto_assign_blocks(int64_t N);
int onda_proto_sigma(const float* proto, const float* sqmean, const float* counter, float* sigma, int K, int C,
                     onda_stream_t s);
int onda_proto_assign(const float* feat, int ldf, const float* prior, int ldp, const float* proto, const float* sigma,
                      int mahalanobis, float tau, float thresh, int64_t* labels, float* soft, float* result,
                      float* ws, int64_t N, int C, int K, onda_stream_t s);
/* the distance matrix alone (:111-138 `mahalanobis_distance` / `distance`): dist[N][K] = D[k] - min_k D, direct form */
int onda_proto_distances(const float* feat, int ldf, const float* proto, const float* sigma, int mahalanobis, float* dist,
                         int64_t N, int C, int K, onda_stream_t s);
/* per-class sums of feat and feat^2 under `cls` (:76-86): sums[2][K][C], counts[K].
 * ws: onda_proto_sums_ws(N, C, K) floats. */
int64_t onda_proto_sums_ws(int64_t N, int C, int K);
int onda_proto_class_sums(const float* feat, int ldf, const int32_t* cls, float* sums, float* counts, float* ws,
                          int64_t N, int C, int K, onda_stream_t s);
/* EMA blend (:88-99): classes with counts > 0 move to lam*p + (1-lam)*sum/count */
int onda_proto_ema(float* proto, float* sqmean, const float* sums, const float* counts, float lam, int K, int C,
                   onda_stream_t s);
/* running mean (:62-74) */
int onda_proto_append(float* proto, float* sqmean, float* counter, const float* sums, const float* counts, int K,
                      int C, onda_stream_t s);

/* ---- optimizer / teacher (adaptation_model.py:88-93 SGD with duplicated entries;
 * prototypes.py:407-416 EMA teacher).  Tables are device arrays of n entries. ---------------- */
typedef struct OndaSgdEntry {
  float* p; const float* g; float* buf; int64_t n; float lr; int times; int fresh;
  int first_block; /* flat launch: blocks of the entries before this one, ceil(n / onda_multi_tensor_block()) each */
} OndaSgdEntry;
int onda_multi_tensor_block(void); /* elements per workgroup of the multi-tensor launches */
/* grad_scale multiplies every gradient element on the way in (1 normally; 1 / world when `g` holds rank sums) */
int onda_sgd_multi(const OndaSgdEntry* table, int n, float momentum, float weight_decay, float grad_scale,
                   int64_t total_blocks, onda_stream_t s);
/* k = k*keep + q*blend  (keep=0, blend=1 copies a buffer, prototypes.py:415-416) */
typedef struct OndaEmaEntry { float* k; const float* q; int64_t n; float keep; float blend; int first_block; } OndaEmaEntry;
int onda_ema_multi(const OndaEmaEntry* table, int n, int64_t total_blocks, onda_stream_t s);

/* ---- input pipeline -----------------------------------------------------------
 * Replaces the per-sample CPU work of framework/dataset/segmentation_db.py:56-99 (`__getitem__`,
 * `_load_img` base_dataset.py:89-95, `preprocess` :98-99, `color_mapper` func.py:88-115) AFTER the
 * PNG decode: Pillow's antialiased BICUBIC resize, RGB->BGR, ToTensor + Normalize, NEAREST label
 * resizes + id map.  Tables come from the host (built in double precision as Pillow builds them);
 * results are bit-identical to the reference's path (fixture G9).
 *   bounds int32[n][2] = (first source index, tap count), kk int32[n][ksize] = 22-bit fixed-point taps. */
/* horizontal pass: in u8[H][Win][3] -> out u8[H][Wout][3] */
int onda_resample_h_u8(const unsigned char* in, unsigned char* out, int H, int Win, int Wout, const int* bounds,
                       const int* kk, int ksize, onda_stream_t s);
/* vertical pass + tensor transform: tmp u8[Hin][W][3] -> out f32[3][Hout][W];
 * out[c] = ((v[flip ? 2-c : c] / 255) - mean3[c]) / std3[c] in fp32; mean3/std3 are HOST pointers */
int onda_resample_v_norm(const unsigned char* tmp, float* out, int Hin, int W, int Hout, const int* bounds, const int* kk,
                         int ksize, const float* mean3, const float* std3, int flip, onda_stream_t s);
/* label path: out u8[Hout][Wout] = lut256[in[ytab[y]][xtab[x]]], in u8[Hin][Win] */
int onda_resize_nearest_lut(const unsigned char* in, unsigned char* out, int Win, int Hout, int Wout, const int* xtab,
                            const int* ytab, const unsigned char* lut256, onda_stream_t s);

/* library identity, for the loader's sanity check */
const char* onda_version(void);
/* factor the second limb of every limb plane is stored with (2048 = 2^11; 1 in measurement builds, csrc/common.h) */
float onda_limb2_scale(void);

/* ---- the hybrid switch on the device (csrc/switch.hip) ---------------------------------------------------------------
 * One monitored series + the two-state machine that reads it, advanced by ONE small launch per adaptation step:
 * replaces Monitor.add / avg / exp / dev_avg for the key "prior static" (framework/utils/monitoring.py:7-96) and
 * model_select.evaluate (prototypes_hybrid_switch.py:22-34) on the step's path -- no read-back, no host decision.
 *   state  : onda_switch_state_doubles(limit) doubles, zero-initialised: [0] exp, [1] avg (median), [2] dev_avg, [3] the
 *            confidence the machine looked at, [8..) the ring of the last `limit` samples
 *   istate : 8 int32, zero-initialised except [2] = [3] = the start state (0 static, 1 dynamic): [0] count, [1] head,
 *            [2] current, [3] current_dev, [4] steps
 *   sample : device scalar (float, or double when sample_f64)
 *   taps   : limit-1 doubles (np.hamming(limit-1) for "hamming", ones for "mean"; unused for level_kind 1 = "median")
 *   flag   : device int32, receives `current` -- the predicate of the dynamic model's launches (OndaConv::run_if)
 * Float64 arithmetic in numpy's order of operations: fixture G5 is reproduced bit for bit. */
typedef struct OndaSwitchCfg {
  int limit;         /* AVG_MONITOR_SIZE */
  int level_kind;    /* 0: weighted level sum(taps*w)/taps_total, 1: median */
  int use_exp;       /* EXP_PR_STATIC: the machine reads the exponential average instead of the median */
  int pad_;
  double exp_const, one_minus_exp_const, taps_total;
  double gray_lo, gray_hi, dev_threshold;
} OndaSwitchCfg;
int onda_switch_state_doubles(int limit);
/* the largest monitor window (AVG_MONITOR_SIZE, monitoring.py:16) the device-side switch holds; longer windows keep the host-side switch */
int onda_switch_max_window(void);
int onda_switch_step(double* state, int32_t* istate, const void* sample, int sample_f64, const double* taps,
                     const OndaSwitchCfg* cfg, int32_t* flag, onda_stream_t s);
/* out[i] = *flag ? wb * b[i] : wa * a[i]: the prior of prototypes_hybrid_switch.py:57-75 picked on the device (`b` may hold
 * anything when *flag == 0: the predicated dynamic forward did not run) */
int onda_select_prior(const int32_t* flag, const float* a, float wa, const float* b, float wb, float* out, int64_t n,
                      onda_stream_t s);
/* out[0] = *flag ? v[0] : NaN (a monitor sample that exists on one side of the switch only) */
int onda_gate_scalar(const int32_t* flag, const float* v, float* out, onda_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
