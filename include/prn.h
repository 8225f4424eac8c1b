/* prn.h -- C ABI of libprn_hip.so: the MI355X (gfx950) kernels behind PlaneRecNet's forward/backward hot path.
 *
 * The reference (EryiXie/PlaneRecNet) has no FFI of its own: its native boundary is the set of PyTorch /
 * torchvision operators its modules call.  Each entry point below replaces one such operator family and
 * cites the reference call sites it serves (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - tensors are fp32, NCHW, contiguous, device pointers; the caller owns every buffer (outputs and
 *     workspaces come from the host framework's allocator -- nothing is allocated, freed or synchronised
 *     inside: the library makes no hipMalloc / hipFree / hip*Synchronize call, tests/test_abi_cpu.py greps for it);
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it;
 *   - the return value is 0 on success, non-zero on error (message: prn_last_error()); nothing throws;
 *   - no global mutable state except the thread-local error string: everything that selects a kernel
 *     (matrix pipe, piece format, thresholds, launch sizes) travels with the call in a prn_gemm_opts value,
 *     pre-cut weight images are passed in by the caller who vouches for them; re-entrant; callable from any host
 *     thread (PyTorch's autograd worker threads call the backward entry points) with different options
 *     per thread.  (The environment is read for tuning overrides only -- PRN_CONV_FORCE and friends, listed
 *     in DESIGN.md -- once, never written.)
 */
#ifndef PRN_H
#define PRN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PRN_BN_SPLITS 32   /* upper bound of per-channel partials in the double workspaces below */

int prn_version(void);
const char* prn_last_error(void);

/* ---- convolution as implicit GEMM on v_mfma_f32_32x32x2_f32 -------------------------------------------
 * replaces ATen/cuDNN conv2d fwd / dgrad / wgrad behind every nn.Conv2d + F.conv2d:
 *   models/backbone.py:22,34-42,45,101,156-164 ; models/dcn.py:25-50 ; models/fpn.py:27,31 ;
 *   planerecnet.py:337,345-353,414-460,510-584,592 ; models/functions/losses.py:91
 * Input-side modes fold the reference's padding / resampling modules into the operand gather:
 *   PRN_IN_ZERO     zero padding `pad`                            (nn.Conv2d(padding=p))
 *   PRN_IN_REFLECT  ReflectionPad2d(1) then conv                  (planerecnet.py:516,522,528,534,579,570)
 *   PRN_IN_UP2_REFLECT  Upsample(x2, nearest) -> ReflectionPad2d(1) -> conv   (planerecnet.py:540-566)
 *   PRN_IN_DILATED  input holds every `dil`-th sample of a zero-dilated tensor (dgrad of a strided conv)
 *   PRN_IN_UP2_PHASE  the SAME operator as PRN_IN_UP2_REFLECT in its sub-pixel form (2.25x fewer multiply-adds): output
 *                   pixel (2i+py, 2j+px) only sees the 2x2 source window at (i+py-1, j+px-1) (replicate border), through
 *                   the phase's own [M, C*2*2] matrix -- sums of the 3x3 taps, built by prn_up2_phase_weights.  KH = KW = 2,
 *                   w = [4][M][C][2][2], Ho = 2H, Wo = 2W; the four phases run as one launch.  For wgrad, dy is phase-major
 *                   [4][B][M][H][W] (prn_space_to_depth2) and dw is [4][M][C][2][2] (prn_up2_wgrad_combine maps it to 3x3).
 */
enum { PRN_IN_ZERO = 0, PRN_IN_REFLECT = 1, PRN_IN_UP2_REFLECT = 2, PRN_IN_DILATED = 3, PRN_IN_UP2_PHASE = 4,
       PRN_IN_EMBED1 = 5 /* Winograd calls only, see prn_conv3x3_winograd */ };
enum { PRN_EPI_NONE = 0, PRN_EPI_RELU = 1, PRN_EPI_SIGMOID = 2 };

/* ---- per-call execution options of the GEMM-shaped operators -------------------------------------------------------------
 * A plain value the caller fills (or zeroes) per call / per descriptor; the library keeps nothing between calls.
 *   split_mode      PRN_SPLIT_OFF: every contraction on the fp32 MFMA kernels (v_mfma_f32_32x32x2_f32).  PRN_SPLIT_PLAN: plain GEMMs
 *                   (stride-1 1x1 convolutions, the 36 transform-domain products of the Winograd path, the DCNv2 column gradient) of at
 *                   least split_min_tiles 128 x 128 output tiles and split_min_gflop GFLOP take the split kernel of prn_gemm_pipe below;
 *                   so do, with the fp16 pieces and C % 32 == 0, three convolution shapes the same kernel walks by TAP (its weight images cut
 *                   tap-major per call): 4x4 / stride 2 / zero padding, 1x1 / stride 2, and the four 2x2 phases of PRN_IN_UP2_PHASE.
 *                   PRN_SPLIT_ALWAYS: wherever that kernel applies (tests: small batches then exercise what batch 8 runs).
 *   split_kind      PRN_PIECES_F16: two fp16 pieces per operand element after an exact power-of-two scaling per weight row / activation
 *                   column, split_products = 3 (l*h, h*l, h*h) or 4 (+ l*l) v_mfma_f32_32x32x16_f16 per multiply-add.  PRN_PIECES_BF16:
 *                   three exact bf16 pieces, six v_mfma_f32_32x32x16_bf16, no scaling.  Both accumulate in fp32.
 *   wgrad_split     PRN_SPLIT_OFF / _PLAN / _ALWAYS for the WEIGHT-GRADIENT GEMMs of the same layers (dW = dy * x^T of stride-1 1x1 convolutions,
 *                   grouped or not, and the 36 products of the Winograd weight gradient; csrc/prn_wgrad16.hip): both operands are
 *                   activations and are cut into two fp16 pieces INSIDE the launch, each row scaled by an exact power of two that follows
 *                   the row's running maximum; three (split_products = 4: four) fp16 MFMA products per multiply-add, fp32 accumulate.
 *   wgrad_wgs       weight-gradient launches are planned for this many workgroups instead of a full residency round (0): a weight
 *                   gradient that shares the GPU with the main chain should not fill every CU's registers.  wgrad_target: workgroups a
 *                   many-tile launch splits up to (0 = 2048).
 * An all-zero value means "fp32 MFMA kernels, full-round weight gradients"; prn_gemm_opts_default() fills the shipping defaults
 * (PRN_SPLIT_PLAN for both, fp16 pieces, three products, 300 tiles, 4 GFLOP).  A NULL `opts` argument means the all-zero value. */
enum { PRN_SPLIT_OFF = 0, PRN_SPLIT_PLAN = 1, PRN_SPLIT_ALWAYS = 2 };
enum { PRN_PIECES_BF16 = 0, PRN_PIECES_F16 = 16 };
typedef struct prn_gemm_opts {
  int32_t split_mode;       /* PRN_SPLIT_*  */
  int32_t split_kind;       /* PRN_PIECES_* */
  int32_t split_products;   /* fp16 pieces: 3 (0 means 3) or 4 */
  int32_t split_min_tiles;  /* PRN_SPLIT_PLAN */
  float split_min_gflop;    /* PRN_SPLIT_PLAN */
  int32_t wgrad_wgs;
  int32_t wgrad_target;
  int32_t wgrad_split;      /* PRN_SPLIT_* */
} prn_gemm_opts;
void prn_gemm_opts_default(prn_gemm_opts* o);

typedef struct prn_conv_desc {
  int32_t B, C, H, W;      /* input tensor [B,C,H,W] as stored                                        */
  int32_t M;               /* output channels                                                          */
  int32_t KH, KW;          /* 1x1, 3x3, 7x7; 2x2 (PRN_IN_UP2_PHASE), 4x4 (zero padding; its input gradient)  */
  int32_t stride, pad;     /* pad is ignored (==1) for the reflect modes                               */
  int32_t Ho, Wo;          /* output spatial size                                                      */
  int32_t in_mode;         /* PRN_IN_*                                                                 */
  int32_t dil;             /* PRN_IN_DILATED: dilation factor of the virtual input                     */
  int32_t epilogue;        /* PRN_EPI_* (fwd only)                                                     */
  int32_t ystride;         /* 0/1: dense output.  2 (fwd only): output pixel (oh,ow) is stored at (2*oh, 2*ow) of  */
  int32_t yH, yW;          /*    a caller-zeroed [B,M,yH,yW] tensor (input gradient of a stride-2 1x1 conv); `addend` is then read at the OUTPUT's index, and
                            *    addend == y adds the result into a tensor that already holds another gradient of the same input (no zero fill, no separate sum) */
  int32_t reserved;
  prn_gemm_opts opts;      /* which kernels this descriptor's calls (and its workspace sizes) are planned for  */
} prn_conv_desc;

/* y[b,m,oh,ow] = epi( sum_{c,r,s} w[m,c,r,s] * gather(x)[b,c,oh*stride-pad+r, ow*stride-pad+s] + bias[m] + addend[b,m,oh,ow] )
 * w is [M, C*KH*KW] row-major; bias and addend may be NULL.  Layers whose output cannot fill the 256 CUs are split
 * along K: `ws` is a caller-owned workspace of prn_conv2d_fwd_ws_bytes(d) bytes (0 => may be NULL); the split partials
 * are summed in a fixed order, so results are deterministic. */
int64_t prn_conv2d_fwd_ws_bytes(const prn_conv_desc* d);
/* 0: the descriptor's forward runs on the fp32-MFMA implicit-GEMM kernel; 1: on the direct HBM-bound kernels (3x3 layers with one or two
 * output channels, or one input channel, over large maps: the depth head); 2 / 3: on the split GEMM kernel (prn_gemm_pipe below, per d->opts)
 * without / with a K split (3: a reduce launch follows, as for a split fp32 launch) -- for profilers that attribute launches to a roofline. */
int prn_conv2d_kernel_kind(const prn_conv_desc* d);
/* Where phase 1 of prn_conv2d_fwd_phase / prn_conv2d_fwd_counted leaves the K-split partial sums of the descriptor's forward: returns s > 1 when
 * ws + *offset_bytes holds s dense [B][M][Ho][Wo] tensors, B*M*Ho*Wo elements apart, whose sum in split order (+ bias + addend) is the result phase 2
 * would write; 0: no K split (or only the tail tiles are split).  A consumer that reads the output exactly once -- the BatchNorm behind a 1x1
 * convolution, models/backbone.py:56-66 -- may take the partial sums instead (prn_bn_train_fwd_partials, prn_bn_bwd_partials): one launch and one
 * pass over the tensor less.  Phase 1 is then called with counters == NULL. */
int prn_conv2d_fwd_partials(const prn_conv_desc* d, int64_t* offset_bytes);
/* Which matrix pipe the plain GEMM y[z][b][m][p] = sum_k w[z][m][k] x[z][b][k][p] (a stride-1 1x1 convolution: nz = 1, HW = H*W; a
 * prn_gemm_batched call: B = 1, HW = P, nz = nb) runs on under `opts`.  0: the fp32 MFMA kernel (v_mfma_f32_32x32x2_f32).  s >= 1: the
 * split kernel with s K splits -- fp32 operands cut into 16-bit pieces whose products are exact in fp32 and accumulated in fp32:
 *   PRN_PIECES_F16 : each weight row / activation column is scaled by an exact power of two into fp16's range, then h = fp16(x),
 *                    l = fp16(x - h) (22 of 24 significand bits); products l*h, h*l, h*h (+ l*l with split_products = 4) as
 *                    v_mfma_f32_32x32x16_f16.  Error against fp64 at the fp32 MFMA kernel's level on this network's tensors; on operands
 *                    spanning > 2^17 inside one row / column the smallest elements lose relative precision (tests/test_ops_gpu.py).
 *   PRN_PIECES_BF16: the three 8-bit slices of the 24-bit significand (exact), six of the nine products as v_mfma_f32_32x32x16_bf16;
 *                    the dropped three are <= 2^-23 of the product.  (Pieces cut by truncation: a small sign-symmetric bias the pixel sign
 *                    pattern below cannot cancel; under PRN_SPLIT_ALWAYS the R101 gradient test reports 4x its bound for it -- not the default.)
 * Both forms flip the sign of the activation pieces of the pixels whose index has odd bit parity (Thue-Morse) and flip the result back: the
 * 16-bit pipe's accumulate step truncates toward -infinity (-1e-9 of sum|a||b| on every output, coherent), and the pattern makes that error
 * cancel in every later sum over pixels, also after stride-2 subsampling.
 * This is NOT the reference's fp32 FMA chain bit for bit (neither is any other summation order); bench.py reports it in `dtype`. */
int prn_gemm_pipe(int M, int K, int B, int HW, int nz, const prn_gemm_opts* opts);
/* The weight side of the split kernel ("images": [z][m tile of 128][k slice of 32][piece][k group][row][8 x 16 bit], zero padded; the fp16
 * kind's row exponents behind them) is cut inside every launch -- into the call's workspace -- unless the caller passes `*_images`: images
 * of EXACTLY that call's weight operand, of the call's piece format, cut since the weight last changed (the caller vouches for it; the
 * library keeps no table of them).  prn_split_images_bytes = size of the images of w[nz][M][K] (either kind); prn_split_prepare cuts one
 * dense weight; prn_split_prepare_batched = ONE launch (two for the fp16 pieces) that cuts many DENSE weights (items_dev: device array of
 * {const float* src; void* dst; int32 M, K, nz, pad; int64 first_row_block -- the item's first block of FOUR rows of nz * ceil(M/128)*128
 * rows, total_row_blocks = their sum (fp16 pieces' row exponents); int64 first} -- `first` = the item's first 256-thread block, an item
 * has ceil(nz * ceil(M/128) * ceil(K/32) * 512 / 256) blocks, total_blocks = their sum).  kind: PRN_PIECES_*.
 * (planerecnet_amd.ops.split_images / split_refresh_all: one prepare launch per training step, validity by version counters.) */
int64_t prn_split_images_bytes(int M, int K, int nz);
int prn_split_prepare(const float* w, void* images, int M, int K, int nz, int kind, void* stream);
int prn_split_prepare_batched(const void* items_dev, int n_items, int64_t total_blocks, int64_t total_row_blocks, int kind, void* stream);
int prn_conv2d_fwd(const prn_conv_desc* d, const float* x, const float* w, const float* bias,
                   const float* addend, float* y, void* ws, void* stream);
/* The same with the K-split sum folded into the GEMM launch (no second kernel, one launch less per split layer): `counters` is
 * a caller-owned buffer of PRN_TILE_COUNTERS uint32, ALL ZERO on entry and all zero again once the launch has completed --
 * allocate and clear it once per stream and hand the same buffer to every call on that stream (launches that may run
 * concurrently must not share one).  Partial tiles are written with agent scope; the workgroup arriving last at a tile's
 * counter sums them in split order, so the result is bit-identical to prn_conv2d_fwd's.  counters == NULL, or a launch the
 * fold does not cover (strided / phase outputs, unaligned tensors), behaves exactly like prn_conv2d_fwd_phase.  The fold is
 * opt-in (environment PRN_CONV_FUSED_REDUCE=1, read once): measured faster per layer, neutral on the training step.
 * phase: 0 = whole operator, 1 = GEMM launch only, 2 = the separate sum only (a no-op when the fold was used).  */
#define PRN_TILE_COUNTERS 4096
/* w_images: NULL, or current split-kernel images of w (see prn_split_prepare) -- used when the descriptor's plan is the split kernel. */
int prn_conv2d_fwd_counted(const prn_conv_desc* d, const float* x, const float* w, const void* w_images, const float* bias, const float* addend, float* y,
                           void* ws, unsigned* counters, void* stream, int phase);
/* Ragged batch: `nseg` dense tensors [B, C, H[s], W[s]] stored back to back, convolved with the SAME weights as one
 * GEMM over all their pixels (SOLOv2 applies its instance-head towers to five grid sizes, planerecnet.py:337-360).  Output
 * and addend are packed the same way with M channels.  Stride-1 "same" 1x1 / 3x3 convolutions with zero padding; desc->H,
 * W, Ho, Wo are ignored.  Every segment must hold a multiple of 64 pixels (B*H*W) for the forward / input-gradient call and
 * a multiple of 16 (and H*W % 4 == 0) for the weight gradient; otherwise the call fails and the caller loops over segments. */
#define PRN_MAX_SEGMENTS 6
typedef struct prn_ragged {
  int32_t nseg;            /* 1..PRN_MAX_SEGMENTS */
  int32_t H[PRN_MAX_SEGMENTS], W[PRN_MAX_SEGMENTS];
} prn_ragged;
int prn_conv2d_fwd_ragged(const prn_conv_desc* d, const prn_ragged* rg, const float* x, const float* w, const float* bias,
                          const float* addend, float* y, void* stream);
int64_t prn_conv2d_wgrad_ragged_ws_bytes(const prn_conv_desc* d, const prn_ragged* rg);
int prn_conv2d_wgrad_ragged(const prn_conv_desc* d, const prn_ragged* rg, const float* x, const float* dy, float* dw, void* ws,
                            void* stream);

/* The same call issued in parts, so that a profiler can bracket the GEMM launch and the split-K reduction separately:
 * phase 0 = everything (== prn_conv2d_fwd), 1 = GEMM launch only, 2 = reduction + epilogue only (no-op without a K split). */
int prn_conv2d_fwd_phase(const prn_conv_desc* d, const float* x, const float* w, const float* bias,
                         const float* addend, float* y, void* ws, void* stream, int phase);

/* wt[c][m][KH-1-r][KW-1-s] = w[m][c][r][s]   (operand layout for dgrad-as-forward) */
int prn_weight_flip_transpose(const float* w, float* wt, int M, int C, int KH, int KW, void* stream);

/* The same permutation for a list of weight tensors in ONE launch (all conv weights of a model, once per training step).
 * `items_dev` is a DEVICE array of n_items descriptors.  The launch moves 32 (M) x 32 (C) blocks: `first` is the running sum
 * of ceil(M/32)*ceil(C/32) over the preceding items and total_blocks the sum over all of them. */
typedef struct prn_flip_item {
  const float* src;   /* [M][C][KH][KW] */
  float* dst;         /* [C][M][KH][KW], flipped */
  int M, C, KH, KW;
  int64_t first;
} prn_flip_item;
int prn_weight_flip_transpose_batched(const prn_flip_item* items_dev, int n_items, int64_t total_blocks, void* stream);

/* --- Winograd F(4x4, 3x3) path for 3x3 / stride 1 / pad 1 convolutions with W % 4 == 0 and H >= 5 (csrc/prn_winograd.hip):
 *   y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A   per 4x4 output tile; 4x fewer multiplies than the direct form.
 * Replaces the same reference calls as prn_conv2d_fwd for those shapes (models/backbone.py:49-60 conv2 of Bottleneck,
 * models/fpn.py:60-66 smoothing convs, planerecnet.py mask / depth heads) and their input gradients.
 * P = prn_winograd_tiles(B, H, W) = B * ceil(H/4) * (W/4) rounded up to a multiple of 4.
 *   prn_winograd_input : x [B][C][H][W] -> V [36][C][P]     (in_mode PRN_IN_ZERO or PRN_IN_REFLECT, pad 1)
 *   prn_gemm_batched   : Y_z [M][P] = U_z [M][C] * V_z [C][P] for z < nb, one launch of the MFMA kernel
 *   prn_winograd_output: Y' [36][M][P] -> y [B][M][H][W]  (+ bias[m], + addend, epilogue PRN_EPI_NONE / PRN_EPI_RELU)
 *   prn_conv3x3_winograd: the three in sequence; ws: prn_conv3x3_winograd_ws_bytes (36 * (C + M) * P floats: V first, then Y'; behind
 *                        them whatever prn_gemm_batched needs under `opts`).
 *   in_mode PRN_IN_EMBED1: H, W describe a VIRTUAL zero tensor of which x [B][C][H-2][W-4] is the block starting at (1, 1);
 *   with pad 1 the output [B][M][H][W] then holds, in its columns 0 .. W-3, the FULL correlation of x (output (H-2)+2 by
 *   (W-4)+2) -- the gradient w.r.t. a reflect-padded tensor, folded onto the unpadded one by prn_pad_fold_pitched.
 * U [36][M][C] comes from prn_winograd_weights_batched (u: forward operand, ut [36][C][M]: operand of the input gradient,
 * i.e. the transform of the 180-degree-rotated taps with the channel roles swapped); either pointer may be NULL.
 * `first` / total_blocks count 32 x 32 (M, C) blocks exactly as for prn_flip_item. */
int64_t prn_winograd_tiles(int B, int H, int W);
typedef struct prn_winograd_item {
  const float* src;   /* [M][C][3][3] */
  float* u;           /* [36][M][C] or NULL */
  float* ut;          /* [36][C][M] or NULL */
  int M, C;
  int64_t first;
} prn_winograd_item;
int prn_winograd_weights_batched(const prn_winograd_item* items_dev, int n_items, int64_t total_blocks, void* stream);
int prn_winograd_input(const float* x, float* V, int B, int C, int H, int W, int in_mode, void* stream);
/* ws: prn_gemm_batched_ws_bytes(M, C, nb, opts) bytes (0 => may be NULL; non-zero only where the split kernel cuts U itself);
 * u_images: NULL or current images of U (then no workspace is needed). */
int64_t prn_gemm_batched_ws_bytes(int M, int C, int P, int nb, const prn_gemm_opts* opts);
int prn_gemm_batched(int M, int C, int P, int nb, const float* U, const void* u_images, const float* V, float* Y, void* ws, const prn_gemm_opts* opts,
                     void* stream);
int prn_winograd_output(const float* Y, const float* bias, const float* addend, float* y, int B, int M, int H, int W, int epilogue, void* stream);
/* The output transform fused with the operator that consumes the convolution's result, for maps whose channel fits one workgroup's registers
 * (prn_bn_kernel_kind(B, H * W) == 1: stages 3 / 4 of the backbone) -- one launch and one pass over the tensor less each:
 *   _bn_fwd: x_out = A^T Y' A (the convolution's result, kept for the backward), then training-mode BatchNorm (+ ReLU) of it -> y, statistics -> stats
 *            [2M] and the running buffers: conv2 -> bn2 of a Bottleneck (models/backbone.py:60-62).
 *   _bn_bwd: Y' is the transform-domain result of the input-gradient convolution of conv2, i.e. the gradient of bn1's output; dx = backward of
 *            BatchNorm (+ ReLU, its sign recomputed from x, stats, gamma, beta) w.r.t. bn1's input x, dgamma / dbeta [M] (models/backbone.py:57-58). */
int prn_winograd_output_bn_fwd(const float* Yt, float* x_out, float* stats, const float* gamma, const float* beta, float* y, float* running_mean,
                               float* running_var, int B, int M, int H, int W, float eps, float momentum, int relu, void* stream);
int prn_winograd_output_bn_bwd(const float* Yt, const float* x, const float* stats, const float* gamma, const float* beta, float* dx, float* dgamma,
                               float* dbeta, int B, int M, int H, int W, int relu, void* stream);
/* Weight gradient on the same path: dw = G^T [ sum over tiles (A dy A^T) .* (B^T x B) ] G.
 *   prn_winograd_dy    : dy [B][M][H][W] -> dY' [36][M][P]
 *   prn_gemm_batched_nt: out_z [M][C] = A_z [M][P] * B_z [C][P]^T for z < nb; partial sums [splits][nb][M][C] go to ws
 *                        (splits = prn_gemm_batched_nt_splits, a fixed function of the shape: deterministic reduction)
 *   prn_winograd_dw    : partials -> dw [M][C][3][3] (sums the splits in fixed order, then G^T . G)
 *   prn_conv3x3_winograd_wgrad: all of it; ws of prn_winograd_wgrad_ws_bytes; phase 0 = everything, 1 / 2 / 3 = transforms /
 *                        products / reduction only (profiler brackets). */
int64_t prn_winograd_wgrad_ws_bytes(int B, int C, int H, int W, int M, const prn_gemm_opts* opts);
int prn_winograd_dy(const float* dy, float* Y, int B, int M, int H, int W, void* stream);
int prn_gemm_batched_nt_splits(int M, int C, int P, int nb, const prn_gemm_opts* opts);
int prn_gemm_batched_nt_kind(int M, int C, int P, int nb, const prn_gemm_opts* opts);   /* 0: fp32 MFMA kernel, 2: fp16-piece kernel */
int prn_gemm_batched_nt(int M, int C, int P, int nb, const float* A, const float* Bm, float* ws, const prn_gemm_opts* opts, void* stream);
int prn_winograd_dw(const float* partials, float* dw, int M, int C, int splits, void* stream);
int prn_conv3x3_winograd_wgrad(const float* x, const float* dy, float* dw, void* ws, int B, int C, int H, int W, int M, int in_mode, const prn_gemm_opts* opts,
                               void* stream, int phase);
/* The same with V = B^T x B supplied by the caller: the first 36 * C * P floats of the workspace of the forward call of the
 * same layer, kept alive until the backward pass (same ws size as above; its V part is then unused). */
int prn_conv3x3_winograd_wgrad_v(const float* V, const float* dy, float* dw, void* ws, int B, int C, int H, int W, int M, const prn_gemm_opts* opts, void* stream);
/* Ragged batches (prn_ragged; zero padding; every segment with W % 4 == 0, H >= 5): the tiles of all segments share the 36
 * products, only the transforms look at the segment table.  P = prn_winograd_tiles_ragged(rg, B). */
int64_t prn_winograd_tiles_ragged(const prn_ragged* rg, int B);
int64_t prn_conv3x3_winograd_ragged_ws_bytes(const prn_ragged* rg, int B, int C, int M, const prn_gemm_opts* opts);
int prn_conv3x3_winograd_ragged(const float* x, const float* U, const void* u_images, const float* bias, const float* addend, float* y, void* ws,
                                const prn_ragged* rg, int B, int C, int M, int epilogue, const prn_gemm_opts* opts, void* stream);
int64_t prn_winograd_wgrad_ragged_ws_bytes(const prn_ragged* rg, int B, int C, int M, const prn_gemm_opts* opts);
int prn_conv3x3_winograd_wgrad_ragged(const float* x, const float* dy, float* dw, void* ws, const prn_ragged* rg, int B, int C, int M, const prn_gemm_opts* opts,
                                      void* stream);
int64_t prn_conv3x3_winograd_ws_bytes(int B, int C, int H, int W, int M, const prn_gemm_opts* opts);
int prn_conv3x3_winograd(const float* x, const float* U, const void* u_images, const float* bias, const float* addend, float* y, void* ws, int B, int C, int H,
                         int W, int M, int in_mode, int epilogue, const prn_gemm_opts* opts, void* stream);

/* dw[m, c*KH*KW + r*KW + s] = sum_{b,oh,ow} dy[b,m,oh,ow] * gather(x)[b,c,oh*stride-pad+r,ow*stride-pad+s]
 * `ws` is a caller-owned workspace of prn_conv2d_wgrad_ws_bytes(d) bytes (deterministic split reduction). */
int64_t prn_conv2d_wgrad_ws_bytes(const prn_conv_desc* d);
/* 0: the descriptor's weight gradient runs on the fp32 MFMA kernel, 1: on the direct HBM-bound kernel (one- / two-channel 3x3 layers), 2: on the
 * fp16-piece kernel (csrc/prn_wgrad16.hip; d->opts.wgrad_split) -- G = layers per launch (prn_conv2d_wgrad_grouped), 1 otherwise. */
int prn_conv2d_wgrad_kernel_kind(const prn_conv_desc* d, int G);
int prn_conv2d_wgrad(const prn_conv_desc* d, const float* x, const float* dy, float* dw, void* ws, void* stream);
/* The same for G layers of one shape in ONE launch (blockIdx.z = layer; x[g], dy[g]: HOST arrays of G device pointers;
 * dw: [G, M, C*KH*KW], caller-owned like the workspace of prn_conv2d_wgrad_grouped_ws_bytes(d, G) bytes).  The 1x1 layers of a
 * ResNet stage are 23 such layers (models/backbone.py:170-184): one of them alone needs ~30 pixel splits to fill the machine,
 * a group of 8 needs 4.  Dense 1x1 / 3x3 / 7x7 descriptors (not the 2x2 phases, not the 1- / 2-channel direct path). */
#define PRN_WGRAD_GROUP_MAX 16
int64_t prn_conv2d_wgrad_grouped_ws_bytes(const prn_conv_desc* d, int G);
int prn_conv2d_wgrad_grouped(const prn_conv_desc* d, int G, const float* const* x, const float* const* dy, float* dw, void* ws, void* stream);
int prn_conv2d_wgrad_phase(const prn_conv_desc* d, const float* x, const float* dy, float* dw, void* ws, void* stream, int phase);

/* --- sub-pixel form of Upsample(x2, nearest) -> ReflectionPad2d(1) -> Conv3x3 (planerecnet.py:540-566), see PRN_IN_UP2_PHASE.
 * wp [4][M][C][2][2]: per-phase sums of the 3x3 taps of w [M][C][3][3]. */
int prn_up2_phase_weights(const float* w, float* wp, int M, int C, void* stream);
/* kd [C][M][4][4]: operand of the input gradient, which is conv(dy [B,M,2H,2W], kd, 4x4, stride 2, zero pad 3) -> [B,C,H+2,W+2]
 * (gradient w.r.t. the replicate-padded source), folded onto [B,C,H,W] by prn_replicate_fold. */
int prn_up2_dgrad_weights(const float* w, float* kd, int M, int C, void* stream);
int prn_replicate_fold(const float* dp, float* dx, int B, int C, int H, int W, void* stream);
/* out [4][B][C][H][W] = the four sub-pixel phases of in [B][C][2H][2W] (dy layout for the PRN_IN_UP2_PHASE wgrad). */
int prn_space_to_depth2(const float* in, float* out, int B, int C, int H, int W, void* stream);
/* dw [M][C][3][3] from the per-phase weight gradients dwp [4][M][C][2][2]. */
int prn_up2_wgrad_combine(const float* dwp, float* dw, int M, int C, void* stream);

/* adjoint of the PRN_IN_REFLECT / PRN_IN_UP2_REFLECT gather: folds dP [B,C,Hv+2,Wv+2] (gradient w.r.t. the
 * virtual padded tensor, Hv = H or 2H) back onto dx [B,C,H,W]. */
int prn_pad_fold(const float* dp, float* dx, int B, int C, int H, int W, int up2, void* stream);
/* the same (up2 = 0) for a dp whose rows are `pitch` >= W + 2 floats long: dp [B][C][H+2][pitch] */
int prn_pad_fold_pitched(const float* dp, float* dx, int B, int C, int H, int W, int pitch, void* stream);

/* out[c] = sum_{b,h,w} x[b,c,h,w]    (bias gradients); ws: C*PRN_BN_SPLITS doubles (fixed-order partials) */
int prn_channel_sum(const float* x, float* out, double* ws, int B, int C, int HW, void* stream);

/* ---- modulated deformable convolution (DCNv2) ------------------------------------------------------------
 * replaces torchvision.ops.deform_conv2d (models/dcn.py:59-66) together with the clamp / 2*sigmoid the
 * wrapper applies to the raw offset / modulator conv outputs (models/dcn.py:53-57).
 * om = raw output of the merged offset(18)+modulator(9) 3x3 conv, [B,27,Ho,Wo].
 * cols[b, c*9+k, ho, wo] = 2*sigmoid(om[b,18+k]) * bilinear(x[b,c], ho*s-1+i+clamp(om[b,2k]), wo*s-1+j+clamp(om[b,2k+1]))
 * The contraction with regular_conv.weight is a prn_conv2d_fwd (1x1) over cols.                            */
int prn_dcn_sample(const float* x, const float* om, float* cols, int B, int C, int H, int W, int Ho, int Wo,
                   int stride, float max_offset, void* stream);
/* backward of prn_dcn_sample: dx and d_om from dcols (both fully overwritten).  d_om is reduced from fixed-order channel
 * partials; dx is gathered through a CSR inversion of the sampling pattern built per call (no float atomics).  Both live
 * in `ws` (prn_dcn_sample_bwd_ws_bytes, -1 on invalid sizes). */
int64_t prn_dcn_sample_bwd_ws_bytes(int B, int C, int H, int W, int Ho, int Wo);
int prn_dcn_sample_bwd(const float* x, const float* om, const float* dcols, float* dx, float* d_om, void* ws,
                       int B, int C, int H, int W, int Ho, int Wo, int stride, float max_offset, void* stream);

/* ---- DCNv2 as ONE operator: sampler fused into the MFMA contraction (csrc/prn_dcnv2.hip) -----------------------------
 * replaces torchvision.ops.deform_conv2d(input, offset, weight, bias, stride, padding, mask=mask) -- the reference's single
 * native-operator call site, models/dcn.py:59-66 -- without the [B, 9*Cin, Ho, Wo] column tensor torchvision (and
 * prn_dcn_sample above) materialise.  3x3 kernel, dilation 1, groups = offset groups = 1 (all the reference uses).
 *   raw = 0: torchvision semantics.  offset [B,18,Ho,Wo] (channel 2k = dy, 2k+1 = dx of tap k = 3*i + j), mask [B,9,Ho,Wo] or
 *            NULL (= 1.0), used as given.
 *   raw = 1: `offset` is the raw [B,27,Ho,Wo] output of the merged offset(18)|modulator(9) conv and `mask` is ignored:
 *            offsets are clamped to +-max_offset and the modulation is 2*sigmoid(raw[18+k]) (models/dcn.py:53-57 folded
 *            in); gradients come back for the raw tensor ([B,27,Ho,Wo] in d_offset, d_mask unused).
 * Call order:  prn_dcnv2_table (offset, mask -> gather table, prn_dcnv2_table_bytes; valid until offset / mask change).  The table is opaque: a
 *              per-pixel part (byte offsets of the 2 x 2 footprint + weights, read by the weight gradient and by the forward of layers it keeps) and,
 *              for even C, a per-PATCH part (64 output pixels of one image: bounds of the input window they sample, LDS offsets + weights) from
 *              which the forward stages that window in LDS -- zero-padded where it leaves the image, which IS deform_conv2d's border rule -- and
 *              gathers from there; a patch whose offsets spread its window beyond 32 x 40 pixels gathers from global memory, bit-identically.
 *   forward :  prn_dcnv2_fwd(x, table, w [M,C,3,3], bias) -> y [B,M,Ho,Wo]        (ws: prn_dcnv2_fwd_ws_bytes, K-split partials; w 8-byte aligned)
 *   backward:  prn_dcnv2_bwd_weight(x, table, dy) -> dw [M,C,3,3]                 (re-samples in the operand loader)
 *              prn_dcnv2_bwd_input(dy, wt [9C,M] = w^T, offset, mask) -> dx       (W^T dy into the head of ws, CSR gather;
 *                                                                                  dx may be NULL: column gradient only)
 *              prn_dcnv2_bwd_offset_mask(x, offset, mask) -> d_offset, d_mask     (reads W^T dy from the SAME ws: call it
 *                                                                                  after prn_dcnv2_bwd_input of that layer)
 *   the bias gradient is prn_channel_sum(dy).  `*_phase`: 0 = everything, 1 = GEMM launch only, 2 = split reduction only
 *   (profiler brackets).  dx is order-nondeterministic in the last bits (CSR bins filled through an atomic cursor).      */
typedef struct prn_dcn_desc {
  int32_t B, C, H, W;      /* input [B,C,H,W]                                   */
  int32_t M;               /* output channels                                   */
  int32_t stride, pad;
  int32_t Ho, Wo;          /* (H + 2*pad - 3) / stride + 1                      */
  int32_t raw;             /* see above                                         */
  float max_offset;        /* raw = 1 only                                      */
  int32_t epilogue;        /* forward: PRN_EPI_NONE / PRN_EPI_RELU              */
  prn_gemm_opts opts;      /* column-gradient GEMM (split kernel or not), weight-gradient launch size */
} prn_dcn_desc;
int64_t prn_dcnv2_table_bytes(const prn_dcn_desc* d);
int prn_dcnv2_table(const prn_dcn_desc* d, const float* offset, const float* mask, void* table, void* stream);
int64_t prn_dcnv2_fwd_ws_bytes(const prn_dcn_desc* d);
int prn_dcnv2_fwd(const prn_dcn_desc* d, const float* x, const void* table, const float* w, const float* bias, float* y, void* ws, void* stream);
int prn_dcnv2_fwd_phase(const prn_dcn_desc* d, const float* x, const void* table, const float* w, const float* bias, float* y, void* ws, void* stream,
                        int phase);
int64_t prn_dcnv2_bwd_weight_ws_bytes(const prn_dcn_desc* d);
int prn_dcnv2_bwd_weight(const prn_dcn_desc* d, const float* x, const void* table, const float* dy, float* dw, void* ws, void* stream);
int prn_dcnv2_bwd_weight_phase(const prn_dcn_desc* d, const float* x, const void* table, const float* dy, float* dw, void* ws, void* stream, int phase);
int64_t prn_dcnv2_bwd_ws_bytes(const prn_dcn_desc* d);
/* wt_images: NULL or current split-kernel images of wt [9C, M] (used when d->opts plans the column-gradient GEMM on the split kernel) */
int prn_dcnv2_bwd_input(const prn_dcn_desc* d, const float* dy, const float* wt, const void* wt_images, const float* offset, const float* mask, float* dx,
                        void* ws, void* stream);
/* phase 0 = everything; 1 / 2 = the column-gradient GEMM (launch / K-split sum); 3 = the CSR gather of dx from the column
 * gradient already in ws. */
int prn_dcnv2_bwd_input_phase(const prn_dcn_desc* d, const float* dy, const float* wt, const void* wt_images, const float* offset, const float* mask,
                              float* dx, void* ws, void* stream, int phase);
int prn_dcnv2_bwd_offset_mask(const prn_dcn_desc* d, const float* x, const float* offset, const float* mask, float* d_offset, float* d_mask,
                              void* ws, void* stream);

/* ---- BatchNorm2d (+ residual add, + ReLU) --------------------------------------------------------------------
 * replaces ATen batch_norm fwd/bwd: models/backbone.py:24,44,48,102,166 ; planerecnet.py:518..582           */
/* training statistics: stats[0:C]=mean, stats[C:2C]=invstd; updates running_mean/var (momentum, unbiased var).
 * ws: 2*C*PRN_BN_SPLITS doubles. */
int prn_bn_stats(const float* x, float* stats, float* running_mean, float* running_var, double* ws,
                 int B, int C, int HW, float eps, float momentum, void* stream);
/* y = relu?( (x-mean)*invstd*gamma + beta + residual? ) ; for eval mode pass stats built from running stats */
int prn_bn_apply(const float* x, const float* stats, const float* gamma, const float* beta, const float* residual,
                 float* y, int B, int C, int HW, int relu, void* stream);
/* training forward in two launches (per-channel partial sums, then normalise with the statistics finalised inside the apply
 * kernel): equivalent to prn_bn_stats + prn_bn_apply; stats[2C] receives mean / invstd for the backward. */
/* 1: training-mode forward / backward of a [B, C, HW] layer run as ONE launch each that reads the activation once (small maps: the
 * channel lives in a workgroup's registers), 0: statistics pass + apply pass -- lets a profiler credit the bytes actually moved; `ws` of
 * prn_bn_train_fwd[_into] / prn_bn_bwd[_from] may be NULL where this returns 1. */
int prn_bn_kernel_kind(int B, int HW);
int prn_bn_train_fwd(const float* x, float* stats, const float* gamma, const float* beta, const float* residual, float* y,
                     float* running_mean, float* running_var, double* ws, int B, int C, int HW, float eps, float momentum,
                     int relu, void* stream);
/* backward (training statistics). g = dy * (y>0 if relu). Outputs dx, dgamma, dbeta and, if dres != NULL, dres = g.
 * y may be NULL for a ReLU layer WITHOUT residual: the sign of the forward output is then recomputed from x, the statistics
 * and gamma / beta (same fused multiply-add as the forward), which saves reading y in both passes.  ws: 2*C*PRN_BN_SPLITS doubles. */
int prn_bn_bwd(const float* dy, const float* x, const float* y, const float* stats, const float* gamma, const float* beta,
               float* dx, float* dres, float* dgamma, float* dbeta, double* ws,
               int B, int C, int HW, int relu, int frozen, void* stream);
/* The same two with the layer's OUTPUT (forward) / output GRADIENT (backward) being a channel slice of a wider tensor: element
 * (b, c, p) at y[b * y_batch_stride + c * HW + p] (stride >= C*HW; a multiple of 4 and a 16-byte aligned slice when HW % 4 == 0).  Two layers whose outputs the
 * reference concatenates along the channels (planerecnet.py:DepthDecoder_FPN: torch.cat([lateral, x], 1)) write into /
 * read from the two halves of one buffer: no concatenation kernel, no slice copies of the gradient. */
int prn_bn_train_fwd_into(const float* x, float* stats, const float* gamma, const float* beta, const float* residual, float* y,
                          int64_t y_batch_stride, float* running_mean, float* running_var, double* ws, int B, int C, int HW, float eps,
                          float momentum, int relu, void* stream);
int prn_bn_bwd_from(const float* dy, int64_t dy_batch_stride, const float* x, const float* y, const float* stats, const float* gamma,
                    const float* beta, float* dx, float* dres, float* dgamma, float* dbeta, double* ws,
                    int B, int C, int HW, int relu, int frozen, void* stream);

/* The one-launch forms (prn_bn_kernel_kind(B, HW) == 1 required) with the layer's input (forward) / output gradient (backward) given as the `nparts`
 * K-split partial sums of the GEMM that produces it (prn_conv2d_fwd_partials: parts + i * part_stride, i < nparts, dense [B][C][HW], 16-byte aligned):
 * the kernel sums them in split order while loading -- bit for bit the tensor the producer's own sum launch would have written -- so that launch and one
 * pass over the tensor disappear.  Forward: the summed input is also written to x_out (the backward reads it).  conv1 -> bn1 and conv3's input
 * gradient -> bn2's backward of every stage-3 / stage-4 Bottleneck (models/backbone.py:56-66).  nparts == 1: the plain one-launch forms. */
int prn_bn_train_fwd_partials(const float* parts, int nparts, int64_t part_stride, float* x_out, float* stats, const float* gamma, const float* beta,
                              const float* residual, float* y, float* running_mean, float* running_var, int B, int C, int HW, float eps, float momentum,
                              int relu, void* stream);
int prn_bn_bwd_partials(const float* dparts, int nparts, int64_t part_stride, const float* x, const float* y, const float* stats, const float* gamma,
                        const float* beta, float* dx, float* dres, float* dgamma, float* dbeta, int B, int C, int HW, int relu, int frozen, void* stream);
/* ... and with the layer's OUTPUT leaving a second time in the form its consumer wants: V = B^T . B [36][C][P4] (P4 = prn_winograd_tiles(B, H, W)), the Winograd
 * input transform (prn_winograd_input, PRN_IN_ZERO) of y (forward: for the 3x3 / pad-1 convolution that follows, bn1 -> conv2, models/backbone.py:57-60)
 * or of dx (backward: for the input-gradient convolution of the 3x3 layer in front, bn2 <- conv2) -- the channel's plane goes through LDS, the transform launch
 * and its pass over the tensor disappear.  [B, C, H, W], W % 4 == 0, prn_bn_kernel_kind(B, H * W) == 1; nparts == 1: `parts` / `dparts` is the tensor itself
 * (x_out may then be NULL). */
int prn_bn_train_fwd_winograd(const float* parts, int nparts, int64_t part_stride, float* x_out, float* stats, const float* gamma, const float* beta,
                              const float* residual, float* y, float* running_mean, float* running_var, float* V, int B, int C, int H, int W, float eps,
                              float momentum, int relu, void* stream);
int prn_bn_bwd_winograd(const float* dparts, int nparts, int64_t part_stride, const float* x, const float* y, const float* stats, const float* gamma,
                        const float* beta, float* dx, float* dres, float* dgamma, float* dbeta, float* V, int B, int C, int H, int W, int relu, int frozen,
                        void* stream);

/* ---- GroupNorm(32) + ReLU ---------------------------------------------------------------------------------------
 * replaces ATen group_norm fwd/bwd + ReLU: planerecnet.py:340-342,419-421,436-437,450-451,463-464           */
int prn_gn_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats /*[B*G*2]*/,
                    int B, int C, int HW, int G, float eps, void* stream);
int prn_gn_relu_bwd(const float* dy, const float* x, const float* beta, const float* stats, const float* gamma,
                    float* dx, float* dgamma_part /*[B*C]*/, float* dbeta_part /*[B*C]*/,
                    int B, int C, int HW, int G, void* stream);
/* The same on a ragged batch: nseg dense [B, C, hw[s]] tensors stored back to back (see prn_ragged); statistics are per
 * (segment, image, group): stats [nseg][B][G][2], dgamma_part / dbeta_part [nseg][B][C]. */
int prn_gn_relu_fwd_ragged(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                           int B, int C, int nseg, const int* hw, int G, float eps, void* stream);
int prn_gn_relu_bwd_ragged(const float* dy, const float* x, const float* beta, const float* stats, const float* gamma,
                           float* dx, float* dgamma_part, float* dbeta_part, int B, int C, int nseg, const int* hw, int G,
                           void* stream);

/* out[z][n] = sum over r < R of in[z][r][n] (r ascending), z < nb: the per-image partials dgamma_part / dbeta_part above -> dgamma / dbeta in one launch
 * ([2][B][C] -> [2][C]; ragged: R = nseg * B). */
int prn_sum_rows(const float* in, float* out, int nb, int R, int N, void* stream);

/* ---- measurement aid -----------------------------------------------------------------------------------------------
 * What do the helper launches that a fusion into a producer's epilogue would remove cost INSIDE the overlapped step (tools/ablation_bounds.py)?
 * Bits 0-7 leave a family OUT (its consumers then read stale sums: RESULTS ARE WRONG, and the garbage changes the timing of data-dependent kernels, so
 * only the weight-gradient family is measured this way); bits 8-15 issue the same families TWICE (idempotent launches: results unchanged; the step's
 * increase is the family's cost on the critical path, second read partly from the Infinity Cache).  Never called by the package; returns the old mask.
 * family bit 1: the forward BatchNorm statistics pass of the two-launch (large-map) form; 2: its backward counterpart; 4: the sums of weight-gradient
 * split partials (reduce_splits); 8: prn_channel_sum; 16: the exact x0.5 resize and its adjoint (models/fpn.py:54). */
int prn_debug_skip_launches(int mask);

/* ---- resampling ---------------------------------------------------------------------------------------------------
 * bilinear, align_corners=False (F.interpolate / nn.Upsample): models/fpn.py:54 ; planerecnet.py:115,381,439,453,594 ;
 * losses.py:143,299.  bwd is the exact adjoint in gather form (overwrites dx, deterministic). */
int prn_resize_bilinear_fwd(const float* x, float* y, int BC, int H, int W, int Ho, int Wo, void* stream);
int prn_resize_bilinear_bwd(const float* dy, float* dx, int BC, int H, int W, int Ho, int Wo, void* stream);
/* y = resize(x) + addend ([BC, Ho, Wo], may be NULL): the running sum over the pyramid levels of SOLOv2MaskHead
 * (planerecnet.py:431-441: `feature_add_all_level += self.convs_all_levels[i](...)`, every level but the first ends in an
 * nn.Upsample) without a separate add pass per level. */
int prn_resize_bilinear_add_fwd(const float* x, const float* addend, float* y, int BC, int H, int W, int Ho, int Wo, void* stream);
/* dx = resize^T(dy) + addend: `addend` ([BC, H, W], optional) is another gradient of the resized tensor (the FPN lateral that also
 * feeds its level's 3x3 conv, fpn.py:51-56; the finest FPN map that feeds the mask head AND split_feats, planerecnet.py:98-100):
 * summed here instead of by autograd's separate accumulation pass over the full-size map. */
int prn_resize_bilinear_bwd_add(const float* dy, const float* addend, float* dx, int BC, int H, int W, int Ho, int Wo, void* stream);
/* MaxPool2d(3, stride 2, pad 1): models/backbone.py:104 */
int prn_maxpool3s2_fwd(const float* x, float* y, unsigned char* arg, int BC, int H, int W, int Ho, int Wo, void* stream);
/* arg (may be NULL in the forward): window position r*3+s of every output's maximum, one byte per output; the backward
 * gathers through it (dx fully written, no atomics). */
int prn_maxpool3s2_bwd(const unsigned char* arg, const float* dy, float* dx, int BC, int H, int W, int Ho, int Wo, void* stream);

/* ---- composite blocks (csrc/prn_blocks.hip): fixed operator sequences of the reference's forward as one call ------------------
 * Plain compositions of the launches above on the caller's stream (same arithmetic as issuing them one by one).
 *
 * prn_plane_prior_fwd -- the plane prior of the depth decoder (planerecnet.py:586-594), in its exact reduced form: the x0.25
 *   bilinear resize at the end reads only the two centre samples of every 4-block per axis and everything between is linear
 *   per pixel, so  resize(conv1x1(sigmoid(K . M)))  ==  conv1x1(mean2x2(sigmoid(K . M[centre samples]))).
 *   seg [B,E,h,w] mask features (h, w multiples of 4); kernels [B,NK,E] the NK = 3728 predicted kernels of each image, one row
 *   per grid cell; w1 [F,NK], b1 [F] = depth_decoder.conv1x1.  pooled [B,NK,h/4,w/4] is written for the weight gradient;
 *   out [B,F,h/4,w/4].  All inputs are detached in the reference (planerecnet.py:589,592), so the only gradients are
 *   prn_plane_prior_wgrad (dw1 = d_out x pooled^T) and prn_channel_sum(d_out) for b1.
 * prn_fpn_level_fwd -- one pyramid level (models/fpn.py:51-63): lateral = conv1x1(x) + b_lat + resize(prev -> HxW) (prev = the
 *   FINER level's lateral or NULL: the reference accumulates bottom-up), p_out = [relu](conv3x3(lateral) + b_out).  u_out: the
 *   3x3 weights in the Winograd domain ([36][F][F], prn_winograd_weights_batched) or NULL -- with them, qualifying shapes take
 *   the F(4x4,3x3) path.                                                                                                  */
int64_t prn_plane_prior_ws_bytes(int B, int E, int h, int w, int NK, int F, const prn_gemm_opts* opts);
int prn_plane_prior_fwd(const float* seg, const float* kernels, const float* w1, const float* b1, float* pooled, float* out, void* ws, int B, int E,
                        int h, int w, int NK, int F, const prn_gemm_opts* opts, void* stream);
/* phase 0 = the whole block (== prn_plane_prior_fwd); 1 centre gather, 2 the B per-image dynamic convolutions (one batched MFMA
 * launch), 3 the 2x2 mean, 4 conv1x1 -- so that a profiler can bracket each launch of the block. */
int prn_plane_prior_fwd_phase(const float* seg, const float* kernels, const float* w1, const float* b1, float* pooled, float* out, void* ws, int B, int E,
                        int h, int w, int NK, int F, const prn_gemm_opts* opts, void* stream, int phase);
int64_t prn_plane_prior_wgrad_ws_bytes(int B, int h, int w, int NK, int F, const prn_gemm_opts* opts);
int prn_plane_prior_wgrad(const float* pooled, const float* d_out, float* dw1, void* ws, int B, int h, int w, int NK, int F, const prn_gemm_opts* opts,
                          void* stream);
/* (the split-kernel launches of these two blocks cut their weights per call, into ws) */
int64_t prn_fpn_level_ws_bytes(int B, int C, int H, int W, int F, int relu, int has_prev, int have_u, const prn_gemm_opts* opts);
int prn_fpn_level_fwd(const float* x, const float* w_lat, const float* b_lat, const float* prev, int Hp, int Wp, const float* w_out, const float* u_out,
                      const float* b_out, float* lateral, float* p_out, void* ws, int B, int C, int H, int W, int F, int relu, const prn_gemm_opts* opts,
                      void* stream);

/* ---- one residual block of the backbone per call (csrc/prn_bottleneck.hip) -------------------------------------------------------
 * replaces Bottleneck.forward and its backward (models/backbone.py:53-73; with the deformable conv2 of models/dcn.py:52-67), the unit the
 * reference's backbone calls 33 times per batch (PlaneRecNet_101), in TRAINING mode (batch statistics):
 *   out = relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + identity),  identity = x  or  bn_d(conv_d(x)) (1x1, the block's stride)
 * A block call issues the launch sequence of the operators above from C -- same launches, same arguments, same results as calling them one by
 * one -- with the producer -> BatchNorm hand-overs as internal state (PRN_BLK_HANDOVER: a K-split GEMM's partial sums are summed by the
 * BatchNorm kernel that reads them, a Winograd convolution's output transform is applied by it, and it writes the input transform the next
 * 3x3 convolution wants; small maps only, prn_bn_kernel_kind == 1).  The caller owns these buffers (save / gsave / ws 256-byte aligned):
 *   save   (PRN_BLK_SAVE_BYTES)   forward -> backward: conv / BatchNorm results the backward pass reads, the statistics, the DCNv2 offset map and gather
 *                                 table, conv2's Winograd input transform (PRN_BLK_KEEP_V); the weight gradients read a1, a2, V from it
 *   gsave  (PRN_BLK_GSAVE_BYTES)  backward -> the caller's weight-gradient launches: d1, d2, d3, dd (gradients of conv1 / conv2 / conv3 / conv_d's
 *                                 results), dom (of the raw offset | modulator map)
 *   bn_grads (PRN_BLK_BN_GRAD_FLOATS floats) the BatchNorm parameter gradients dgamma1 | dbeta1 | dgamma2 | dbeta2 | dgamma3 | dbeta3 [| dgamma_d | dbeta_d]
 *   ws     (PRN_BLK_FWD_WS_BYTES / PRN_BLK_BWD_WS_BYTES) scratch, dead when the call returns (one buffer per stream serves every block)
 * The weight gradients are the caller's (prn_conv2d_wgrad[_grouped], prn_conv3x3_winograd_wgrad_v, prn_dcnv2_bwd_weight, prn_channel_sum on the
 * operands above): nothing inside the backward pass reads them, and the caller batches them across blocks on its own stream.
 * prn_bottleneck_plan fills a caller-owned, relocatable plan (prn_bottleneck_plan_bytes() bytes; no pointers inside) from the descriptor: built
 * once per shape.  prn_bottleneck_plan_info reports sizes, offsets and which derived operands the plan reads (PRN_BLK_* indices). */
enum { PRN_BLK_WINOGRAD = 1,      /* conv2 (plain 3x3, stride 1) and its input gradient take the F(4x4,3x3) path where the operator layer's rule allows (W % 4 == 0, H >= 8, >= 64 channels, >= 128 tiles) */
       PRN_BLK_HANDOVER = 2,      /* the producer -> BatchNorm hand-overs described above */
       PRN_BLK_KEEP_V = 4,        /* conv2's Winograd input transform stays in `save` for its weight gradient (up to 128 MB) */
       PRN_BLK_SCATTER_ACC = 8 }; /* stride-2 downsample blocks: prn_bottleneck_train_bwd may be asked to ADD into a dx that holds another gradient */
enum { PRN_BLK_CONV2_DIRECT = 0, PRN_BLK_CONV2_WINOGRAD = 1, PRN_BLK_CONV2_DCN = 2 };
typedef struct prn_bottleneck_desc {
  int32_t B, C, H, W;        /* block input [B,C,H,W] */
  int32_t planes;            /* bottleneck width P; the block's output is [B, 4P, Ho, Wo] */
  int32_t stride;            /* of conv2 and of the downsample convolution: 1 or 2 */
  int32_t dcn;               /* conv2 = offset|modulator conv (27 channels) + DCNv2 on its raw output (models/dcn.py:52-67) */
  int32_t downsample;        /* identity path = conv1x1(stride) + BatchNorm (models/backbone.py:156-166) */
  int32_t flags;             /* PRN_BLK_* */
  float eps[4], momentum[4]; /* bn1, bn2, bn3, downsample BatchNorm */
  float max_offset;          /* dcn: the offset clamp, max(h, w) / 4 of the block's conv2 input */
  int32_t reserved;
  prn_gemm_opts opts;
} prn_bottleneck_desc;
/* Device pointers of a block's parameters and of the derived operand layouts the CALLER keeps current (NULL where the plan does not read one:
 * prn_bottleneck_plan_info's PRN_BLK_NEEDS_* say which images a plan uses; a NULL image makes the launch cut the weight itself, into ws). */
typedef struct prn_bottleneck_params {
  const float *w1, *w2, *w3, *wd;                 /* [P,C,1,1], [P,P,3,3] (dcn: regular_conv.weight), [4P,P,1,1], [4P,C,1,1] */
  const float *b2;                                /* dcn: regular_conv.bias or NULL */
  const void *w1_img, *w3_img, *wd_img;           /* split-kernel images (prn_split_prepare) of w1 / w3 / wd (stride 1 only) */
  const float* u2; const void* u2_img;            /* conv2 in the Winograd domain [36][P][P] (prn_winograd_weights_batched: u) and its images */
  const float *w1_t, *w2_t, *w3_t, *wd_t;         /* input-gradient layouts (prn_weight_flip_transpose): [C,P,1,1], [P,P,3,3], [P,4P,1,1], [C,4P,1,1] */
  const void *w1_t_img, *w3_t_img, *wd_t_img;
  const float* ut2; const void* ut2_img;          /* prn_winograd_weights_batched: ut */
  const float *w27, *b27, *w27_t;                 /* dcn: merged offset(18) | modulator(9) weights [27,P,3,3], bias [27], input-gradient layout [P,27,3,3] */
  const float* w2_cols_t; const void* w2_cols_t_img;   /* dcn: regular_conv.weight as [9P, P] transposed (the column-gradient GEMM's operand) and its images */
  const float *gamma[4], *beta[4];
  float *running_mean[4], *running_var[4];
} prn_bottleneck_params;
enum { PRN_BLK_SAVE_BYTES = 0, PRN_BLK_GSAVE_BYTES, PRN_BLK_FWD_WS_BYTES, PRN_BLK_BWD_WS_BYTES, PRN_BLK_HO, PRN_BLK_WO, PRN_BLK_CONV2_PATH, PRN_BLK_KEEPS_V,
       PRN_BLK_OFF_A1, PRN_BLK_OFF_V, PRN_BLK_OFF_A2, PRN_BLK_OFF_OM, PRN_BLK_OFF_TABLE,        /* byte offsets into save: bn1's / bn2's outputs, V, the raw offset map, the gather table */
       PRN_BLK_OFF_D1, PRN_BLK_OFF_D2, PRN_BLK_OFF_D3, PRN_BLK_OFF_DD, PRN_BLK_OFF_DOM,   /* byte offsets into gsave */
       PRN_BLK_BN_GRAD_FLOATS,
       PRN_BLK_HANDOVERS,         /* bit 0: conv1 sums in bn1, 1: bn1 writes V, 2: conv2's output transform in bn2, 3 / 4 / 5: their backward counterparts */
       PRN_BLK_NEEDS_W1_IMG, PRN_BLK_NEEDS_W3_IMG, PRN_BLK_NEEDS_WD_IMG, PRN_BLK_NEEDS_W1T_IMG, PRN_BLK_NEEDS_W3T_IMG, PRN_BLK_NEEDS_WDT_IMG, PRN_BLK_NEEDS_U_IMG,
       PRN_BLK_NEEDS_COLT_IMG, PRN_BLK_INFO_COUNT };
int64_t prn_bottleneck_plan_bytes(void);
int64_t prn_bottleneck_params_bytes(void);
int prn_bottleneck_plan(const prn_bottleneck_desc* d, void* plan);
int prn_bottleneck_plan_info(const void* plan, int64_t* out, int n);
int prn_bottleneck_train_fwd(const void* plan, const prn_bottleneck_params* p, const float* x, float* y, void* save, void* ws, void* stream);
/* dy: gradient of y.  dx: gradient of x, fully written -- or, dx_accumulate != 0 (stride-2 downsample blocks planned with PRN_BLK_SCATTER_ACC): dx already
 * holds the gradient the block's other readers sent back (models/backbone.py:45: the stage outputs also feed the FPN and the depth decoder) and the
 * block's own gradient is added to it in place (no zero fill of the strided input gradient, no separate sum over the full-size map). */
int prn_bottleneck_train_bwd(const void* plan, const prn_bottleneck_params* p, const float* x, const float* y, const float* dy, float* dx, int dx_accumulate,
                             const void* save, void* gsave, float* bn_grads, void* ws, void* stream);

/* prn_frame_to_input -- input staging of the inference entry point in ONE launch (simple_inference.py:143-152): src = the decoded
 *   uint8 BGR frame [Hs][Ws][3] on the device (uploaded as bytes); cv2.resize(INTER_LINEAR) to Hr x Wr in OpenCV's fixed-point
 *   arithmetic, zero padding to Hp x Wp (funcs.py:204-210), FastBaseTransform (augmentations.py:496-530): mode 0 (x - mean) / std,
 *   1 x - mean, 2 x / 255 (transform.to_float), 3 unchanged, BGR -> RGB.  mean_bgr / std_bgr: HOST arrays of three floats.  dst [3][Hp][Wp] float; frame_bgr
 *   (may be NULL) [Hp][Wp][3] float: the resized, padded frame itself (what the caller draws on).                            */
int prn_frame_to_input(const unsigned char* src, int Hs, int Ws, int Hr, int Wr, int Hp, int Wp, const float* mean_bgr, const float* std_bgr, int mode,
                       float* dst, float* frame_bgr, void* stream);

/* ---- joint loss: fused per-term reductions (csrc/prn_loss.hip) -------------------------------------------------------
 * Instance masks: Dice (models/functions/losses.py:69-118,355-368) + lava (losses.py:169-197,288-329) in one pass.
 *   logits [P][HW]  raw dynamic-conv outputs of the P positive grid cells (sigmoid applied inside), rows grouped by image
 *   labels [P][HW]  uint8 0/1 targets;  img [P] int64 image of each row;  adj [B][HW] depth-gradient map pulled back to mask
 *   resolution, gsum [B] its full-resolution sum, npos [B] rows per image (adj / gsum / npos NULL: no lava term)
 *   out2 = { ins = w_ins * mean_i (1 - 2 sum(p t) / (sum p^2 + sum t^2 + 0.002)),
 *            lav = w_lav * mean over images with gsum > 0 and npos > 0 of  sum_i sum(p adj) / (gsum * npos) }
 *   coef [P][3]: per-row coefficients kept for prn_mask_loss_bwd; ws: prn_mask_loss_ws_floats(P) floats.
 *   backward: dlogits [P][HW] from the upstream gradients of the two scalars (device scalars; NULL = 0).  HW % 4 == 0, B <= 64. */
int prn_mask_loss_ws_floats(int P);
int prn_mask_loss_fwd(const float* logits, const unsigned char* labels, const float* adj, const int64_t* img, const float* gsum, const float* npos,
                      float* out2, float* coef, float* ws, int P, int HW, int B, float w_ins, float w_lav, void* stream);
int prn_mask_loss_bwd(const float* logits, const unsigned char* labels, const float* adj, const int64_t* img, const float* coef, const float* g_ins,
                      const float* g_lav, float* dlogits, int P, int HW, void* stream);

/* Depth-gradient weights of the lava term from the GT depth alone (models/functions/losses.py:288-329): out[b][y][x] = w, w = min(S / max(gt, res)^2, 1e-2),
 * set to 0 where w < 1e-4; S = gx^2 + gy^2 of the 3x3 Sobel / 8 on the reflect-padded map.  One launch for the ~20 elementwise launches of the tensor form. */
int prn_lava_gt_weights(const float* gt, float* out, int B, int H, int W, float depth_resolution, void* stream);

/* Virtual-normal loss, per-triplet part (models/functions/vnl.py:57-165 for all planes of all images at once):
 *   pred / gt [B*H*W] depths, gid [3][n] int32 cloud-point index of each triplet's points, seg [n] its segment (plane or
 *   non-planar region of one image), per segment: is_plane (uint8), plane normal (double[3]), image; fx / fy [B] (double).
 *   -> loss [n] double = 1 - |cos(predicted normal, plane normal | GT normal)|, valid [n] uint8 (the filter of vnl.py:71-104:
 *   on the predicted cloud for planes, on the GT cloud for the non-planar region), g3 [n][3] = d loss / d pred depth of the
 *   triplet's three points (forward-mode differentiation in the same pass; 0 where the loss is NaN).
 *   prn_vnl_scatter: d_depth[r] = sum over (triplet, point) pairs that sampled cloud point r of g_loss[t] * g3[t][p];
 *   order = argsort of the flattened gid, start [npts + 1] = exclusive prefix sum of the per-point counts (fixed order).  */
int prn_vnl_triplets(const float* pred, const float* gt, const int* gid, const int64_t* seg, const unsigned char* seg_is_plane, const double* seg_normal,
                     const int64_t* seg_img, const double* fx, const double* fy, double* loss, unsigned char* valid, float* g3, int n, int H, int W,
                     float delta_z, void* stream);
int prn_vnl_scatter(const double* g_loss, const float* g3, const int64_t* order, const int64_t* start, float* d_depth, int npts, int n, void* stream);

/* Category and depth terms of the joint loss, one pass each way (fp64 partials in a fixed order; ws: prn_loss_ws_doubles(B) doubles):
 *   prn_focal_sum: x [rows][C] logits, labels [rows] int64 class of the row's cell (C = background: no positive class);
 *     out[0] = sum over (row, c) of w * BCEwithLogits(x, t) * (1 - p_t)^gamma, t = (labels[row] == c), p_t = t ? sigmoid(x) : 1 - sigmoid(x),
 *     w = alpha >= 0 ? (t ? alpha : 1 - alpha) : 1   (models/functions/losses.py:121-138; the caller divides by num_pos + 1);  bwd: dx = g_out[0] * d out / dx
 *   prn_rmse_log: pred, gt [B][HW]; valid = gt > min_depth; out[0] = mean_b sqrt( sum_valid (log max(pred, clamp) - log max(gt, clamp))^2 / n_valid_b )
 *     (losses.py:142-147,371-392); coef [B] is kept for the backward pass: dpred = g_out[0] * coef[b] * (log pred - log gt) / pred on valid pixels. */
int prn_loss_ws_doubles(int B);
int prn_focal_sum_fwd(const float* x, const int64_t* labels, float* out, double* ws, int64_t rows, int C, float alpha, float gamma, void* stream);
int prn_focal_sum_bwd(const float* x, const int64_t* labels, const float* g_out, float* dx, int64_t rows, int C, float alpha, float gamma, void* stream);
int prn_rmse_log_fwd(const float* pred, const float* gt, float* out, float* coef, double* ws, int B, int HW, float min_depth, float clamp, void* stream);
int prn_rmse_log_bwd(const float* pred, const float* gt, const float* coef, const float* g_out, float* dpred, int B, int HW, float min_depth, float clamp,
                     void* stream);

/* Virtual-normal trimming rule (models/functions/vnl.py:106-117,133-165): per sampled region the valid losses sorted ascending, the lowest
 * quarter dropped, the rest averaged (NaN losses count as kept zeros); per image the regions' means summed over (planes + 1 if the non-planar
 * region had a valid triplet).  loss [n] double, valid [n], seg [n] region of each triplet (contiguous runs), seg_start [nseg];
 *   prn_vnl_trim_key: key = 4 seg + (valid ? (NaN ? 1.5 : loss) : 2) -- sort it (any sort), order = the permutation;
 *   prn_vnl_trim_fwd: out [B]; seg_sum / seg_m / seg_coef [nseg] are kept for the backward pass; ws: prn_vnl_trim_ws_bytes(nseg) bytes (every region
 *                     is summed in 16 chunks, added in chunk order: the regions' sizes differ by three orders of magnitude);
 *   prn_vnl_trim_bwd: dloss [n] = g_out[image] * d out / d loss (one coefficient per kept triplet, 0 elsewhere). */
int prn_vnl_trim_key(const double* loss, const unsigned char* valid, const int64_t* seg, double* key, int n, void* stream);
int prn_vnl_trim_fwd(const double* loss, const unsigned char* valid, const int64_t* order, const int64_t* seg_start, const unsigned char* seg_is_plane,
                     const int64_t* seg_img, const double* nplanes, int nseg, int n, int B, double* out, double* seg_sum, int* seg_m, double* seg_coef,
                     void* ws, void* stream);
int64_t prn_vnl_trim_ws_bytes(int nseg);
int prn_vnl_trim_bwd(const double* loss, const int64_t* order, const int64_t* seg_start, const int* seg_m, const double* seg_coef, const int64_t* seg_img,
                     const double* g_out, int nseg, int n, double* dloss, void* stream);

/* ---- depth-error metrics of one frame ------------------------------------------------------------------------------
 * replaces the ~25 elementwise / boolean-index / reduction launches of compute_depth_metrics (eval.py:164-207):
 * over the pixels with gt > 0.5 and pred > 0.5, pred clamped to [min_depth, max_depth] (cfg.dataset):
 *   out[0..7] = abs_rel, sq_rel, rmse, log10, a1, a2, a3 (thresholds 1.25, 1.25^2, 1.25^3), number of valid pixels   (doubles)
 * ws: prn_depth_metrics_ws_doubles() doubles (fixed-order partials -> deterministic).  The median ratio of the reference's
 * return tuple is a selection, not a sum, and stays with the caller. */
int prn_depth_metrics_ws_doubles(void);
int prn_depth_metrics(const float* pred, const float* gt, double* out, double* ws, int64_t n, float min_depth, float max_depth, void* stream);

/* ---- pairwise IoU of one frame's detections against its ground truth ----------------------------------------------------
 * replaces mask_iou / bbox_iou as compute_segmentation_metrics calls them (eval.py:214-215; models/functions/funcs.py:58-71
 * and :9-55): masks_a [A, HW] and masks_b [B, HW] are byte masks (non-zero = set; the reference multiplies their 0/1 float
 * copies), boxes [A, 4] / [B, 4] are fp32 (x1, y1, x2, y2).
 *   mask_iou[a][b] = |a & b| / (|a| + |b| - |a & b|)   (integer counts, fp32 quotient: bit-identical to the reference; 0/0 = NaN)
 *   box_iou[a][b]  = inter / (area_a + area_b - inter), inter = clamp(min(x2) - max(x1), 0) * clamp(min(y2) - max(y1), 0)
 * Either pair of inputs may be NULL (then its output is not written).  ws: prn_pairwise_iou_ws_bytes(A, B, HW) bytes
 * (the bit-packed masks and their areas).  1 <= A, B < 65536, HW < 2^24. */
int64_t prn_pairwise_iou_ws_bytes(int A, int B, int64_t HW);
int prn_pairwise_iou(const unsigned char* masks_a, const unsigned char* masks_b, const float* boxes_a, const float* boxes_b, int A, int B, int64_t HW,
                     float* mask_iou, float* box_iou, void* ws, void* stream);
/* boxes[n,4] = (x0, y0, x1, y1) of the set pixels of n byte masks [n,H,W] (non-zero = set), as floats: the tight boxes the
 * reference derives per instance with torch.where (planerecnet.py:282-287); (H+W, H+W, -1, -1) for an empty mask. */
int prn_mask_boxes(const unsigned char* masks, int n, int H, int W, float* boxes, void* stream);
/* count[r] = #{seg[r][p] > thr}, msum[r] = sum of those values, for n rows of HW floats: seg_masks.sum((1,2)) and
 * (seg_preds * seg_masks.float()).sum((1,2)) of the post-process (planerecnet.py:227-240) in one pass; fixed summation order per row,
 * independent of n. */
int prn_mask_stats(const float* seg, int n, int64_t HW, float thr, float* count, float* msum, void* stream);
/* out[b][(y * S + x)][c] = s if s == max(s over the window {y-1, y} x {x-1, x}) else 0, s = sigmoid(x[b][c][y][x]): the category scores of one
 * S x S grid level after the reference's sigmoid + point_nms (planerecnet.py:113, models/functions/nms.py:8-12), channels-last; `out` points at
 * the level's first row of the [B, cells, C] matrix over all levels (out_batch_stride = cells * C). */
int prn_sigmoid_point_nms(const float* x, float* out, int B, int C, int S, int64_t out_batch_stride, void* stream);
/* Matrix NMS score decay (models/functions/nms.py:15-50) from the [n, n] mask-IoU matrix of the detections in descending score order
 * (prn_pairwise_iou of the masks with themselves), their labels and scores: out[j] = scores[j] * min_i kernel(decay[i][j]) / kernel(comp_i),
 * gaussian (exp(-sigma x^2)) or linear (1 - x) -- the dense torch form's ~15 [n, n] passes in two launches, same operations per element.
 * ws: n floats. */
int prn_matrix_nms(const float* iou, const int64_t* labels, const float* scores, int n, float sigma, int gaussian, float* out, float* ws, void* stream);

/* ---- GT-only preparation of the loss on the device (csrc/prn_targets.hip) ---------------------------------------------------
 * The per-instance mask work of SOLOv2's target assignment (models/functions/losses.py:200-286: centre of mass, empty-mask flag,
 * the 1/4-scale masks) and the virtual-normal triplet sampling (models/functions/vnl.py:43-70,119-140), which the reference runs
 * on the host per image / per plane.  masks: uint8 [Ntot][H][W], the plane masks of all B images back to back; img_first [B+1]
 * (device): first mask of every image.  REGIONS: r < Ntot = plane mask r; r = Ntot + b = the pixels of image b no plane covers.
 *   prn_gt_mask_stats     -> totals [Ntot+B][3] uint64 = (pixels, sum of x, sum of y) per region (exact integers), segcnt
 *                            [Ntot+B][H*W/64] set pixels per 64-pixel segment, segstart [Ntot+B][H*W/64 + 1] their exclusive prefix
 *   prn_gt_quarter_masks  -> out [N][H/4][W/4]: cv2 INTER_LINEAR at exactly 1/4 (mean of the 2x2 centre of each 4x4 block, rounded
 *                            half up) -- imrescale(mask, 0.25) of losses.py:243-247
 *   prn_gt_sample_triplets: n_tot triplets; triplet t belongs to sampled region list entry trip_seg[t] (seg_region / seg_img: its
 *                            region and image); its three points are the pixels of RANK r_j in that region (raster order, i.e.
 *                            np.flatnonzero(mask)[r_j]).  ranks [3][n_tot] != NULL: injected ranks (the reference's numpy stream
 *                            drawn by the caller: bit-identical to the host path); NULL: drawn on the device (Philox4x32-10, counter
 *                            = (t, call), key = seed: `seed` identifies the run and the rank, `call` the batch -- the global
 *                            iteration, so a resumed run continues the stream), rank = floor(u * pixels).
 *                            gid [3][n_tot] = image * H*W + pixel.                                                           */
int64_t prn_gt_segments(int H, int W);
int prn_gt_mask_stats(const unsigned char* masks, const int* img_first, int B, int Ntot, int H, int W, unsigned char* segcnt, int* segstart,
                      unsigned long long* totals, void* stream);
int prn_gt_quarter_masks(const unsigned char* masks, unsigned char* out, int N, int H, int W, void* stream);
int prn_gt_sample_triplets(const unsigned char* masks, const int* img_first, int B, int Ntot, int H, int W, const int* segstart, const int* trip_seg,
                           const int* seg_region, const int* seg_img, const int* ranks, unsigned long long seed, unsigned long long call, int64_t n_tot, int* gid,
                           void* stream);

/* ---- optimizer step ------------------------------------------------------------------------------------------------------
 * replaces optimizer.step() of the reference's optim.Adam (train.py:251-256,362; no weight decay, no amsgrad): every
 * parameter tensor of the model in ONE launch.  Tables (device memory, built once by the caller): chunks [nchunks][2] =
 * (tensor index, first element) covering each tensor in pieces of prn_adam_chunk_elems() elements; p / m / v [ntensors]
 * device pointers, numel [ntensors], lr [ntensors] (a group's learning rate, per tensor); g [ntensors] is rewritten every
 * step (gradients are fresh allocations).  step [ntensors] (device): per tensor, the number of updates applied so far (advanced
 * here) -- optim.Adam's state['step'].  found_inf (device scalar or NULL): non-zero => nothing is updated and the counters stay
 * (the collective skip of train.py:353).
 * grad_scale (device scalar or NULL): gradients are divided by it.
 *   m = m + (1 - b1)(g - m);  v = b2 v + (1 - b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps),  t = step + 1 */
int prn_adam_chunk_elems(void);
int prn_adam_step(const int* chunks, int nchunks, int ntensors, float* const* p, const float* const* g, float* const* m, float* const* v, const int* numel,
                  const float* lr, float* step, const float* found_inf, const float* grad_scale, double beta1, double beta2, float eps, void* stream);
/* The same for data-parallel runs: present [any length] (device) holds, per parameter of the gradient exchange, how many ranks
 * produced a gradient for it (all-reduced); present_idx [ntensors] maps this launch's tensors into it.  A tensor whose count is 0 is
 * left untouched -- parameter, exp_avg, exp_avg_sq and its step counter -- exactly as optim.Adam skips a parameter whose .grad is None
 * (train.py:362), without a device-to-host round trip.  Both NULL: prn_adam_step. */
int prn_adam_step_masked(const int* chunks, int nchunks, int ntensors, float* const* p, const float* const* g, float* const* m, float* const* v, const int* numel,
                         const float* lr, float* step, const float* found_inf, const float* grad_scale, double beta1, double beta2, float eps,
                         const float* present, const int* present_idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif
