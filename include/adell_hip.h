/*
 * adell_hip.h -- C ABI of libadellhip.so, the MI355X (gfx950) kernel library
 * behind the adell_mri U-Net / UNETR forward-backward path.
 *
 * The reference (CCIG-Champalimaud/adell-mri) is pure Python on torch and has
 * no FFI boundary of its own: the operators on its hot path are stock
 * torch.nn calls at the call sites cited on each entry point below (paths are
 * relative to the reference root). This header is the boundary a maintainer
 * binds instead of those calls (ctypes stub: INTEGRATION.md).
 *
 * Conventions
 *  - all tensors are fp32, dense, NDHWC ("channels_last_3d": the memory a
 *    torch tensor of logical shape [N,C,D,H,W] has after
 *    .contiguous(memory_format=torch.channels_last_3d));
 *  - pointers are device pointers unless stated; `stream` is a hipStream_t
 *    (NULL = default stream); nothing synchronises the host;
 *  - every function returns ADELL_OK or a negative ADELL_E_* code and never
 *    throws; adell_last_error() gives the message of the calling thread's
 *    last failure.
 */
#ifndef ADELL_HIP_H
#define ADELL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared in this header
 * are exported (their definitions inherit the visibility of these declarations). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define ADELL_OK 0
#define ADELL_E_BADARG (-1)
#define ADELL_E_UNSUPPORTED (-2)
#define ADELL_E_HIP (-3)
#define ADELL_E_NOMEM (-4)

#define ADELL_ABI_VERSION 2

/* activation ids (reference: adell_mri/modules/activations.py:6-31) */
enum {
  ADELL_ACT_IDENTITY = 0,
  ADELL_ACT_SILU = 1, /* "swish" */
  ADELL_ACT_RELU = 2,
  ADELL_ACT_LEAKY_RELU = 3,
  ADELL_ACT_PRELU = 4,
  ADELL_ACT_GELU = 5,
  ADELL_ACT_SIGMOID = 6,
  ADELL_ACT_TANH = 7,
  ADELL_ACT_ELU = 8
};

int adell_abi_version(void);
const char* adell_last_error(void);

/* Launch-plan epoch. Every entry point that writes per-block partial sums (`stat_partials`,
 * `partials`) lays them out as [N][rows][C][2] with `rows` decided by the launch plan, and the
 * plan depends on process-wide switches (adell_set_tuning). ABI version 2: each such entry point
 * takes `partial_rows` = the rows per batch item the caller sized the buffer for (from the matching
 * *_ntiles query) and returns ADELL_E_BADARG -- before anything is launched -- when the plan of the
 * call would write a different number. adell_plan_epoch() changes whenever a switch changes, so a
 * caller may cache an *_ntiles answer per epoch. (Round 3 had a fault here: a row count cached on
 * the host across an adell_set_tuning flip made a kernel write twice the rows it had been given.) */
long adell_plan_epoch(void);

/* ------------------------------------------------------------------------
 * 3D convolution. Replaces torch.nn.Conv3d at unet.py:260-273 (conv_block_3d),
 * res_blocks.py:150-178 (ResidualBlock3d), unet.py:641-655 (final layer); the
 * two sources x0/x1 are the operands of torch.concat at unet.py:817 (never
 * materialised); `residual` is the "+ X" of res_blocks.py:192.
 * ---------------------------------------------------------------------- */
typedef struct adell_conv3d_desc {
  int32_t N, D, H, W; /* input batch / spatial size */
  int32_t C0, C1;     /* channels of source 0 and source 1 (C1 may be 0) */
  int32_t Cout;
  int32_t KD, KH, KW; /* 1..3 */
  int32_t SD, SH, SW; /* 1..2 */
  int32_t PD, PH, PW;
  int32_t Do, Ho, Wo; /* must equal (D + 2P - K) / S + 1 */
} adell_conv3d_desc;

/* Repack weights from the torch layout into the GEMM-B layouts the kernels
 * read. mode 0: conv [Cout=dim0][Cin=dim1][taps] -> [tap][Cin][Cout] (forward);
 * mode 1: same source -> [flipped tap][Cout][Cin] (backward-data);
 * mode 2: convT [Cin=dim0][Cout=dim1][8] -> [Cin][8][Cout] (forward);
 * mode 3: same source -> [tap][Cout][Cin] (backward-data). */
int adell_pack_weight(const float* w, float* out, int mode, int dim0, int dim1,
                      int KD, int KH, int KW, void* stream);

/* Rows of the statistics-partials buffer adell_conv3d_fwd writes per batch
 * item (buffer shape [N][ntiles][Cout][2] floats), or a negative error. */
int adell_conv3d_fwd_ntiles(const adell_conv3d_desc* d);

/* y = conv(concat(x0,x1), w) + bias + residual.  bias, residual, x1 and
 * stat_partials may be NULL. When stat_partials is given, per-block
 * per-channel (sum, sum of squares) of y are written for adell_stats_finalize
 * (InstanceNorm3d / BatchNorm3d statistics without re-reading y). */
int adell_conv3d_fwd(const adell_conv3d_desc* d, const float* x0, const float* x1,
                     const float* w_packed, const float* bias,
                     const float* residual, float* y, float* stat_partials,
                     int partial_rows, void* stream);

/* dX of the convolution above (autograd mirror of the same call sites).
 * dx0 receives channels [0,C0), dx1 channels [C0,C0+C1). */
int adell_conv3d_bwd_data(const adell_conv3d_desc* d, const float* dy,
                          const float* w_packed_bwd, float* dx0, float* dx1,
                          void* stream);

/* The same two operations on the f16 MFMA with error-compensated operand splitting
 * ("f16x3": a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulate, ~2^-22 relative
 * per product; per-chunk / per-layer power-of-two scaling inside the kernel). Tensors
 * stay fp32; only the packed weights differ: adell_pack_weight_f16x3 (mode 0 forward,
 * mode 1 backward-data) writes adell_pack_weight_f16x3_bytes(...) bytes of split fp16
 * tiles plus wscale: one undo factor per GEMM column (mode 0: Cout floats, mode 1: Cin). */
long adell_pack_weight_f16x3_bytes(int mode, int dim0, int dim1, int taps);
int adell_pack_weight_f16x3(const float* w, void* out, float* wscale, int mode, int dim0,
                            int dim1, int KD, int KH, int KW, void* stream);
/* The same for many weights in one launch: `table` is a DEVICE array of `entries` rows of 8
 * int64 {w pointer, out pointer, wscale pointer, mode, dim0, dim1, taps, first block}, rows
 * ordered by first block (= running sum of the GEMM-column counts: dim0 for mode 0, dim1 for
 * mode 1); total_blocks = that sum over all rows. */
int adell_pack_weight_f16x3_multi(const long* table, int entries, long total_blocks, void* stream);
int adell_conv3d_fwd_ntiles_f16x3(const adell_conv3d_desc* d);
/* the same for adell_conv3d_fwd_f16x3_ws (the split-K layers write one row per voxel range of
 * their fold, not per brick) */
int adell_conv3d_fwd_ntiles_f16x3_ws(const adell_conv3d_desc* d);
/* in_absmax / dy_absmax (optional, one zero-initialised uint32 on the device): receive
 * the bit pattern of the absmax of the kernel's input tensor(s) as a by-product;
 * adell_conv3d_bwd_weight_f16x3 takes them as its operand scales. */
int adell_conv3d_fwd_f16x3(const adell_conv3d_desc* d, const float* x0, const float* x1,
                           const void* w_split, const float* wscale, const float* bias,
                           const float* residual, float* y, float* stat_partials,
                           int partial_rows, uint32_t* in_absmax, void* stream);
int adell_conv3d_bwd_data_f16x3(const adell_conv3d_desc* d, const float* dy,
                                const void* w_split_bwd, const float* wscale, float* dx0,
                                float* dx1, uint32_t* dy_absmax, void* stream);

/* adell_conv3d_fwd_f16x3 with split-row sources (see adell_norm_act_fwd_split): xk0 / xk1 non-NULL
 * = that source is rows with exponents xk[N][C / 16], NULL = fp32. 3x3x3 stride-1 layers whose
 * launch plan is a specialised instance: adell_conv3d_f16x3_rows_ok(d) == 1 (else
 * ADELL_E_UNSUPPORTED). stat_partials rows: adell_conv3d_fwd_ntiles_f16x3(d). */
int adell_conv3d_f16x3_rows_ok(const adell_conv3d_desc* d);
int adell_conv3d_fwd_f16x3_rows(const adell_conv3d_desc* d, const void* x0, const int* xk0,
                                const void* x1, const int* xk1, const void* w_split,
                                const float* wscale, const float* bias, const float* residual,
                                float* y, float* stat_partials, int partial_rows,
                                uint32_t* in_absmax, void* stream);

/* Backward-data of a stride-2, k = 3 convolution by parity classes: dX[2i + p], p in {0,1}^3, is a
 * stride-1 convolution of dY with the sub-kernel w[:, :, t0z::2, t0y::2, t0x::2] (t0 = (p + P) mod 2
 * per axis), written onto the stride-2 lattice of dX. w_split[c] / wscale[c] (c = 4 pz + 2 py + px):
 * that sub-kernel packed with adell_pack_weight_f16x3 mode 1. Even input dims, C1 = 0. */
int adell_conv3d_bwd_data_s2_f16x3(const adell_conv3d_desc* d, const float* dy,
                                   const void* const* w_split, const float* const* wscale,
                                   float* dx, uint32_t* dy_absmax, void* stream);
/* ... + add0 ([N, D, H, W, C0] like dx: the gradient another consumer of the same input produced,
 * e.g. the decoder's half of a U-Net skip fork, unet.py:768-822) added in every class launch's
 * epilogue instead of by a separate full-size pass. */
int adell_conv3d_bwd_data_s2_f16x3_add(const adell_conv3d_desc* d, const float* dy,
                                       const void* const* w_split, const float* const* wscale,
                                       const float* add0, float* dx, uint32_t* dy_absmax,
                                       void* stream);

/* The U-Net downsampling layer (32 -> 32 channels, k = 3, stride 2, padding 1, even input dims:
 * unet.py:571-579) in ONE launch: a persistent block keeps the split weight in LDS, stages the dY
 * halo of a brick once and produces all eight parity classes of the dX voxels behind it
 * (csrc/conv_dgrad_s2.hip). w_split_bwd / wscale: adell_pack_weight_f16x3 mode 1 of the full
 * weight; add0: null or as above. _applicable: 1 when the layer qualifies. */
int adell_conv3d_bwd_data_s2_fused_applicable(const adell_conv3d_desc* d);
/* The forward of the same layer in one persistent launch (csrc/conv_fwd_s2.hip): the eight parity
 * sub-lattices of x behind an output brick are staged one after the other (a 51 KB halo each) and
 * accumulate into the same registers; bias, statistics partials ([N][_ntiles][32][2]) and the
 * absmax by-product as adell_conv3d_fwd_f16x3. w_split / wscale: adell_pack_weight_f16x3 mode 0. */
int adell_conv3d_fwd_s2_fused_applicable(const adell_conv3d_desc* d);
int adell_conv3d_fwd_s2_fused_ntiles(const adell_conv3d_desc* d);
int adell_conv3d_fwd_s2_fused(const adell_conv3d_desc* d, const float* x, const void* w_split,
                              const float* wscale, const float* bias, float* y,
                              float* stat_partials, int partial_rows, uint32_t* in_absmax,
                              void* stream);
int adell_conv3d_bwd_data_s2_fused(const adell_conv3d_desc* d, const float* dy,
                                   const void* w_split_bwd, const float* wscale,
                                   const float* add0, float* dx, uint32_t* dy_absmax, void* stream);

/* The same two calls with a caller-provided workspace (adell_conv3d_splitk_workspace bytes; 0 =
 * never needed): layers with too few output bricks to fill the chip (the 8^3 - 16^3 levels)
 * share the channel chunks of a brick out over several blocks (split-K) and fold the partial
 * outputs in fixed order -- bit-reproducible, bias / residual / statistics as in the plain call.
 * Without a workspace (or with one that is too small) they behave like the plain calls. */
long adell_conv3d_splitk_workspace(const adell_conv3d_desc* d, int backward_data);
int adell_conv3d_fwd_f16x3_ws(const adell_conv3d_desc* d, const float* x0, const float* x1,
                              const void* w_split, const float* wscale, const float* bias,
                              const float* residual, float* y, float* stat_partials,
                              int partial_rows, uint32_t* in_absmax, void* workspace,
                              size_t workspace_bytes, void* stream);
int adell_conv3d_bwd_data_f16x3_ws(const adell_conv3d_desc* d, const float* dy,
                                   const void* w_split_bwd, const float* wscale, float* dx0,
                                   float* dx1, uint32_t* dy_absmax, void* workspace,
                                   size_t workspace_bytes, void* stream);
/* the same with add0 ([N][D][H][W][C0]; C1 must be 0) added to dx0 in the epilogue: the gradient a
 * residual link (res_blocks.py:192, `op(X) + X`) sends straight to the block input */
int adell_conv3d_bwd_data_f16x3_add(const adell_conv3d_desc* d, const float* dy,
                                    const void* w_split_bwd, const float* wscale,
                                    const float* add0, float* dx0, uint32_t* dy_absmax,
                                    void* workspace, size_t workspace_bytes, void* stream);

/* Backward-data fused with the backward of the norm -> dropout -> activation site(s) whose OUTPUT
 * the destination(s) are gradients of (reference: the autograd chain Conv3d <- activation <- Dropout
 * <- InstanceNorm3d of adn_fn.py:140-152 feeding unet.py:260-273 / res_blocks.py:150-178). site0 /
 * site1 describe the sites behind dx0 / dx1 (NULL: plain destination); their destinations receive
 *   dt = dout * act'(u) * keep / (1 - p),   u = dropout((y - mean) * rstd)
 * instead of dout, and `partials` ([N][ntiles][C0 + C1][2]) the per-brick sums (sum dt, sum dt xhat)
 * that adell_norm_act_bwd_from_dt folds. ntiles = adell_conv3d_bwd_data_f16x3_adn_ntiles(d); 0 means
 * the problem does not take the fused epilogue (the caller then uses the plain entry points). */
typedef struct adell_adn_site {
  const float* y;         /* the site's input, [N][D][H][W][C] like the destination */
  const float* mean;      /* [N][C] instance statistics of y */
  const float* rstd;
  const void* keep_mask;  /* keep bits written by adell_norm_act_fwd_mask; NULL iff drop_p == 0 */
  float drop_p;
  float act_p;            /* LeakyReLU slope */
  int32_t act;            /* ADELL_ACT_IDENTITY / _SILU / _RELU / _LEAKY_RELU */
} adell_adn_site;
int adell_conv3d_bwd_data_f16x3_adn_ntiles(const adell_conv3d_desc* d);
int adell_conv3d_bwd_data_f16x3_adn(const adell_conv3d_desc* d, const float* dy,
                                    const void* w_split_bwd, const float* wscale,
                                    const float* add0, float* dx0, float* dx1,
                                    uint32_t* dy_absmax, const adell_adn_site* site0,
                                    const adell_adn_site* site1, float* partials,
                                    int partial_rows, void* stream);

/* dW in torch's canonical [Cout][Cin][kD][kH][kW] layout (split-K over voxel
 * bricks, fixed-order reduction: deterministic) and, when db != NULL, the bias
 * gradient db[Cout] = sum over voxels of dy from the same pass. workspace:
 * device scratch of at least adell_conv3d_bwd_weight_workspace(d) bytes. */
long adell_conv3d_bwd_weight_workspace(const adell_conv3d_desc* d);
int adell_conv3d_bwd_weight(const adell_conv3d_desc* d, const float* x0,
                            const float* x1, const float* dy, float* dw, float* db,
                            void* workspace, size_t workspace_bytes, void* stream);

/* The same on the f16 MFMA with error-compensated splitting (f16x3); X and dY get one
 * power-of-two scale per tensor (x_absmax / dy_absmax from the forward and backward-data
 * kernels, or NULL to have them reduced inside the call). */
long adell_conv3d_bwd_weight_f16x3_workspace(const adell_conv3d_desc* d);
int adell_conv3d_bwd_weight_f16x3(const adell_conv3d_desc* d, const float* x0,
                                  const float* x1, const float* dy, float* dw, float* db,
                                  const uint32_t* x_absmax, const uint32_t* dy_absmax,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* The same with split-row sources of X (see adell_norm_act_fwd_split): xk0 / xk1 non-NULL = that
 * source holds rows, ONE exponent per tensor (xk[0]); x_absmax then covers the fp32 source only.
 * _rows_ok(d) == 1: the z-ring kernel serves the problem and no 32-channel tile straddles the two
 * sources. Workspace: adell_conv3d_bwd_weight_f16x3_workspace(d). */
int adell_conv3d_bwd_weight_f16x3_rows_ok(const adell_conv3d_desc* d);
int adell_conv3d_bwd_weight_f16x3_rows(const adell_conv3d_desc* d, const void* x0, const int* xk0,
                                       const void* x1, const int* xk1, const float* dy, float* dw,
                                       float* db, const uint32_t* x_absmax,
                                       const uint32_t* dy_absmax, void* workspace,
                                       size_t workspace_bytes, void* stream);

/* db[c] = sum over rows of dy[rows][C] (torch's bias gradient of Conv3d /
 * ConvTranspose3d). workspace >= adell_bias_grad_workspace(rows, C) bytes. */
long adell_bias_grad_workspace(long rows, int C);
int adell_bias_grad(const float* dy, long rows, int C, float* db, void* workspace,
                    size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * ConvTranspose3d(kernel=2, stride=2, padding=0): unet.py:445-458
 * (init_upscale_ops, upscale_type="transpose"), unetr.py:286-308.
 * x [N,D,H,W,Cin] -> y [N,2D,2H,2W,Cout].
 * ---------------------------------------------------------------------- */
/* General form: kernel = stride = (FD,FH,FW), each 1 or 2 (anisotropic upscaling of
 * backbone encoders, e.g. strides [2,2,1]); weights [Cin][Cout][FD][FH][FW]. */
int adell_convtranspose3d_fwd(int N, int D, int H, int W, int Cin, int Cout, int FD, int FH,
                              int FW, const float* x, const float* w_packed, const float* bias,
                              float* y, void* stream);
int adell_convtranspose3d_bwd_data(int N, int D, int H, int W, int Cin, int Cout, int FD,
                                   int FH, int FW, const float* dy, const float* w_packed_bwd,
                                   float* dx, void* stream);
/* The same two on the f16x3 kernels. Forward: w_split / wscale = adell_pack_weight_f16x3(mode 0)
 * of the virtual 1x1x1 conv weight V[(f, co)][ci] = w[ci][co][f] (F*Cout columns);
 * backward-data: mode 0 of the torch weight read as [Cin outputs][Cout inputs][taps]. */
int adell_convtranspose3d_fwd_f16x3(int N, int D, int H, int W, int Cin, int Cout, int FD, int FH,
                                    int FW, const float* x, const void* w_split,
                                    const float* wscale, const float* bias, float* y,
                                    uint32_t* in_absmax, void* stream);
int adell_convtranspose3d_bwd_data_f16x3(int N, int D, int H, int W, int Cin, int Cout, int FD,
                                         int FH, int FW, const float* dy, const void* w_split_bwd,
                                         const float* wscale, float* dx, uint32_t* dy_absmax,
                                         void* stream);
long adell_convtranspose3d_bwd_weight_workspace(int N, int D, int H, int W, int Cin, int Cout,
                                                int FD, int FH, int FW);
int adell_convtranspose3d_bwd_weight(int N, int D, int H, int W, int Cin, int Cout, int FD,
                                     int FH, int FW, const float* x, const float* dy, float* dw,
                                     void* workspace, size_t workspace_bytes, void* stream);
int adell_convtranspose3d_k2s2_fwd(int N, int D, int H, int W, int Cin, int Cout,
                                   const float* x, const float* w_packed,
                                   const float* bias, float* y, void* stream);
int adell_convtranspose3d_k2s2_bwd_data(int N, int D, int H, int W, int Cin,
                                        int Cout, const float* dy,
                                        const float* w_packed_bwd, float* dx,
                                        void* stream);
/* dw in torch's canonical [Cin][Cout][2][2][2] layout. */
long adell_convtranspose3d_k2s2_bwd_weight_workspace(int N, int D, int H, int W,
                                                     int Cin, int Cout);
int adell_convtranspose3d_k2s2_bwd_weight(int N, int D, int H, int W, int Cin,
                                          int Cout, const float* x, const float* dy,
                                          float* dw, void* workspace,
                                          size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Normalisation statistics and the fused Norm -> Dropout -> Activation of
 * ActDropNorm with ordering "NDA" (adn_fn.py:56-152; unet.py:697-714).
 * ---------------------------------------------------------------------- */
/* mean / rstd from partials [N][ntiles][C][2]; count = voxels per item; biased
 * variance, rstd = 1/sqrt(var+eps). per_item=1: [N][C] (torch.nn.InstanceNorm3d);
 * per_item=0: [C] over the whole batch (torch.nn.BatchNorm3d in training). */
long adell_stats_finalize_workspace(int N, int ntiles, int C); /* bytes, may be 0 */
int adell_stats_finalize(const float* partials, int N, int ntiles, int C,
                         long count, float eps, int per_item, float* mean,
                         float* rstd, void* workspace, size_t workspace_bytes,
                         void* stream);

/* Running statistics of a BatchNorm site in training, from the batch (mean, rstd) of
 * adell_stats_finalize(per_item = 0) over `count` elements per channel (torch.nn.BatchNorm3d:
 * running = (1 - m) running + m batch, with the unbiased variance; num_batches_tracked (int64, may be
 * NULL) += 1; momentum < 0: the cumulative average m = 1 / num_batches_tracked). One launch. */
int adell_bn_running_update(const float* mean, const float* rstd, float* running_mean,
                            float* running_var, long long* num_batches_tracked, int C, long count,
                            float eps, float momentum, void* stream);
/* Partials of an arbitrary tensor x [N][V][C] (same buffer format). */
int adell_channel_partials_ntiles(long V);
int adell_channel_partials(const float* x, int N, long V, int C, float* partials,
                           void* stream);

typedef struct adell_norm_act_desc {
  int64_t N, V;           /* batch items, voxels per item */
  int32_t C;
  int32_t stats_per_item; /* 1: mean/rstd are [N][C] (instance); 0: [C] (batch) */
  int32_t act;            /* ADELL_ACT_* */
  int32_t act_w_n;        /* PReLU weight count (1 or C) when act_w != NULL */
  float act_p;            /* LeakyReLU slope / ELU alpha */
  float drop_p;           /* dropout probability, 0 = off (eval) */
  uint64_t seed;          /* Philox key; mask is a function of (seed, offset, index) */
  uint32_t rng_offset;    /* distinguishes ADN sites sharing one seed */
} adell_norm_act_desc;

/* out = act(dropout((x - mean) * rstd * gamma + beta)); mean/rstd, gamma, beta,
 * act_w may be NULL. */
int adell_norm_act_fwd(const adell_norm_act_desc* d, const float* x,
                       const float* mean, const float* rstd, const float* gamma,
                       const float* beta, const float* act_w, float* out,
                       void* stream);

/* The same, also writing the dropout keep bits (drop_p > 0): element el of batch item n is bit
 * (el >> 2) & 63 of 64-bit word (n * groups + (el >> 8)) * 4 + (el & 3), groups = ceil(V C / 256);
 * adell_norm_act_mask_bytes(d) bytes. Needs C % 4 == 0 and a power-of-two C <= 1024 (returns
 * ADELL_E_UNSUPPORTED otherwise: the caller then keeps the regenerating backward). */
long adell_norm_act_mask_bytes(const adell_norm_act_desc* d);
int adell_norm_act_fwd_mask(const adell_norm_act_desc* d, const float* x, const float* mean,
                            const float* rstd, const float* gamma, const float* beta,
                            const float* act_w, float* out, void* keep_mask, void* stream);
/* Split rows (round 4): an activation stored as the LDS row image of the f16x3 convolution kernels
 * instead of fp32 values -- per voxel and 16-channel chunk one 64-byte row
 *   [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15]   fp16, hi = fp16(x 2^k), lo = fp16(x 2^k - hi)
 * (same bytes per element; 22 of the 24 mantissa bits, exactly what those kernels keep of an fp32
 * operand), with exponents xk[N][C / 16]. The consumer then stages its halo as a copy: no
 * block-wide absmax, no conversion on the vector ALU. Replaces the fp32 tensor between
 * `ActDropNorm` and the `Conv3d` that is its only reader (unet.py:260-273, res_blocks.py:150-178).
 *
 * adell_norm_act_fwd_split: adell_norm_act_fwd(_mask) writing rows scaled by 2^split_exp (the
 * caller knows a bound: an instance-normalised value is at most sqrt(V) in magnitude, dropout
 * scales by 1 / (1 - p), and |act(u)| <= |u| for the activations it is used with); C a power of two
 * in 16..1024; keep_mask as adell_norm_act_fwd_mask or NULL.
 * adell_split_rows_from_f32 / _to_f32: the conversions for any tensor with known exponents. */
int adell_norm_act_fwd_split(const adell_norm_act_desc* d, const float* x, const float* mean,
                             const float* rstd, const float* gamma, const float* beta,
                             const float* act_w, void* out_rows, int split_exp, void* keep_mask,
                             void* stream);
int adell_split_rows_from_f32(const float* x, int N, long V, int C, const int* xk, void* rows,
                              void* stream);
int adell_split_rows_to_f32(const void* rows, int N, long V, int C, const int* xk, float* x,
                            void* stream);
/* Second half of the site's backward when the producer of dout already applied the activation /
 * dropout derivative (adell_conv3d_bwd_data_f16x3_adn): dx = rstd * (dt - c1 - xhat * c2) with
 * c1 / c2 the means of the partial sums. partials: [N][ntiles][pstride][2], the site's channels
 * at columns [poff, poff + C). dx may alias dt. Instance statistics, no affine parameters.
 * workspace: (2 + 2 * ceil(ntiles / 256)) * N * C floats. */
int adell_norm_act_bwd_from_dt(const adell_norm_act_desc* d, const float* x, const float* dt,
                               const float* mean, const float* rstd, const float* partials,
                               int ntiles, int pstride, int poff, float* dx, void* workspace,
                               size_t workspace_bytes, void* stream);

/* dx (and optionally dgamma / dbeta [C]) of adell_norm_act_fwd; the dropout
 * mask is regenerated from (seed, rng_offset). workspace >=
 * adell_norm_act_bwd_workspace(d) bytes (needed when a norm or affine grad is
 * involved). */
long adell_norm_act_bwd_workspace(const adell_norm_act_desc* d);
int adell_norm_act_bwd(const adell_norm_act_desc* d, const float* x, const float* dout,
                       const float* mean, const float* rstd, const float* gamma,
                       const float* beta, const float* act_w, float* dx, float* dgamma,
                       float* dbeta, void* workspace, size_t workspace_bytes,
                       void* stream);
/* adell_norm_act_bwd with a LOW-RANK upstream gradient dout[v][c] = sum_o g[v][o] * w[o][c]
 * (g: [N][V][co] channels-last, w: [co][C], 1 <= co <= 4): the site in front of a 1x1x1 conv with
 * few output channels -- the logits head Conv3d -> ADN -> Conv3d(C -> n_classes, k = 1) of
 * lib/modules/segmentation/unet.py:626-655 -- takes that conv's dY and weight, and the conv's
 * backward-data tensor is neither written nor read. Instance statistics without affine
 * parameters; power-of-two C in 4..1024, 16-byte aligned x / dx; workspace as for
 * adell_norm_act_bwd. */
int adell_norm_act_bwd_lowrank(const adell_norm_act_desc* d, const float* x, const float* g,
                               const float* w, int co, const float* mean, const float* rstd,
                               float* dx, void* workspace, size_t workspace_bytes, void* stream);
/* Gradient of the PReLU weight(s) of the same fused op (torch.nn.PReLU is the reference's
 * default activation_fn): dact_w[act_w_n] with act_w_n = 1 or C; operands as above. */
long adell_prelu_wgrad_workspace(const adell_norm_act_desc* d);
int adell_prelu_wgrad(const adell_norm_act_desc* d, const float* x, const float* dout,
                      const float* mean, const float* rstd, const float* gamma, const float* beta,
                      float* dact_w, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Segmentation loss: binary generalised dice + binary focal on probabilities
 * (losses.py:14-54,251-292 with weight=1, scale=1; losses.py:112-164 with
 * alpha=1, threshold=0.5, scale=1, no label smoothing), the CompoundLoss of
 * sample_configs/u-net-3d-resnet.yaml evaluated at segmentation/pl.py:218-222.
 * prob/target are [B][S]; dice/focal are the per-item losses; sums [B][3] is
 * scratch kept for the backward.
 * ---------------------------------------------------------------------- */
long adell_dice_focal_workspace(int B, long S);
int adell_dice_focal_fwd(const float* prob, const float* target, int B, long S,
                         float smooth, float dice_eps, float gamma, float focal_alpha, float focal_eps,
                         float* dice, float* focal, float* sums, void* workspace,
                         size_t workspace_bytes, void* stream);
/* dprob = gdice * d(dice_b)/dprob + gfocal * d(focal_b)/dprob */
int adell_dice_focal_bwd(const float* prob, const float* target, int B, long S,
                         float smooth, float dice_eps, float gamma, float focal_alpha, float focal_eps,
                         const float* sums, float gdice, float gfocal, float* dprob,
                         void* stream);
/* the same with per-item upstream gradients on the device (gdice[B], gfocal[B], NULL = 0) */
int adell_dice_focal_bwd_dev(const float* prob, const float* target, int B, long S, float smooth,
                             float dice_eps, float gamma, float focal_alpha, float focal_eps,
                             const float* sums, const float* gdice, const float* gfocal,
                             float* dprob, void* stream);
/* (focal_alpha: weight of the positive-class term, `alpha` of binary_focal_loss, losses.py:112-164)
 *
 * Per-(item, class) sums (sum p t, sum p, sum t) of probabilities / targets laid out [B][V][C]: the
 * building block of the Tversky-type losses (binary_focal_tversky_loss losses.py:295-337,
 * mc_focal_tversky_loss :656-698 and, through them, hybrid_focal / unified_focal :386-462, 737-808).
 * Backward: dp[b][v][c] = gsums[b][c][0] * t[b][v][c] + gsums[b][c][1]. */
long adell_class_sums_workspace(int B, long V, int C);
int adell_class_sums_fwd(const float* p, const float* t, int B, long V, int C, float* sums,
                         void* workspace, size_t workspace_bytes, void* stream);
int adell_class_sums_bwd(const float* t, const float* gsums, int B, long V, int C, float* dp,
                         void* stream);

/* ------------------------------------------------------------------------
 * Optimiser / EMA updates over flat fp32 buffers (16-byte aligned).
 * adell_sgd_step: torch.optim.SGD(momentum, nesterov, weight_decay), the
 *   default optimiser of UNetBasePL.configure_optimizers (segmentation/pl.py:563-569);
 * adell_adamw_step: torch.optim.AdamW (self_supervised/pl.py:245-250);
 * adell_ema_update: ExponentialMovingAverage.update (utils/utils.py:447-493).
 * grad_scale multiplies the gradient first (1/world_size after a sum
 * all-reduce, 1/accumulate_grad_batches).
 * ---------------------------------------------------------------------- */
int adell_sgd_step(float* param, const float* grad, float* momentum_buf, long n, float lr,
                   float momentum, float weight_decay, int nesterov, int first_step,
                   float grad_scale, void* stream);
int adell_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                     long n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, long step, float grad_scale, void* stream);
/* torch.optim.Adam ("adam" of optimizer_factory.py:5-14): as above with the weight decay added
 * to the gradient instead of decoupled. */
int adell_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, long step,
                    float grad_scale, void* stream);

/* The other optimisers of the reference's factory (utils/optimizer_factory.py:5-14), torch.optim
 * single-tensor semantics, L2 weight decay added to the gradient: kind 0 Adamax (state1 exp_avg,
 * state2 exp_inf), 1 Adagrad (sum), 2 NAdam (exp_avg, exp_avg_sq), 3 RAdam (exp_avg, exp_avg_sq),
 * 4 RMSprop (square_avg; momentum 0, not centered). c5: HOST array of the five per-step scalars
 * documented at adell_optim_kernel (csrc/loss_optim.hip). */
int adell_optim_step(int kind, float* param, const float* grad, float* state1, float* state2,
                     long n, float weight_decay, float eps, float grad_scale, const float* c5,
                     void* stream);
int adell_ema_update(float* shadow, const float* param, long n, float decay, void* stream);

/* ------------------------------------------------------------------------
 * Token-sequence kernels of the ViT encoder (UNETR): vit.py:844-1002,
 * linear_blocks.py:358-417. (torch.nn.Linear runs on adell_conv3d_* as a
 * 1x1x1 convolution over the token axis.)
 * ---------------------------------------------------------------------- */
/* torch.nn.LayerNorm over the last dim of x [rows][C]; gamma/beta may be NULL;
 * mean/rstd [rows] are kept for the backward. */
int adell_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                        float* mean, float* rstd, long rows, int C, float eps, void* stream);
long adell_layernorm_bwd_workspace(long rows, int C);
int adell_layernorm_bwd(const float* x, const float* dy, const float* gamma,
                        const float* mean, const float* rstd, float* dx, float* dgamma,
                        float* dbeta, long rows, int C, void* workspace,
                        size_t workspace_bytes, void* stream);
/* out[i] = a[i] + b[i % period] (X + positional_embedding, vit.py:866-867) and the
 * gradient of b: db[j] = sum_k g[k*period + j]. */
int adell_add_bcast(const float* a, const float* b, float* out, long n, long period,
                    void* stream);
int adell_sum_bcast(const float* g, float* db, long n, long period, void* stream);
/* softmax(q k^T * scale + bias) v per sequence: q,k [BH][T][A]; v,out [BH][T][Dv];
 * bias [nbias][T][T] or NULL (sequence bh uses bias[bh % nbias]); lse [BH][T].
 * drop_p > 0 drops attention probabilities (dropout_p of the same call): the mask of entry
 * (bh, query, key) is a Philox function of (seed, rng_offset), regenerated by the backward.
 * F.scaled_dot_product_attention as called at linear_blocks.py:407-414. */
int adell_attention_fwd(const float* q, const float* k, const float* v, const float* bias,
                        int nbias, int BH, int T, int A, int Dv, float scale, float drop_p,
                        unsigned long long seed, unsigned int rng_offset, float* out,
                        float* lse, void* stream);
int adell_attention_bwd(const float* q, const float* k, const float* v, const float* bias,
                        int nbias, const float* out, const float* dout, const float* lse,
                        int BH, int T, int A, int Dv, float scale, float drop_p,
                        unsigned long long seed, unsigned int rng_offset, float* dq, float* dk,
                        float* dv, void* stream);
/* The same attention with every operand addressed by element strides: sequence bh = b * H + h,
 * `strides` holds (item b, head h, token row) triples in the order q, k, v, out (forward: 12
 * values) and q, k, v, out, dout, dq, dk, dv (backward: 24). Q / K / V may stay inside the packed
 * [B][T][H][q | k | v] projection output of linear_blocks.py:372-385, O / dO are [B][T][H * Dv]
 * token rows, dV can land inside the packed gradient: the slice / permute copies around
 * F.scaled_dot_product_attention (linear_blocks.py:380-417) are not launches. Strides are
 * multiples of 4 elements, pointers 16-byte aligned; MFMA-shaped heads only
 * (adell_attention_strided_ok: A, Dv in {32, 64, 128}, T >= 16), ADELL_E_BADARG otherwise. */
int adell_attention_strided_ok(int T, int A, int Dv);
int adell_attention_fwd_strided(const float* q, const float* k, const float* v, const float* bias,
                                int nbias, int B, int H, int T, int A, int Dv, const long* strides,
                                float scale, float drop_p, unsigned long long seed,
                                unsigned int rng_offset, float* out, float* lse, void* stream);
int adell_attention_bwd_strided(const float* q, const float* k, const float* v, const float* bias,
                                int nbias, const float* out, const float* dout, const float* lse,
                                int B, int H, int T, int A, int Dv, const long* strides,
                                float scale, float drop_p, unsigned long long seed,
                                unsigned int rng_offset, float* dq, float* dk, float* dv,
                                void* stream);

/* ------------------------------------------------------------------------
 * Data movement for the U-Net++ dense links (standard_blocks.py:365-371):
 * N-way channel concat / split of NDHWC tensors, and nearest-neighbour
 * resampling (F.interpolate's default mode) with its backward.
 * ---------------------------------------------------------------------- */
/* direction 0: full[v][coff + c] = part[v][c]; direction 1: the reverse copy. */
int adell_copy_channels(float* full, float* part, long V, int Cfull, int Cpart, int coff,
                        int direction, void* stream);
int adell_interp_nearest_fwd(const float* x, float* y, int N, int C, int Di, int Hi, int Wi,
                             int Do, int Ho, int Wo, void* stream);
int adell_interp_nearest_bwd(const float* dy, float* dx, int N, int C, int Di, int Hi, int Wi,
                             int Do, int Ho, int Wo, void* stream);
/* torch.nn.Upsample(scale_factor, mode="bilinear" | "trilinear", align_corners=False) of the
 * "upsample" upscaling path (unet.py:419-443): x [N][Di][Hi][Wi][C] -> y [N][Do][Ho][Wo][C] with
 * Do = floor(Di * scale_d) etc.; a 2-D tensor is Di = Do = 1, scale_d = 1. align_corners != 0:
 * F.interpolate(size=(Do,Ho,Wo), align_corners=True) (the deep-supervision targets of
 * pl.py:305-309; the scale arguments are then ignored). The backward gathers (deterministic). */
int adell_interp_linear_fwd(const float* x, float* y, int N, int C, int Di, int Hi, int Wi,
                            int Do, int Ho, int Wo, float scale_d, float scale_h, float scale_w,
                            int align_corners, void* stream);
int adell_interp_linear_bwd(const float* dy, float* dx, int N, int C, int Di, int Hi, int Wi,
                            int Do, int Ho, int Wo, float scale_d, float scale_h, float scale_w,
                            int align_corners, void* stream);

/* y[n][v][c] = x[n][v][c] * s[n][c] on NDHWC activations (x, y: [N][V][C]; s: [N][C]): the tabular
 * feature gates of the decoder (unet.py:803-810) and U-out (regularization.py:48-55). The
 * backward is the same call on dy (dx = dy * s) plus ds[n][c] = sum_v dy * x
 * (adell_scale_bc_dscale; workspace of adell_scale_bc_dscale_workspace_floats floats). */
/* out[n][z][y][ox][kx * Cin + ci] = x[n][z][y][ox - P + kx][ci] (zeros outside the row and in the
 * slots >= K * Cin), x [N][D][H][W][Cin], out [N][D][H][W + 2P - K + 1][Cp]: the x taps of a
 * small-Cin convolution folded into a 16-channel chunk, so that a Kd x Kh x K conv over Cin <= 4
 * channels runs as a Kd x Kh x 1 conv over Cp channels (first layers: unet.py:260-273,
 * res_net.py:60-130). */
int adell_fold_x_taps(const float* x, float* out, int N, int D, int H, int W, int Cin, int K,
                      int P, int Cp, void* stream);
int adell_scale_bc(const float* x, const float* s, float* y, int N, long V, int C, void* stream);
long adell_scale_bc_dscale_workspace_floats(int N, long V, int C);
int adell_scale_bc_dscale(const float* x, const float* dy, float* ds, int N, long V, int C,
                          float* workspace, void* stream);

/* Concurrent squeeze-and-excite gate used where the multi-branch U-Net merges its encoders
 * (reference adell_mri/modules/layers/self_attention.py:21-150 ConcurrentSqueezeAndExcite{2,3}d,
 * called at adell_mri/modules/segmentation/unet.py:1186-1207): y = acc + x * (s[n][v] + c[n][ch]) *
 * inv[n]. x, acc, y: [N][V][C] (NDHWC); s: [N][V] spatial gate; c: [N][C] channel gate; inv: [N] or
 * null (1); acc null = 0. Backward: dx, ds [N][V], dc [N][C] (C <= 512; workspace of
 * adell_cse_apply_bwd_workspace_floats floats). */
int adell_cse_apply(const float* x, const float* s, const float* c, const float* inv,
                    const float* acc, float* y, int N, long V, int C, void* stream);
long adell_cse_apply_bwd_workspace_floats(int N, long V, int C);
int adell_cse_apply_bwd(const float* x, const float* dy, const float* s, const float* c,
                        const float* inv, float* dx, float* ds, float* dc, int N, long V, int C,
                        float* workspace, void* stream);
/* out[n][v][ch] = g[n][ch] * scale: gradient of the per-channel spatial mean the channel gate is
 * computed from (self_attention.py:95-96, torch.flatten(X, 2).mean(-1)). */
int adell_bcast_nc(const float* g, float* out, int N, long V, int C, float scale, void* stream);

/* torch.nn.MaxPool3d (ceil_mode False, dilation 1, -inf padding): unet.py:335,368,
 * 595-603, res_net.py:180,209. Geometry in an adell_conv3d_desc (C0 = channels, C1 and
 * Cout ignored); argmax [N][Do][Ho][Wo][C] holds the winner's (z*H+y)*W+x. */
int adell_maxpool3d_fwd(const adell_conv3d_desc* d, const float* x, float* y,
                        int32_t* argmax, void* stream);
int adell_maxpool3d_bwd(const adell_conv3d_desc* d, const float* dy, const int32_t* argmax,
                        float* dx, void* stream);

/* ------------------------------------------------------------------------
 * ConvNeXt / VICReg self-supervised path (BASELINE config 4).
 * Depthwise Conv3d(groups=C, stride 1, "same" padding, odd kernels):
 * res_blocks.py:552-558. w / dw in torch's [C][1][KD][KH][KW] layout.
 * ---------------------------------------------------------------------- */
int adell_dwconv3d_fwd(int N, int C, int D, int H, int W, int KD, int KH, int KW,
                       const float* x, const float* w, const float* bias, float* y,
                       void* stream);
int adell_dwconv3d_bwd_data(int N, int C, int D, int H, int W, int KD, int KH, int KW,
                            const float* dy, const float* w, float* dx, void* stream);
/* workspace: adell_dwconv3d_bwd_weight_workspace_floats(...) floats (split partial sums of
 * the tiled kernel; may be NULL when that returns 0) */
long adell_dwconv3d_bwd_weight_workspace_floats(int N, int C, int D, int H, int W, int KD, int KH,
                                                int KW);
int adell_dwconv3d_bwd_weight(int N, int C, int D, int H, int W, int KD, int KH, int KW,
                              const float* x, const float* dy, float* dw, float* db,
                              float* workspace, void* stream);
/* VICReg terms of two [B][D] embeddings (self_supervised/losses/vicreg.py:60-140):
 * out3 = (invariance, variance, covariance), unweighted; scratch of
 * adell_vicreg_scratch_floats(B, D) floats is kept for the backward, which returns
 * g3[0]*d(inv) + g3[1]*d(var) + g3[2]*d(cov) (g3: 3 floats on the device) w.r.t. x1 (dx1) and x2 (dx2, may be NULL). */
long adell_vicreg_scratch_floats(int B, int D);
int adell_vicreg_fwd(const float* x1, const float* x2, int B, int D, float min_var, float eps,
                     float* scratch, float* out3, void* stream);
int adell_vicreg_bwd(const float* x1, const float* x2, int B, int D, float min_var, float eps,
                     const float* scratch, const float* g3, float* dx1, float* dx2,
                     void* stream);

/* Cosine-similarity losses between two [B][D] embedding batches, the non-VICReg choices of
 * SelfSLBasePL.init_loss (self_supervised/pl.py:202-212): kind 0 simsiam_loss, kind 1 byol_loss
 * (self_supervised/losses/functional.py:138-164), kind 2 NTXentLoss (losses/ntxent.py:11-46;
 * temperature, apply_relu). loss: 1 float on the device. scratch of
 * adell_pair_loss_scratch_floats(B, D) floats is kept for the backward, which returns
 * g[0] * dloss/dx1 (dx1) and / or dx2 (either may be NULL). 2 B <= 256. */
long adell_pair_loss_scratch_floats(int B, int D);
int adell_pair_loss_fwd(const float* x1, const float* x2, int B, int D, int kind,
                        float temperature, int apply_relu, float* scratch, float* loss,
                        void* stream);
int adell_pair_loss_bwd(const float* x1, const float* x2, int B, int D, int kind,
                        float temperature, int apply_relu, const float* scratch, const float* g,
                        float* dx1, float* dx2, void* stream);

/* Local contrastive loss of the semi-supervised U-Net (LocalContrastiveLoss.forward,
 * semi_supervised_segmentation/losses.py:498-526; called by UNetContrastiveSemiSL.step_semi_sl_loco,
 * semi_supervised_segmentation/pl.py:244-281): f1 / f2 = decoder features of the two views, NDHWC
 * [B][S][C] (B <= 8, C % 4 == 0). loss[i] = mean_s -log(max(softmax_j(cos(f2[i,s], f1[j,s]) / T)[i],
 * eps)). Backward: gloss[B] (device) = dL/dloss; df1 / df2 [B][S][C], either may be NULL.
 * Workspace of the forward: adell_loco_loss_workspace(B, S, C) bytes (per-block partial sums,
 * folded in fixed order). */
long adell_loco_loss_workspace(int B, long S, int C);
int adell_loco_loss_fwd(const float* f1, const float* f2, int B, long S, int C, float temperature,
                        float eps, float* loss, void* workspace, size_t workspace_bytes,
                        void* stream);
int adell_loco_loss_bwd(const float* f1, const float* f2, const float* gloss, int B, long S, int C,
                        float temperature, float eps, float* df1, float* df2, void* stream);

/* ConvTranspose3d with kernel = stride = 2 on all three axes and 32 / 64 channels on both sides
 * (unet.py:445-458, the upscaling of the high-resolution decoder levels) as streaming GEMMs on the
 * fp32 MFMA: every input voxel feeds exactly its 8 output voxels, so nothing needs a halo or a
 * packed weight. w / dw: torch's canonical [Cin][Cout][2][2][2]. `applicable`: factors 2x2x2,
 * 32 / 64 input and 16 / 32 / 64 output channels, at least 32 768 input voxels (below that the
 * implicit-GEMM entry points above stay in use). x / dy 16-byte aligned.
 * adell_convt_k221_*: the same for factors (2, 2, 1) -- depth and height doubled, width kept
 * (SWIN-UNet's anisotropic upscaling, unetr.py:902-925); w / dw [Cin][Cout][2][2][1], 32 / 64
 * channels on both sides. */
int adell_convt_k2_applicable(int N, int D, int H, int W, int Cin, int Cout);
int adell_convt_k2_fwd(int N, int D, int H, int W, int Cin, int Cout, const float* x,
                       const float* w, const float* bias, float* y, void* stream);
int adell_convt_k2_bwd_data(int N, int D, int H, int W, int Cin, int Cout, const float* dy,
                            const float* w, float* dx, void* stream);
long adell_convt_k2_wgrad_workspace(int N, int D, int H, int W, int Cin, int Cout);
/* db (optional, [Cout]): the bias gradient, a by-product of the same pass over dy */
int adell_convt_k2_bwd_weight(int N, int D, int H, int W, int Cin, int Cout, const float* x,
                              const float* dy, float* dw, float* db, void* workspace,
                              size_t workspace_bytes, void* stream);
int adell_convt_k221_applicable(int N, int D, int H, int W, int Cin, int Cout);
int adell_convt_k221_fwd(int N, int D, int H, int W, int Cin, int Cout, const float* x,
                         const float* w, const float* bias, float* y, void* stream);
int adell_convt_k221_bwd_data(int N, int D, int H, int W, int Cin, int Cout, const float* dy,
                              const float* w, float* dx, void* stream);
long adell_convt_k221_wgrad_workspace(int N, int D, int H, int W, int Cin, int Cout);
int adell_convt_k221_bwd_weight(int N, int D, int H, int W, int Cin, int Cout, const float* x,
                                const float* dy, float* dw, float* db, void* workspace,
                                size_t workspace_bytes, void* stream);

/* 3x3x3 stride-1 convolution with 1..4 input channels and a wide output (the 2 -> 32 conv of the
 * U-Net input block, unet.py:260-273; UNETR's first encoder, unetr.py:225-237) as one small GEMM per
 * brick over K = 27 Cin on the fp32 MFMA (exact fp32 products): forward (+ bias, + the statistics
 * partials [N][adell_conv_cinfold_ntiles][Cout][2] of the fused norm) and weight / bias gradient.
 * x [N][D][H][W][Cin], w / dw canonical [Cout][Cin][3][3][3]. `applicable` tells whether a
 * descriptor takes this path. */
int adell_conv_cinfold_applicable(const adell_conv3d_desc* d);
int adell_conv_cinfold_ntiles(const adell_conv3d_desc* d);
int adell_conv_cinfold_fwd(const adell_conv3d_desc* d, const float* x, const float* w,
                           const float* bias, float* y, float* stat_partials, int partial_rows,
                           void* stream);
/* the same arithmetic on the f16 MFMA with error-compensated operand splits (two input channels;
 * other channel counts run the exact kernel above) */
int adell_conv_cinfold_fwd_f16x3(const adell_conv3d_desc* d, const float* x, const float* w,
                                 const float* bias, float* y, float* stat_partials, int partial_rows,
                           void* stream);
long adell_conv_cinfold_wgrad_workspace(const adell_conv3d_desc* d);
int adell_conv_cinfold_bwd_weight(const adell_conv3d_desc* d, const float* x, const float* dy,
                                  float* dw, float* db, void* workspace, size_t workspace_bytes,
                                  void* stream);
/* the same on the f16 MFMA with error-compensated operand splits (every channel count 1..4; same
 * workspace): bound by the one pass over dy instead of by the fp32 MFMA */
int adell_conv_cinfold_bwd_weight_f16x3(const adell_conv3d_desc* d, const float* x, const float* dy,
                                        float* dw, float* db, void* workspace,
                                        size_t workspace_bytes, void* stream);
/* backward-data of the same convs (dx has 1..4 channels): one GEMM per dY voxel over K = Cout
 * (Cout <= 64, multiple of 4) into the 27 Cin (tap, channel) columns, gathered into dx along a
 * z march. dy 16-byte aligned. */
int adell_conv_cinfold_dx_applicable(const adell_conv3d_desc* d);
int adell_conv_cinfold_bwd_data(const adell_conv3d_desc* d, const float* dy, const float* w,
                                float* dx, void* stream);
/* the same with the per-voxel GEMM on the f16 MFMA with error-compensated operand splits */
int adell_conv_cinfold_bwd_data_f16x3(const adell_conv3d_desc* d, const float* dy, const float* w,
                                      float* dx, void* stream);

/* 1x1x1 convolution with Cout <= 4 (the logits head, unet.py:712-731) on canonical weights
 * w [Cout][C0+C1]: one HBM-bound pass each way. `applicable` tells whether a descriptor takes
 * this path (k = 1, stride 1, no padding, Cout <= 4, Cin <= 512). */
int adell_conv1_small_applicable(const adell_conv3d_desc* d);
int adell_conv1_small_fwd(const adell_conv3d_desc* d, const float* x0, const float* x1,
                          const float* w, const float* bias, float* y, void* stream);
int adell_conv1_small_bwd_data(const adell_conv3d_desc* d, const float* dy, const float* w,
                               float* dx0, float* dx1, void* stream);
long adell_conv1_small_wgrad_workspace(const adell_conv3d_desc* d);
int adell_conv1_small_bwd_weight(const adell_conv3d_desc* d, const float* x0, const float* x1,
                                 const float* dy, float* dw, float* db, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* Convolutions with Cin <= 4 (the 2-channel input block, unet.py:260-273), k = 3 in H and W,
 * 1 or 3 in D, stride 1, one source, canonical weights w [Cout][Cin][KD][3][3]: exact fp32 on
 * the vector ALU instead of an MFMA tile padded to 16 input channels. The forward writes the
 * same (sum, sum of squares) partials as adell_conv3d_fwd, [N][ntiles][Cout][2]. The
 * backward-data produces dX [N][D][H][W][Cin] from dY (Cout a multiple of 4). */
int adell_conv_cin_small_applicable(const adell_conv3d_desc* d);
int adell_conv_cin_small_ntiles(const adell_conv3d_desc* d);
int adell_conv_cin_small_fwd(const adell_conv3d_desc* d, const float* x, const float* w,
                             const float* bias, float* y, float* stat_partials, int partial_rows,
                           void* stream);
int adell_conv_cin_small_bwd_data(const adell_conv3d_desc* d, const float* dy, const float* w,
                                  float* dx, void* stream);

/* dst[off_r + i] = src_r[i] for rows r of a DEVICE table of `rows` triples (source pointer,
 * destination offset in elements, element count <= 16384 per row): gathers the parameter
 * gradients autograd produced into the flat gradient buffer of the fused optimisers. */
int adell_multi_copy(const long* table, int rows, float* dst, void* stream);

/* ---- shifted-window (SWIN) token path: vit.py:33-45,95-129,1005-1256; linear_blocks.py:358-417 */
/* out (contiguous over sizes[0..nd)) = gather of `in`: out dim d adds coord*mult[d] to input
 * axis axis[d]; input axis a has extent / stride (elements) / cyclic shift:
 * in_coord[a] = (sum + shift[a]) mod extent[a]. One kernel for every einops rearrange and
 * torch.roll of the path (window partition + cyclic shift, its inverse, einops_rescale). */
int adell_gather_nd(const float* in, float* out, int nd, const int* sizes, const int* axis,
                    const long* mult, int na, const long* extent, const long* stride,
                    const long* shift, void* stream);
/* LayerNorm over rows of C <= 512 values; input row r starts at (r/inner)*so + (r%inner)*si
 * (contiguous rows: inner = 1, so = C); y, dy, mean, rstd are contiguous; dx is strided by
 * (dso, dsi) the same way, so q / k slices of a QKV buffer are normalised in place. */
int adell_layernorm_rows_fwd(const float* x, long rows, int C, int inner, long so, long si,
                             const float* gamma, const float* beta, float eps, float* y,
                             float* mean, float* rstd, void* stream);
long adell_layernorm_rows_bwd_workspace(long rows, int C);
int adell_layernorm_rows_bwd(const float* x, const float* dy, const float* gamma,
                             const float* mean, const float* rstd, long rows, int C, int inner,
                             long so, long si, float* dx, long dso, long dsi, float* dgamma,
                             float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
/* Attention inside windows of T <= 64 tokens (token row = w*T + i). q, k, out: [tokens][H][A|Dv]
 * contiguous; v (and dv) element (t,h,d) at v[t*v_ts + h*v_hs + d] (inside a QKV buffer).
 * S = scale*q k^T + rel[H][T][T] + mask[w % n_mask][T][T]; O = dropout(softmax(S)) v (Philox
 * mask from (seed, rng_offset), regenerated in the backward); lse [W][H][T] is kept for the
 * backward; ds (optional [W][H][T][T]) receives dS, the gradient of the additive bias. */
int adell_winattn_fwd(const float* q, const float* k, const float* v, long v_ts, long v_hs,
                      const float* rel, const float* mask, int n_mask, long W, int H, int T, int A,
                      int Dv, float scale, float drop_p, unsigned long seed, unsigned rng_offset,
                      float* out, float* lse, void* stream);
int adell_winattn_bwd(const float* q, const float* k, const float* v, long v_ts, long v_hs,
                      const float* rel, const float* mask, int n_mask, const float* o,
                      const float* dout, const float* lse, long W, int H, int T, int A, int Dv,
                      float scale, float drop_p, unsigned long seed, unsigned rng_offset,
                      float* dq, float* dk, float* dv, float* ds, void* stream);

/* Row-major fp32 GEMM (fp32 MFMA) behind torch.nn.Linear (layers/linear_blocks.py,
 * res_blocks.py:559-566, res_net.py:278-324): C[M][N] = A x B (+ bias[N]) (+ residual).
 * a_kc != 0: A(m,k) = A[m*lda + k], else A[k*lda + m]; b_kc != 0: B(k,n) = B[n*ldb + k], else
 * B[k*ldb + n]. Forward X W^T: (1,1); backward-data dY W: (1,0); backward-weight dY^T X: (0,0).
 * workspace: adell_gemm_f32_workspace_floats(M, N, K) floats of split-k slabs (NULL if 0). */
long adell_gemm_f32_workspace_floats(int M, int N, int K);
int adell_gemm_f32(int M, int N, int K, const float* A, long lda, int a_kc, const float* B,
                   long ldb, int b_kc, float* C, long ldc, const float* bias,
                   const float* residual, long ldr, float* workspace, void* stream);

/* The same GEMM on the f16 MFMA by error-compensated splitting (3 MFMAs per product, fp32-class
 * accuracy: csrc/gemm_f16x3.hip) -- torch.nn.Linear forward / dX / dW of the ConvNeXt point-wise
 * MLPs (res_blocks.py:559-566), the ViT / SWIN projections (linear_blocks.py) and the projection
 * heads (res_net.py:278-324). a_absmax / b_absmax: both NULL (operand scales chosen inside the
 * kernel per block and 64-k stage), or device words with the float bits of the absmax of the A / B
 * tensors (adell_absmax_f32 into a zero-initialised word): one scale per tensor. _applicable: 1 when the
 * operands qualify (16-byte alignment; leading dimensions, K and the outer extent of an
 * outer-contiguous operand multiples of 4); otherwise the call returns ADELL_E_UNSUPPORTED and the
 * caller uses adell_gemm_f32. */
int adell_absmax_f32(const float* x, long n, uint32_t* out, void* stream);
int adell_gemm_f16x3_applicable(int M, int N, int K, const float* A, long lda, int a_kc,
                                const float* B, long ldb, int b_kc);
long adell_gemm_f16x3_workspace_floats(int M, int N, int K);
int adell_gemm_f16x3(int M, int N, int K, const float* A, long lda, int a_kc, const float* B,
                     long ldb, int b_kc, float* C, long ldc, const float* bias,
                     const float* residual, long ldr, const uint32_t* a_absmax,
                     const uint32_t* b_absmax, float* workspace, void* stream);
/* The same GEMM with an activation (ADELL_ACT_*) in its epilogue: a Linear -> activation pair
 * without an element-wise pass over the 4 C-wide intermediate (res_blocks.py:559-566: pwconv1 ->
 * GELU -> pwconv2; linear_blocks.py MLP). act_out != NULL: C = A B^T + bias (+ residual) and
 * act_out = act(C), both [M][ldc] (the backward needs the pre-activation, the next layer the
 * activation: one GEMM writes both). dact_in != NULL: C = (A B^T + ...) * act'(dact_in), with
 * dact_in [M][ldc] the saved pre-activation -- the gradient through the activation, applied by the
 * GEMM that produces the gradient of the activation's output. A must be K-contiguous (a_kc = 1). */
int adell_gemm_f16x3_act(int M, int N, int K, const float* A, long lda, int a_kc, const float* B,
                         long ldb, int b_kc, float* C, long ldc, const float* bias,
                         const float* residual, long ldr, const uint32_t* a_absmax,
                         const uint32_t* b_absmax, float* workspace, int act, float act_p,
                         float* act_out, const float* dact_in, void* stream);

/* Element-wise segmentation losses beyond the fused binary dice + focal pair, on probabilities
 * p[B][V][C] (NDHWC; C = 1 for the binary family) against targets of the same layout:
 * kind 0 binary_cross_entropy (losses.py:79-109), 1 cat_cross_entropy (:528-562), 2 mc_focal_loss
 * (:565-607), 3 mc_generalized_dice_loss (:610-653). cw[C] = class weights / alpha (not used by
 * kind 0, which takes w_pos). loss[B]; sums[B][C][2] is kept for the backward, which writes
 * dp = gout[b] * d loss[b] / d p. workspace: adell_seg_loss_workspace(B, V, C) bytes. */
long adell_seg_loss_workspace(int B, long V, int C);
int adell_seg_loss_fwd(int kind, const float* p, const float* t, const float* cw, int B, long V,
                       int C, float eps, float scale, float label_smoothing, float gamma,
                       float smooth, float w_pos, float* loss, float* sums, void* workspace,
                       size_t workspace_bytes, void* stream);
int adell_seg_loss_bwd(int kind, const float* p, const float* t, const float* cw, int B, long V,
                       int C, float eps, float scale, float label_smoothing, float gamma,
                       float smooth, float w_pos, const float* sums, const float* gout, float* dp,
                       void* stream);

/* Softmax over the channel axis of an NDHWC tensor (rows = N * voxels, C <= 32 contiguous
 * class values per row): the n_classes > 2 head, torch.nn.Softmax(dim=1) at unet.py:641-655.
 * backward: dx = y * (dy - sum_c dy * y). */
int adell_channel_softmax_fwd(const float* x, float* y, long rows, int C, void* stream);
int adell_channel_softmax_bwd(const float* y, const float* dy, float* dx, long rows, int C,
                              void* stream);

/* out[n][c] = max over the V voxels of NDHWC x[n][v][c], arg = the first voxel attaining it
 * (torch.max semantics): X.flatten(2).max(-1).values in front of the bottleneck classifier
 * (unet.py:826-828). backward: dx = 0 except dx[n][arg[n][c]][c] = dout[n][c]. */
int adell_channel_max_fwd(const float* x, float* out, int* arg, int N, long V, int C,
                          void* stream);
int adell_channel_max_bwd(const float* dout, const int* arg, float* dx, int N, long V, int C,
                          void* stream);

/* test hook: force one conv tile configuration (0..3), -1 = heuristic */
void adell_debug_force_conv_cfg(int cfg);

/* Launch-plan switches (ten): each selects another BUILT path that the parity tests compare with
 * the default -- "igemm_nospec" (3^3 convs on the generic f16x3 instance), "igemm_no8" (no 8x8x8
 * bricks), "no_splitk", "wgrad_nozring", "wgrad_no16", "igemm_no16" (16-channel layers off their
 * 16-column kernels), "attn_nomfma", "dw_nomfma", "dw_wgrad_nomfma" (depthwise 7^3 on the
 * vector-ALU kernels), "gemm_norows" (Linear layers never on the streaming GEMM). Initialised once
 * at load from the environment variables of the same names (ADELL_ prefix, upper case); the launch
 * path itself never reads the environment. The kernel timing experiments ("igemm_dbg", "zr_dbg":
 * results become WRONG) exist only in -DADELL_DEBUG builds of the library. adell_set_tuning
 * returns ADELL_E_BADARG for an unknown name; adell_get_tuning -1. */
int adell_set_tuning(const char* name, int value);
/* Replay counter of the dropout offsets: a device word (0 after load) that every dropout kernel of
 * the library ADDS to its `rng_offset` argument. Eager callers never touch it. A caller that
 * captures a training step in a HIP graph enqueues adell_rng_advance(K, 0, stream) as the LAST
 * node, K = the number of offsets the step draws: replay r then uses offsets c + r K .. where the
 * capture drew c .., i.e. the masks eager step r would have drawn. set != 0: the word is set to
 * `delta` instead (0 when the caller goes back to eager launches). Stream-ordered. */
int adell_rng_advance(uint32_t delta, int set, void* stream);
int adell_get_tuning(const char* name);

/* ------------------------------------------------------------------------
 * Device-side batch augmentation (csrc/augment.hip): the arithmetic of the MONAI transforms the
 * reference composes in transform_factory/augmentations.py:19-178 (get_augmentations_unet) on
 * volumes resident in HBM. The random draws are the caller's; these are the per-element passes.
 * ---------------------------------------------------------------------- */
/* out[N][4] = (min, max, mean, population std) of each item of per_item contiguous floats
 * (what RandAdjustContrast / RandStdShiftIntensity need of an image) */
long adell_item_stats_workspace(int N, long per_item);
int adell_item_stats(const float* x, int N, long per_item, float* out, void* workspace,
                     size_t workspace_bytes, void* stream);
/* One pass: gamma contrast -> std shift -> Rician noise. params[N][8] per item:
 * {min, max - min, gamma (<= 0: skip), shift, noise std (<= 0: skip), unused x 3}:
 *   v = ((x - min) / (range + 1e-7))^gamma * range + min;  v += shift;
 *   v = sqrt((v + n1)^2 + n2^2), n1, n2 ~ N(0, std) from Philox(seed, rng_offset, element). */
int adell_aug_intensity(const float* x, float* out, int N, long per_item, const float* params,
                        uint64_t seed, uint32_t rng_offset, void* stream);
/* Affine resampling of NDHWC volumes: out[n][v] = sample(x[n], theta[n] (v - centre) + centre) in
 * voxel coordinates (theta[N][12]: rows of a 3 x 4 matrix over (z, y, x)); linear: 1 trilinear,
 * 0 nearest; pad_mode: 0 zeros, 1 border, 2 reflection (about -0.5 / size - 0.5). */
int adell_affine_sample(const float* x, float* out, int N, int D, int H, int W, int C,
                        const float* theta, int linear, int pad_mode, void* stream);
/* Round 4, the rest of get_augmentations_unet (augmentations.py:52-127). All on NDHWC volumes.
 * adell_axis_filter: 1-D filter along spatial axis 0 / 1 / 2 with zero padding, taps[N][2 R + 1]
 *   per item (three calls = the separable Gaussian of RandGaussianSmoothd, "blur").
 * adell_bias_field: out = x * exp(sum c[i][j][k] P_i(z) P_j(y) P_k(x)), Legendre polynomials over
 *   linspace(-1, 1, size), coef[N][64] = dense 4 x 4 x 4 cube (RandBiasFieldd degree 3, "rbf").
 * adell_axis_lut_sample: resampling through per-axis coordinate tables lut[N][D + H + W] (input
 *   voxel coordinate per output index), trilinear / nearest, border padding (RandGridDistortiond,
 *   "distort").
 * adell_gibbs_lowpass: per item the spectrum (three-axis DFT, any axis length <= 1024) is zeroed
 *   outside the sphere of radius[n] about the centre of the SHIFTED spectrum and transformed back
 *   (RandGibbsNoised, the k-space half of "noise"); workspace adell_gibbs_workspace bytes. */
int adell_axis_filter(const float* x, float* out, int N, int D, int H, int W, int C, int axis,
                      const float* taps, int radius, void* stream);
int adell_bias_field(const float* x, float* out, int N, int D, int H, int W, int C,
                     const float* coef, void* stream);
int adell_axis_lut_sample(const float* x, float* out, int N, int D, int H, int W, int C,
                          const float* lut, int linear, void* stream);
/* ------------------------------------------------------------------------
 * Dispatch queries of the depthwise 7^3 kernels (csrc/dw_mfma.hip, dw_dense.hip, dw_wgrad_mfma.hip):
 * 1 when adell_dwconv3d_fwd / _bwd_data / _bwd_weight run this problem on the f16x3 MFMA forms
 * (planes of 9 .. 16 rows and columns, channels in fours, 16-byte aligned tensors, the "dw_nomfma" /
 * "dw_wgrad_nomfma" switches off), resp. on the dense small-volume form (volumes of <= 4^3 voxels).
 * x / y: the two tensors of the launch (alignment is part of the answer). */
int adell_dw_mfma_ok(int N, int C, int D, int H, int W, int KD, int KH, int KW, const float* x,
                     const float* y);
int adell_dw_dense_ok(int N, int C, int D, int H, int W, int KD, int KH, int KW, const float* x,
                      const float* y);
int adell_dw_wgrad_mfma_ok(int N, int C, int D, int H, int W, int KD, int KH, int KW, const float* x,
                           const float* dy);

/* Launch plan of the z-ring weight-gradient kernel (csrc/conv_wgrad_zring.hip) for a conv of these
 * extents: returns 1 and fills plan[8] = {column tiles x, y, z segments, planes per segment, input /
 * output channel tiles, resident blocks, 1 when the 16 x 16 tile form (16-channel layers) runs}, or 0
 * when adell_conv3d_bwd_weight_f16x3 runs the layer on the per-plane kernel instead. */
int adell_wgrad_zring_plan(int N, int D, int H, int W, int C0, int C1, int Cout, int KD, int KH,
                           int KW, int SD, int SH, int SW, int Do, int Ho, int Wo, int* plan);

/* Layer scale folded into a Linear layer's parameters (ConvNeXtBlock3d, res_blocks.py:588-604:
 * gamma * (h W^T + b) = h (gamma W)^T + gamma b): W2[c][k] = gamma[c] W[c][k], b2[c] = gamma[c] b[c]
 * (b / b2 may be NULL), and the backward of that algebra: dW = gamma dW2, db = gamma db2,
 * dgamma[c] = sum_k dW2[c][k] W[c][k] + db2[c] b[c]. W, W2, dW2, dW: [C][K] row-major. */
int adell_rowscale_fwd(const float* gamma, const float* W, const float* b, float* W2, float* b2, int C,
                       int K, void* stream);
int adell_rowscale_bwd(const float* gamma, const float* W, const float* b, const float* dW2,
                       const float* db2, float* dgamma, float* dW, float* db, int C, int K,
                       void* stream);

/* Spatial window of a dense channels-last volume: out[n][d][h][w][:] = in[n][d + od][h + oh][w + ow][:]
 * where that voxel exists, zeros elsewhere (in [N][Di][Hi][Wi][C], out [N][Do][Ho][Wo][C]). Positive
 * offsets and a smaller output: crop_to_size (adell_mri/modules/layers/utils.py:30-52); negative
 * offsets and a larger output: its gradient (a zero frame around dY). */
int adell_window_ndhwc(const float* in, float* out, int N, int C, int Di, int Hi, int Wi, int Do, int Ho,
                       int Wo, int od, int oh, int ow, void* stream);

long adell_gibbs_workspace(int N, int D, int H, int W, int C);
int adell_gibbs_lowpass(const float* x, float* out, int N, int D, int H, int W, int C,
                        const float* radius, void* workspace, size_t workspace_bytes, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* ADELL_HIP_H */
