/* mrdis.h -- C ABI of libmrdis_hip.so: the MI355X (gfx950) hot path of the
 * multi-modal MR representation-disentanglement training step.
 *
 * The reference (ouyangjiahong/representation-disentanglement) is pure
 * Python/PyTorch and has no FFI; its operator seam is the class factory
 * `Conv2d(is_cond)` / `CondConv2d.forward` (src/model.py:2075-2120) and the
 * ATen ops its blocks dispatch.  Each entry point below
 * names the reference call site it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller; nothing is
 *    allocated or freed inside the library, nothing synchronises the host.
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it.
 *  - activations are NHWC fp32 "views": element (n,h,w,c) of a tensor lives at
 *    p[((n*H + h)*W + w)*ld + c]; `ld` (>= C, in floats) lets a view address a
 *    channel slice of a wider buffer (concat elision).
 *  - return value: 0 = ok, negative = MRDIS_E* (see mrdis_strerror).  No
 *    entry point throws or aborts.
 *  - thread-safety: re-entrant for distinct streams / distinct buffers.
 */
#ifndef MRDIS_H
#define MRDIS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRDIS_OK            0
#define MRDIS_EINVAL       -1   /* bad shape / argument                       */
#define MRDIS_EUNSUPPORTED -2   /* geometry outside what the kernels cover    */
#define MRDIS_EWORKSPACE   -3   /* workspace too small                        */
#define MRDIS_ELAUNCH      -4   /* hipLaunch reported an error                */
#define MRDIS_EALIGN       -5   /* pointer / ld alignment requirement broken  */

#define MRDIS_MAX_TAPS 16

/* epilogue flags of mrdis_conv2d_fwd */
#define MRDIS_EPI_NONE   0
#define MRDIS_EPI_LRELU  1      /* y = leaky_relu(conv + bias, 0.2): model.py:2227, 2375-2394, 2774 */

const char* mrdis_strerror(int code);
int mrdis_version(void);

/* Process-wide switches.  Each one is initialised from its environment variable when the library is first used and is
 * changed afterwards only through mrdis_set_option (nothing on the launch path calls getenv):
 *   "wino"  (MRDIS_WINO, default 1): 0 = direct convolution kernels only, 1 = fused Winograd F(2x2,3x3) where it measured
 *           faster (csrc/mrdis_conv.hip wino_wanted), 2 = Winograd wherever the kernel applies (tests);
 *   "nt_mb" (MRDIS_NT_MB, default 128): Winograd outputs of at least this many MB are written with non-temporal stores;
 *   "wino_pipe" (MRDIS_WINO_PIPE, default 1): 1 = the software-pipelined Winograd kernels (csrc/mrdis_wino2.hip: forward / data
 *           gradient for Cout > 32, weight gradient for Ci, Co multiples of 64), 0 = the phase-by-phase kernels everywhere
 *           (same arithmetic; results agree to 1e-5);
 *   "wino_u" (MRDIS_WINO_U, default 1): 1 = the pipelined kernels read the pre-transformed filter image a caller passes (w_wino), 0 = they
 *           always transform the taps themselves (F(2x2): bit-identical either way; also keeps the F(4x4) kernel, which has no in-kernel
 *           filter transform, out);
 *   "wino4" (MRDIS_WINO4, default 1): 1 = Winograd F(4x4,3x3) (csrc/mrdis_wino4.hip, mrdis_wino4r.hip: 1.78x fewer multiplies than F(2x2),
 *           results within ~5e-6 of the maximum of the direct kernel's) for the 3x3 stride-1 filters whose image mrdis_wino_u_format() gives
 *           format 4 or 5, where the caller passes that image and the launch fills >= 3/4 of the chip (>= 192 workgroups, maps >= 16 x 32):
 *             format 4 (36-point image + the 16-point one): reduction channels R % 8 == 0, couts S >= 64, S % 4 == 0, and R >= 64 for a plain
 *                       filter, R >= 32 for the fused gamma | beta filter of mrdis_conv2d_fwd_spade (C % 8 == 0, C >= 32);
 *             format 5 (36-point image of the 32-cout forms, wino4r / wino4n kernels): R % 8 == 0, R >= 16, S == 32;
 *             format 2 (16-point F(2x2) image) otherwise;
 *           the same option gates the F(3x3,4x4) weight gradient (csrc/mrdis_wino4w.hip) inside mrdis_conv2d_bwd_weight (Ci % 32 == 0, Co % 64 == 0, H and W
 *           multiples of 8, >= 192 workgroups of >= 24 iterations each); 0 = never (F(2x2) everywhere, every image format 2); 2 = wherever those kernels apply (tests: R >= 16,
 *           format 5 from 4 couts).  The option picks the format when an image is BUILT; a built image keeps its format, which the caller
 *           passes beside the pointer (w_wino_fmt) -- changing the option later never makes a kernel read an image in another layout;
 *   "debug_now16" (MRDIS_DEBUG_NOW16, default 0): 1 = the dedicated narrow-layer kernels off -- Cout <= 16 weight gradient
 *           (mrdis_wgrad16.hip), stride-2 first layers and 4 -> C weight gradient (mrdis_wgrad_s2.hip), 1x1 head (mrdis_pointwise.hip),
 *           16-cout and 4-cout 3x3 layers (mrdis_c16.hip, mrdis_co4.hip): the generic tile kernels run those layers (tests and
 *           tools/ use it for A/B; results agree to fp32 rounding);
 *   "debug_nopack" (MRDIS_DEBUG_NOPACK, default 0): 1 = the four output-parity classes of a stride-2 data gradient as four launches
 *           instead of one (bit-identical results; A/B switch);
 *   other "debug_*": kernel-selection overrides used by tools/ (see csrc/mrdis_elem.hip OPT_DEFS).
 * set: 0 or MRDIS_EINVAL (unknown name); get: the value, or MRDIS_EINVAL for an unknown name.  Not synchronised with launches
 * in flight on other threads. */
int mrdis_set_option(const char* name, long long value);
long long mrdis_get_option(const char* name);
/* Launches since load / the last reset of one kernel family (counted on the host at launch): "wino" | "wino_spade" (phase-by-phase F(2x2)),
 * "wino2" | "wino2_spade" (pipelined F(2x2)), "wino4" | "wino4_spade" (F(4x4) 64-cout forms), "wino4n" | "wino4r" (32-cout forms: shared transform / register-fed),
 * "wino_wgrad" | "wino_wgrad2" (F(2x2) weight gradient), "wino4_wgrad" (F(3x3,4x4)); "bconv3" | "bconv3_spade" | "bconv4" | "bconv4_spade" (bf16 3x3 forms);
 * "split6_c4" | "split6_c16" | "split6_wgrad16" | "split6_co4" | "split6_c3d" | "split6_w3d" | "split6_tap" (option split6: the 4 -> C kernel, the 32 -> 16 forward, its weight gradient, the
 * C -> 4 kernel, the 3-D 16 -> 16 forward / data-gradient and weight-gradient kernels, the tap-table kernel fed by mrdis_s6_filter_image, as six bf16 products per fp32 product); "all" (every kernel launch of the library).  MRDIS_EINVAL for an unknown name.  Diagnostics: the parity tests
 * use it to prove that the form under test is the one that ran. */
long long mrdis_launch_count(const char* family);
void mrdis_launch_count_reset(void);
/* The largest DYNAMIC LDS size each kernel was launched with so far, as "kernel expression=bytes" lines (at most cap - 1 characters, whole lines only); returns the number of
 * lines written.  rocprofv3's kernel trace only shows the static group segment (0 for the kernels that size their LDS at launch). */
int mrdis_dynamic_lds_table(char* buf, int cap);

/* ---- expert mixing: model.py:2111-2113 --------------------------------------
 * W   : (E, Co, Ci, kh, kw) checkpoint layout (OIHW with leading expert dim)
 * r   : (E) routing weights sigmoid(fc(type)) -- one row; the reference's
 *       inputs_type is constant over the batch at every call site
 *       (model.py:3138, 3169, 3190, 3211), so one mixed kernel serves a call.
 * w_tck : out, [T][Ci][Co]   (T = kh*kw)  forward / wgrad layout
 * w_tkc : out, [T][Co][Ci]                data-gradient layout
 * E = 1 with r = {1} turns a plain nn.Conv2d weight into the two layouts.    */
int mrdis_mix_experts_fwd(const float* W, const float* r, float* w_tck, float* w_tkc,
                          int E, int Co, int Ci, int T, void* stream);

/* backward of the above.  dw_tck: [T][Ci][Co] gradient of the mixed kernel.
 * dW (E,Co,Ci,T) = r[e] * dWm ;  dr[e] += <dWm, W[e]>  (two-level ordered
 * reduction through `workspace`, bit-reproducible; caller zeroes dr).        */
size_t mrdis_mix_experts_bwd_workspace(int E, int Co, int Ci, int T);
int mrdis_mix_experts_bwd(const float* dw_tck, const float* W, const float* r,
                          float* dW, float* dr, void* workspace, size_t workspace_bytes,
                          int E, int Co, int Ci, int T, void* stream);

/* Routed forms: the routing r = sigmoid(fc_w @ type_row + fc_b) of `_routing.forward` (model.py:2071-2073) is
 * evaluated inside the mix kernel (r_out (E) is kept for the backward), and the backward returns the
 * gradients of the routing Linear directly: dfc_w (E,emb), dfc_b (E).  type_row: (emb) device floats. */
int mrdis_mix_experts_routed_fwd(const float* W, const float* fc_w, const float* fc_b, const float* type_row, int emb,
                                 float* r_out, float* w_tck, float* w_tkc, int E, int Co, int Ci, int T, void* stream);
int mrdis_mix_experts_routed_bwd(const float* dw_tck, const float* W, const float* r, const float* type_row, int emb,
                                 float* dW, float* dfc_w, float* dfc_b, void* workspace, size_t workspace_bytes,
                                 int E, int Co, int Ci, int T, void* stream);

/* All M type rows of a layer at once (types (M, emb); a step calls every CondConv2d with each modality
 * label, model.py:3138).  w_tck / w_tkc / dw_tck are HOST arrays of M device pointers; a NULL dw_tck[m]
 * means "type m received no gradient".  r_out (M, E).  dW, dfc_w, dfc_b are the sums over the types.     */
int mrdis_mix_experts_routed_multi_fwd(const float* W, const float* fc_w, const float* fc_b, const float* types, int emb, int M,
                                       float* r_out, float* const* w_tck, float* const* w_tkc,
                                       void* const* w_bf16_tck, void* const* w_bf16_tkc,   /* both NULL, or M bf16 buffers each: bf16 copies of
                                                                                              the two layouts (what mrdis_cast_bf16 would give) */
                                       int ld_tck, long long tap_tkc,   /* 0 = dense.  Otherwise the row pitch of the [T][Ci][.] outputs and the tap
                                                                           pitch of the [T][.][Ci] outputs: the pointers address a column / row block of a
                                                                           wider filter (the halves of a fused gamma | beta filter, model.py:2443-2444) */
                                       int E, int Co, int Ci, int T, void* stream);
size_t mrdis_mix_experts_routed_multi_bwd_workspace(int M, int E, int Co, int Ci, int T);
int mrdis_mix_experts_routed_multi_bwd(const float* const* dw_tck, const float* W, const float* r, const float* types,
                                       int emb, int M, float* dW, float* dfc_w, float* dfc_b,
                                       int accumulate,      /* 1: the three results are ADDED to dW / dfc_w / dfc_b (gradient sinks) */
                                       int ld_dw,           /* 0 = dense; otherwise the row pitch of the dw_tck tensors (a column block of a wider gradient) */
                                       void* workspace, size_t workspace_bytes, int E, int Co, int Ci, int T, void* stream);


/* Every CondConv2d layer of the model in one launch (forward) and one launch pair (backward): the per-layer launches above over a
 * table of jobs in device memory.  A job = one layer, or one half of a fused gamma | beta filter pair; its fields (pointers to the
 * expert weights, routing parameters, the per-label outputs, the gradient sinks, its block range) are laid out as `MixJob` in
 * csrc/mrdis_conv.hip (mrdis_mix_job_bytes() = its size, 368; the Python binding mirrors it with ctypes).  Block-to-element mapping and
 * summation order are those of the per-layer launches (mrdis_mix_job_blocks blocks per job): bit-identical results.
 * dw_table: [njobs][8] device pointers to this step's filter gradients [T][Ci][ld_dw], null where a label's filter got none.
 * Replaces ~90 x routing.forward + sum(r * W) (model.py:2071-2073, 2113) per step and their backward. */
size_t mrdis_mix_job_bytes(void);
int mrdis_mix_job_blocks(int Co, int Ci, int T);
int mrdis_mix_jobs_fwd(const void* jobs, int njobs, int total_blocks, const float* types, int emb, int M, void* stream);
int mrdis_mix_jobs_bwd(const void* jobs, int njobs, int total_blocks, const void* dw_table, const float* types, int emb, int M, void* stream);

/* ---- convolution: F.conv2d at model.py:2104 (CondConv2d._conv_forward) and
 * nn.Conv2d of the discriminator model.py:2773-2789 ---------------------------
 * x : NHWC view (N,H,W,Ci) ld=ldx ; y : NHWC view (N,Ho,Wo,Co) ld=ldy
 * w_tck from mrdis_mix_experts_fwd ; bias (Co) or NULL.
 * kernel (kh,kw) in {1x1,3x3,4x4}; stride in {1,2}; zero padding `pad`.
 *
 * dtype (BASELINE.json configs[2], SURVEY.md 8b "dtype"):
 *   MRDIS_DT_F32       fp32 activations, fp32 MFMA (v_mfma_f32_32x32x2_f32 / Winograd) -- the parity path;
 *   MRDIS_DT_F32_BF16M fp32 activations in HBM, bf16 MFMA operands (v_mfma_f32_32x32x16_bf16, inputs rounded RNE on the
 *                      way into LDS), fp32 accumulate / bias / epilogue.  Needs the bf16 copy of the filter with the
 *                      REDUCTION axis contiguous (mrdis_cast_bf16 of the other layout): forward w_bf16_tkc = bf16
 *                      [T][Co][Ci], data gradient w_bf16_tck = bf16 [T][Ci][Co].  Geometries the bf16 kernel does not
 *                      cover (reduction axis not a multiple of 16, fewer than 16 output channels) and a NULL bf16 filter
 *                      run the fp32 kernels: the result is then exact fp32;
 *   MRDIS_DT_BF16      bf16 activations in HBM (x, y, dy, dx are bf16 views, ld in bf16 elements), bf16 MFMA operands, fp32
 *                      accumulate; bias, filters' master copy, weight / bias gradients and all statistics stay fp32.
 *                      Convolutions: the geometries of the bf16 kernels only (reduction axis % 16 == 0, at least 16 output
 *                      channels, % 4) -- others return MRDIS_EUNSUPPORTED and the caller casts the view (mrdis_cast_view)
 *                      around the fp32 kernel.  The norm / resize / activation entry points take either storage type
 *                      through their own `dtype` argument (arithmetic is fp32 inside).                               */
#define MRDIS_DT_F32        0
#define MRDIS_DT_F32_BF16M  1
#define MRDIS_DT_BF16       2
/* mixed storage at the two ends of a bf16 stretch (`compute_dtype: bf16` keeps tensors with < 16 channels in fp32): named after the
 * LAYER's input x (= dx) and output y (= dy), the same code goes to its forward, data-gradient and weight-gradient entry points.
 *   MRDIS_DT_XBF16_YF32  x / dx bf16 views, y / dy fp32: the 1x1 decoder head 16 -> <= 8 (forward, data gradient, weight gradient); the 3x3
 *                        C -> 4 layer (ana_dec.output): forward (x bf16, C = 32 / 64, y (N, H, W, 4) fp32; w_tck is then [9][C][16], columns >= 4
 *                        zero), data gradient (dy fp32 -> dx bf16; w_tkc is then [9][16][Ci], rows >= 4 zero) and weight gradient (x bf16 with
 *                        C = 32 / 64, dy (N, H, W, 4) fp32 -> dw_tck (9, C, 4) + the 4 bias sums; maps 64 / 128 / 256 wide, else MRDIS_EUNSUPPORTED)
 *   MRDIS_DT_XF32_YBF16  x / dx fp32, y / dy bf16: the 3x3 4 -> C si_layers -- forward (w_tck is then the [9][16][Co] layout, rows >= 4 zero),
 *                        weight gradient (dw_tck (9, 4, Co); maps 64 / 128 / 256 wide, Co 32 / 64 / 128, else MRDIS_EUNSUPPORTED) and data
 *                        gradient (dy bf16, Co = 32 / 64 -> dx (N, H, W, 4) fp32; w_tkc is then [9][Co][16], columns >= 4 zero)
 *                        (the 16-row / 16-column filter layouts are those the mixing launch writes for narrow layers under bf16 storage)
 * These kernels multiply on the bf16 matrix pipe with the fp32 side carried as two (bf16-output forms) or three (fp32-output forms: exact
 * products) bf16 terms; MRDIS_EUNSUPPORTED outside the shapes named.
 * Under MRDIS_DT_BF16 mrdis_conv2d_bwd_weight also takes the stride-2 layers (3x3 / 4x4, pad 1, even H and W, Ci 16 or a multiple of 32, Co % 8 == 0):
 * four input-parity classes in one launch.                                                                                          */
#define MRDIS_DT_XBF16_YF32 3
#define MRDIS_DT_XF32_YBF16 4
/* mrdis_conv2d_bwd_weight only, OR-ed onto one of the two mixed-storage types: dw_tck has the STORED shape of a filter that the mixing launch keeps
 * zero-padded to 16 -- [T][16][Co] for a 4 -> C layer (MRDIS_DT_XF32_YBF16), [T][Ci][16] for a C -> 4 layer (MRDIS_DT_XBF16_YF32) -- and the rows /
 * columns beyond the layer's own are written as zeros (no pad of the result afterwards).  MRDIS_EUNSUPPORTED where the four-channel kernels decline. */
#define MRDIS_DT_DW_PAD16   0x100
/* w_wino (fp32 paths, may be NULL): the filter already in the Winograd domain, the image mrdis_wino_u_jobs builds from w_tck (role:
 * forward) -- used where a software-pipelined Winograd kernel takes the layer, ignored elsewhere.  w_wino_fmt: the format that image was
 * BUILT in (the job's fmt: 2 | 4 | 5; ignored when w_wino is NULL); MRDIS_EINVAL if no image of this filter shape can have it.  Format 2
 * (F(2x2,3x3)): same results bit for bit with or without the image; formats 4 / 5: select the F(4x4,3x3) kernels, which exist only on the
 * image (with the option "wino4" at 0 a format-4 image is read through its trailing 16-point part, a format-5 image is ignored). */
int mrdis_conv2d_fwd(const void* x, int ldx, const float* w_tck, const void* w_bf16_tkc, const float* bias,
                     void* y, int ldy, int N, int H, int W, int Ci, int Co,
                     int kh, int kw, int stride, int pad, int epilogue, int dtype, const float* w_wino, int w_wino_fmt, void* stream);

/* ---- filter images for the software-pipelined Winograd kernels (csrc/mrdis_wino2.hip).  U = G g G^T of a 3x3 filter does not depend
 * on the activations; a training step uses each mixed filter 8-16 times, so the transform is taken out of the convolution kernels:
 * one launch over a job table builds, per (filter, role), the 16-point image in the order the kernel's (input-channel chunk, 64-cout
 * tile) walk consumes it: [cout tile][chunk of 8][8][4][64][4] floats, zero-padded (format 2); format 4 is the 36-point image of the
 * F(4x4,3x3) kernel, [cout tile][chunk of 4][18 point pairs][4][128] floats (csrc/mrdis_wino4.h), FOLLOWED by the format-2 image of the same filter
 * (what a call runs on whose grid the F(4x4) kernel declines: small maps); format 5 is the 36-point image of the 32-cout F(4x4) forms alone.
 * mrdis_wino_u_format(R, S, spadeC) says which one a filter gets NOW (a function of the filter's shape and the current value of the option
 * "wino4"); the job records it, mrdis_wino_u_image_floats_fmt sizes an image of a given format (-1: that shape never has it), and the
 * convolution entry points take it beside the image pointer (w_wino_fmt).
 * Job (`WinoUJob`, mrdis_wino_u_job_bytes() = 48):
 *   { const float* w; float* img; int R, S, flip, spadeC, block0, nblk, fmt, pad; }        fmt = mrdis_wino_u_format(R, S, spadeC) when built
 * w = [9][R][S]: role forward: w_tck, R = Ci, S = Co, flip = 0; role data gradient: w_tkc, R = Co, S = Ci, flip = 1; role SPADE (the fused
 * gamma | beta filter of mrdis_conv2d_fwd_spade): w_tck, R = Ci, S = 2 C, spadeC = C.  img: mrdis_wino_u_image_floats(R, S, spadeC) floats,
 * 16-byte aligned.  block0 / nblk: the job's block range (mrdis_wino_u_job_blocks each), total_blocks = their sum.                     */
size_t mrdis_wino_u_job_bytes(void);
int mrdis_wino_u_format(int R, int S, int spadeC);
long long mrdis_wino_u_image_floats(int R, int S, int spadeC);                 /* = _fmt(..., mrdis_wino_u_format(R, S, spadeC)) */
long long mrdis_wino_u_image_floats_fmt(int R, int S, int spadeC, int fmt);
int mrdis_wino_u_job_blocks(int R, int S, int spadeC);
int mrdis_wino_u_jobs(const void* jobs, int njobs, int total_blocks, void* stream);

/* Six-product filter image (option split6; mrdis_s6conv.hip): the layers without a Winograd form -- the 4x4 / 3x3 stride-2 convolutions of the encoders,
 * F.conv2d at model.py:2104 -- multiply on the bf16 matrix pipe with both fp32 operands split into three bf16 terms (the six products of order <= 2,
 * fp32 accumulation: fp32-equivalent results).  The filter is split ONCE into this image; the convolution entry points take it through their w_wino
 * argument with w_wino_fmt = 6 (fp32 views only; without it, or where the geometry does not fit, the fp32 MFMA kernels run).
 * w = [taps][Cred][Cout] fp32: role forward: w_tck (Cred = Ci, Cout = Co); role data gradient: w_tkc (Cred = Co, Cout = Ci).
 * image: mrdis_s6_filter_image_bytes(taps, Cred, Cout) bytes, 16-byte aligned (0: Cred is not a multiple of 8).                                  */
size_t mrdis_s6_filter_image_bytes(int taps, int Cred, int Cout);
int mrdis_s6_filter_image(const float* w, int taps, int Cred, int Cout, void* image, size_t image_bytes, void* stream);

/* data gradient (autograd convolution_backward, input part).
 * dy view (N,Ho,Wo,Co) ld=lddy -> dx view (N,H,W,Ci) ld=lddx ; w_tkc layout. */
int mrdis_conv2d_bwd_data(const void* dy, int lddy, const float* w_tkc, const void* w_bf16_tck,
                          void* dx, int lddx, int N, int H, int W, int Ci, int Co,
                          int kh, int kw, int stride, int pad, int dtype, const float* w_wino /* role data gradient, or NULL */, int w_wino_fmt, void* stream);

/* fp32 -> bf16, round to nearest even (the bf16 filter copies above); src 16-byte, dst 8-byte aligned. */
int mrdis_cast_bf16(const float* src, void* dst_bf16, long long n, void* stream);
/* NHWC view cast (P rows; ld in ELEMENTS of the respective type) between the storage types, optionally changing the channel
 * count: the first min(C_src, C_dst) channels are copied, the rest of a wider destination is zero.  The boundary between bf16
 * activations and the fp32-only tensors: e.g. the 4-channel anatomy map -> a 16-channel bf16 view the bf16 MFMA kernels accept,
 * and a 16-channel bf16 head output -> its first 7 channels in fp32.  Any combination of MRDIS_DT_F32 / MRDIS_DT_BF16.          */
int mrdis_cast_view(const void* src, int ld_src, int src_dtype, int C_src, void* dst, int ld_dst, int dst_dtype, int C_dst,
                    long long P, void* stream);

/* Word-wise copy src -> dst by a kernel (nbytes % 4 == 0, 4-byte aligned).  src may be pinned host memory: the way small
 * host -> device transfers inside a step avoid the copy engine's host round trip (the reference's CPU-drawn eps, model.py:3159-3162). */
int mrdis_copy_bytes(const void* src, void* dst, long long nbytes, void* stream);
/* Measurement probe: fills n_floats (multiple of 4, 16-byte aligned) with `value` by non-temporal 16-byte stores and reads nothing: the rate a store-only kernel
 * reaches on this device (bench.py reports it beside the north-star convolution, whose traffic is 89 % stores). */
int mrdis_stream_fill(float* dst, long long n_floats, float value, void* stream);

/* A SPADE block's  InstanceNorm(z) * (1 + gamma(s)) + beta(s)  (model.py:2440-2446) with the gamma | beta convolution and the modulation in
 * ONE launch: x = the si_layers output (N, H, W, Ci), w_tck = the fused [9][Ci][2 C] filter (gamma couts first), bias (2 C), z (N, H, W, C)
 * and its instance statistics (mrdis_instnorm_stats).  Writes mix and gamma (the backward, mrdis_instnorm_spade_bwd, needs gamma).
 * dtype MRDIS_DT_F32 (x, z, mix, gamma fp32; w_tck) or MRDIS_DT_BF16 (bf16 views; w_bf16_tkc = the bf16 [9][2 C][Ci] filter).
 * MRDIS_EUNSUPPORTED where the pipelined kernels are not the kernels of choice: run mrdis_conv2d_fwd + mrdis_instnorm_spade_fwd instead. */
int mrdis_instnorm_stats(const void* z, int ldz, float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                         int N, long long HW, int C, float eps, int dtype, void* stream);
int mrdis_conv2d_fwd_spade(const void* x, int ldx, const float* w_tck, const void* w_bf16_tkc, const float* bias, const void* z, int ldz,
                           const float* mean, const float* rstd, void* mix, int ldmix, void* gamma, int ldg,
                           int N, int H, int W, int Ci, int C, int dtype, const float* w_wino /* role SPADE, or NULL */, int w_wino_fmt /* 2 | 4 */, void* stream);

/* weight gradient.  Two-pass, bit-reproducible: partial slabs in `workspace`
 * (size from mrdis_conv2d_bwd_weight_workspace) then an ordered reduction.
 * dw_tck: [T][Ci][Co] ; dbias (Co) or NULL.                                   */
size_t mrdis_conv2d_bwd_weight_workspace(int N, int H, int W, int Ci, int Co,
                                         int kh, int kw, int stride, int pad);
/* accumulate_bias != 0: dbias += column sums of dy (instead of =), e.g. straight into the parameter's gradient.
 * dtype MRDIS_DT_F32_BF16M: x and dy are rounded to bf16 on their way into LDS and multiplied on bf16 MFMA with fp32
 * accumulation (stride-1 "same" layers with Ci % 32 == 0; others run the fp32 kernels); dbias stays an fp32 sum.       */
int mrdis_conv2d_bwd_weight(const void* x, int ldx, const void* dy, int lddy,
                            float* dw_tck, float* dbias, void* workspace, size_t workspace_bytes,
                            int N, int H, int W, int Ci, int Co,
                            int kh, int kw, int stride, int pad, int accumulate_bias, int dtype, void* stream);

/* ---- LeakyReLU backward (model.py:2227/2240, 2375-2394): dx = dy * (y>0 ? 1 : slope),
 * y being the activation OUTPUT (sign-preserving for slope > 0).              */
int mrdis_lrelu_bwd(const void* dy, int lddy, const void* y, int ldy, void* dx, int lddx,
                    long long P, int C, float slope, int dtype, void* stream);

/* ---- BatchNorm2d, training mode: model.py:2132/2151, 2179/2191, 2776-2785 --
 * x view (P = N*H*W rows, C) ; writes y view, save_mean/save_rstd (C) and
 * updates running_mean/var (momentum 0.1, unbiased var) when non-NULL.
 * groups = G > 1: the view holds G batches of P rows that the reference normalises in G separate calls of the SAME layer (the
 * modalities of one encoder pass, model.py:3135-3157): statistics per group (save_mean / save_rstd, and dgamma / dbeta of the
 * backward, hold G * C entries), the running statistics are updated group by group in order, rounded to fp32 in between.
 * workspace: groups * mrdis_norm_workspace(1, P, C) bytes (each group is chunked as a call of its own: bit-identical statistics). */
size_t mrdis_norm_workspace(int groups, long long P, int C);
int mrdis_bn_train_fwd(const void* x, int ldx, void* y, int ldy, const float* gamma,
                       const float* beta, float* running_mean, float* running_var,
                       float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                       long long P, int C, float eps, float momentum, int groups, int dtype, void* stream);
/* inference mode (model.eval(), main_missing.py:338): y = (x - running_mean) * rsqrt(running_var + eps) * gamma + beta */
int mrdis_bn_eval_fwd(const void* x, int ldx, void* y, int ldy, const float* gamma, const float* beta,
                      const float* running_mean, const float* running_var, long long P, int C, float eps, int dtype, void* stream);
/* acc_dgamma / acc_dbeta (both or neither): running parameter-gradient sums this call's dgamma / dbeta are
 * added to in the same launch (a module called several times per step needs no separate accumulation).   */
int mrdis_bn_train_bwd(const void* dy, int lddy, const void* x, int ldx, const float* gamma,
                       const float* save_mean, const float* save_rstd, void* dx, int lddx,
                       float* dgamma, float* dbeta, float* acc_dgamma, float* acc_dbeta,
                       void* workspace, size_t workspace_bytes, long long P, int C, int groups, int dtype, void* stream);

/* ---- InstanceNorm2d(affine=False) fused with the SPADE modulation:
 * model.py:2431/2440 + 2446:  out = IN(z) * (1 + gamma) + beta ---------------
 * z, gamma, beta, out : (N, HW, C) views ; save_mean/save_rstd : (N*C), written -- or, with workspace = NULL, READ: the
 * statistics of z are then taken from them (mrdis_bilinear_up2_stats_fwd computed them when z was produced).            */
int mrdis_instnorm_spade_fwd(const void* z, int ldz, const void* gamma, int ldg,
                             const void* beta, int ldb, void* out, int ldo,
                             float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                             int N, long long HW, int C, float eps, int dtype, void* stream);
size_t mrdis_instnorm_spade_bwd_workspace(int N, long long HW, int C);
int mrdis_instnorm_spade_bwd(const void* dout, int lddo, const void* z, int ldz,
                             const void* gamma, int ldg, const float* save_mean, const float* save_rstd,
                             void* dz, int lddz, void* dgamma, int lddg, void* dbeta, int lddb,
                             void* workspace, size_t workspace_bytes,
                             int N, long long HW, int C, int dtype, void* stream);
/* The same backward when z = nn.Upsample(scale_factor=2, bilinear)(x) (model.py:2551-2573 in front of a SPADE block): z, dout, gamma, dgamma, dbeta
 * are (N, 2 Hi, 2 Wi, C); dx (N, Hi, Wi, C) receives the gradient of x -- the apply pass and the resize's adjoint in one kernel, the full-resolution
 * d z is neither written nor read back.  xlo != NULL: x itself (N, Hi, Wi, C); z may then be NULL -- both passes interpolate z from x exactly as
 * mrdis_bilinear_fwd stored it, so the up-sampled map need not be kept for the backward.  Workspace: ALWAYS size it with
 * mrdis_instnorm_spade_bwd_up2_workspace (>= mrdis_instnorm_spade_bwd_workspace(N, 4 Hi Wi, C), which is all the two-pass form needs: with the
 * smaller size the entry point takes the two-pass form).  MRDIS_EUNSUPPORTED: views that are not 16-byte aligned / C % 4 != 0 / a last 32-channel chunk
 * whose width is not 4, 8, 16 or 32 (the caller runs mrdis_instnorm_spade_bwd + mrdis_bilinear_bwd).
 * With xlo and a workspace of mrdis_instnorm_spade_bwd_up2_workspace(N, Hi, Wi, C, dtype) bytes: ONE pass over the full-resolution tensors -- d z is linear
 * in (dzh, 1, zh) and so is the resize's adjoint, so the kernel writes U^T dzh and the partial sums, and a kernel over the LOW-resolution map finishes
 * d x = rstd (U^T dzh - 4 s0 / HW - (s1 / HW) rstd (U^T U x - 4 mean)): dout and gamma are read once, z never (bf16 maps: U^T dzh travels in fp32
 * through the workspace). */
size_t mrdis_instnorm_spade_bwd_up2_workspace(int N, int Hi, int Wi, int C, int dtype);
int mrdis_instnorm_spade_bwd_up2(const void* dout, int lddo, const void* z, int ldz,
                                 const void* gamma, int ldg, const float* save_mean, const float* save_rstd,
                                 void* dx, int lddx, void* dgamma, int lddg, void* dbeta, int lddb,
                                 void* workspace, size_t workspace_bytes,
                                 int N, int Hi, int Wi, int C, const void* xlo, int ldxlo, int dtype, void* stream);

/* ---- bilinear resize: nn.Upsample at model.py:2175 (align_corners=True),
 * 2432 / 2501-2509 (align_corners=False, arbitrary output size) --------------*/
int mrdis_bilinear_fwd(const void* x, int ldx, void* y, int ldy, int N, int Hi, int Wi,
                       int Ho, int Wo, int C, int align_corners, int dtype, void* stream);
/* nn.Upsample(scale_factor=(2,2), bilinear) between two SPADE blocks (model.py:2551-2573, 2622-2627) together with the InstanceNorm
 * statistics of its result (model.py:2440): y (N, 2 Hi, 2 Wi, C) as mrdis_bilinear_fwd(align_corners = 0) writes it, save_mean /
 * save_rstd (N*C) as mrdis_instnorm_stats would compute them from y -- taken from the values while they are stored, so the 4x tensor
 * is not read back for a statistics pass.  C % 4 == 0, 16-byte aligned views; MRDIS_EUNSUPPORTED otherwise.  The consumer passes the
 * statistics on: mrdis_conv2d_fwd_spade takes them as arguments, mrdis_instnorm_spade_fwd with workspace = NULL uses the ones in
 * save_mean / save_rstd instead of computing them.                                                                                */
size_t mrdis_bilinear_up2_stats_workspace(int N, int Hi, int C);
/* 1 where mrdis_bilinear_up2_stats_fwd takes (N, Wi, C) on dense aligned views, 0 where it would return MRDIS_EUNSUPPORTED (callers choose their path beforehand) */
int mrdis_bilinear_up2_stats_applies(int N, int Wi, int C);
int mrdis_bilinear_up2_stats_fwd(const void* x, int ldx, void* y, int ldy, int N, int Hi, int Wi, int C,
                                 int out_block, long long out_block_stride,
                                 float* save_mean, float* save_rstd, float eps, void* workspace, size_t workspace_bytes,
                                 int dtype, void* stream);
/* out_block = 0 (or >= N): y is the dense (N, 2 Hi, 2 Wi, ldy) view.  0 < out_block < N (N % out_block == 0): the N images leave as
 * N / out_block blocks of out_block images, block k at y + k * out_block_stride elements (dense inside a block) -- the shared SPADE decoder
 * writes its result for modality label j straight into the [decoder i][label j] arrangement the per-modality decoders read (model.py:3200-3224),
 * which was a concatenation copy of 268 MB per decoder call.                                                                            */
int mrdis_bilinear_bwd(const void* dy, int lddy, void* dx, int lddx, int N, int Hi, int Wi,
                       int Ho, int Wo, int C, int align_corners, int dtype, void* stream);

/* ---- softmax over [100*mask_img, s] with channel 0 dropped: model.py:3150-3153
 * s view (P, C) ; mask_img (P) ; out view (P, C).                             */
int mrdis_softmax_mask_drop_fwd(const float* s, int lds, const float* mask_img, float* out, int ldo,
                                long long P, int C, float mask_scale, void* stream);
int mrdis_softmax_mask_drop_bwd(const float* dout, int lddo, const float* out, int ldo,
                                float* ds, int ldds, long long P, int C, void* stream);

/* ---- per-sample mean |gt - x| (p=1) or (gt-x)^2 (p=2) over (C,H,W):
 * model.py:3260-3266.  out (N).  bwd: dx = w[n] * d|.|/dx / (C*HW).           */
size_t mrdis_recon_err_workspace(int N, long long HW, int C);
int mrdis_recon_err_fwd(const float* gt, int ldgt, const float* x, int ldx, float* out,
                        void* workspace, size_t workspace_bytes,
                        int N, long long HW, int C, int p, void* stream);
int mrdis_recon_err_bwd(const float* gt, int ldgt, const float* x, int ldx, const float* w,
                        float* dx, int lddx, int N, long long HW, int C, int p, void* stream);

/* ---- evaluate() reconstruction metrics on the device: util.py:935-978
 * (compute_reconstruction_metrics[_single], called at main_missing.py:528).
 * n_img images of H x W: image i, pixel p at target[(i*H*W + p) * ldt] (i.e. channel 0
 * of an NHWC view).  Both images are shifted by their own minimum, data range = max of
 * the shifted target.  out (n_img, 3) = { MSE (the reference's 'rmse' key), PSNR, SSIM }.
 * SSIM follows skimage.metrics.structural_similarity defaults (7x7 uniform window,
 * sample covariance, K1 0.01, K2 0.03, mean over the window-valid region).            */
size_t mrdis_recon_metrics_workspace(int n_img, int H);
int mrdis_recon_metrics(const float* target, int ldt, const float* pred, int ldp, float* out,
                        void* workspace, size_t workspace_bytes, int n_img, int H, int W, void* stream);

/* ---- batch assembly from HBM-resident volumes: ZeroDoseDataset.__getitem__ + default collate
 * (util.py:471-566).  vol_ptrs (B, M): device pointers (as 64-bit integers, 0 = contrast missing for that subject)
 * to volumes stored as [D][H][W] planes; slice_idx (B): centre slice, already clamped to [block, D-1-block];
 * drop (B): contrast zeroed by the drop-off augmentation or -1.  All three live in device memory.
 * inputs: NHWC view (B,H,W,ld_in), channel m*(2*block+1)+k = slice slice_idx-block+k of contrast m;
 * mask (B, M); mask_img (B,H,W) = (inputs[:,0] == 0).                                              */
int mrdis_slice_gather(const void* vol_ptrs, const int* slice_idx, const int* drop, float* inputs, int ld_in,
                       float* mask, float* mask_img, int B, int M, int H, int W, int D, int block, void* stream);

/* ---- max_pool2d(kernel k x k, stride k): model.py:3448-3451 ---------------- */
int mrdis_maxpool_fwd(const float* x, int ldx, float* y, int32_t* argmax, int N, int H, int W, int C,
                      int k, void* stream);
int mrdis_maxpool_bwd(const float* dy, const int32_t* argmax, float* dx, int lddx, int N, int H, int W,
                      int C, int k, void* stream);

/* ---- optimizer side: main_missing.py:272-278 (clip + finite check) and :118/:283
 * (Adam, amsgrad, L2 weight decay) over one flat fp32 parameter arena --------
 * sumsq_finite: out[0] += sum(g^2), out[1] += count(non-finite)  (zero `out` first). */
size_t mrdis_sumsq_workspace(void);
int mrdis_sumsq_finite(const float* g, long long n, float* out, void* workspace, size_t workspace_bytes,
                       void* stream);
/* One fused step.  norm_finite points at the device scalar pair written by mrdis_sumsq_finite (or NULL): with
 * max_norm > 0 the gradient is scaled by coef = min(1, max_norm / (sqrt(sumsq) * grad_scale + 1e-6)) (clip_grad_norm_,
 * main_missing.py:272); the step is skipped on the device when the non-finite count is > 0 (the reference stops in pdb
 * there, :273-278), with max_norm = 0 the pair only gates.
 * step_count / step_state: bias correction uses the 1-based host `step_count`, or -- when step_state (device float[2]) is
 * given -- a device-side counter: [0] = steps applied (incremented by this call unless it is skipped), [1] = steps skipped
 * as non-finite; a skipped step does not advance the bias correction.
 * gates (n_gates <= 32): HOST arrays gate_ranges[2k], [2k+1] = index range [lo, hi) of the arena, gate_flag_index[k] = entry
 * of the DEVICE array gate_flags that gates it: a range whose flag is 0 is left untouched (torch's Adam skips parameters
 * whose grad is None: a decoder whose modality is absent from the whole batch).
 * gate_steps (DEVICE float[3 * n_flags], zero-initialised, or NULL; needs step_state): per-flag step counters.  torch's Adam
 * keeps `step` per parameter and does not advance it while the gradient is None, so the ranges of flag k are bias-corrected
 * with [k] = the number of applied steps in which flag k was set (incremented by this call); [n_flags ..) is scratch
 * for the (1 - beta1^t, sqrt(1 - beta2^t)) pairs.  NULL: gated ranges use the arena-wide counter.                  */
int mrdis_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax,
                            long long n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int step_count, float* step_state, const float* norm_finite,
                            float max_norm, float grad_scale, const long long* gate_ranges, const int* gate_flag_index,
                            int n_gates, const float* gate_flags, float* gate_steps, int n_flags, void* stream);

/* ==== 3-D path: the Conv3d / GroupNorm / Upsample layers of BasicBlock, UNet3D, VAEBranch and
 * NVNet3D (src/model.py:1856-2060).  Tensors are NDHWC fp32 views (torch.channels_last_3d); 1x1x1 convolutions go
 * through mrdis_conv2d_* on the (N*D, H, W) view. -------------------------------------------------------------- */

/* nn.Conv3d(k = 3, padding = 1, stride in {1,2}) (model.py:1861, 1864, 1969-1984): x (N,D,H,W,Ci) ld=ldx ->
 * y (N,Do,Ho,Wo,Co) ld=ldy.  w_tck [27][Ci][Co] with tap = (kd*3 + kh)*3 + kw (mrdis_mix_experts_fwd, E = 1, T = 27).
 * residual (same extents as y, ld=ldres) or NULL: y = conv + bias + residual -- BasicBlock's `x + residul`
 * (model.py:1873) in the epilogue.                                                                             */
int mrdis_conv3d_fwd(const float* x, int ldx, const float* w_tck, const float* bias, const float* residual, int ldres,
                     float* y, int ldy, int N, int D, int H, int W, int Ci, int Co,
                     int k, int stride, int pad, void* stream);
/* data gradient: dy (N,Do,Ho,Wo,Co) -> dx (N,D,H,W,Ci); w_tkc [27][Co][Ci] */
int mrdis_conv3d_bwd_data(const float* dy, int lddy, const float* w_tkc, float* dx, int lddx,
                          int N, int D, int H, int W, int Ci, int Co, int k, int stride, int pad, void* stream);
/* weight (+ bias) gradient: split-K slabs in `workspace`, ordered reduction; dw_tck [27][Ci][Co], dbias (Co) or NULL */
size_t mrdis_conv3d_bwd_weight_workspace(int N, int D, int H, int W, int Ci, int Co, int k, int stride, int pad);
int mrdis_conv3d_bwd_weight(const float* x, int ldx, const float* dy, int lddy, float* dw_tck, float* dbias,
                            void* workspace, size_t workspace_bytes,
                            int N, int D, int H, int W, int Ci, int Co, int k, int stride, int pad, void* stream);

/* nn.GroupNorm(G, C) followed by nn.ReLU (model.py:1859-1860, 1862-1863, 1889-1890) on (N, P = D*H*W, C) rows:
 * y = relu?((x - mean[n,g]) * rstd[n,g] * gamma + beta); biased variance, eps inside the sqrt.  save_mean /
 * save_rstd: (N, G).  The backward takes dy w.r.t. the ReLU output and recomputes the ReLU mask from x.          */
size_t mrdis_groupnorm_workspace(int N, long long P, int C, int G);
int mrdis_groupnorm_relu_fwd(const float* x, int ldx, float* y, int ldy, const float* gamma, const float* beta,
                             float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                             int N, long long P, int C, int G, float eps, int relu, void* stream);
int mrdis_groupnorm_relu_bwd(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta,
                             const float* save_mean, const float* save_rstd, float* dx, int lddx,
                             float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                             int N, long long P, int C, int G, int relu, void* stream);
/* the same with a second gradient of x summed in: dx = (GroupNorm+ReLU backward) + add (add: (N, P, C) rows, ld = ldadd; NULL = none).  BasicBlock's input
 * x feeds the normalised branch AND the residual addition (model.py:1873): both gradients of x then leave in one pass instead of autograd's extra add.   */
int mrdis_groupnorm_relu_bwd_add(const float* dy, int lddy, const float* x, int ldx, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_rstd, float* dx, int lddx,
                                 float* dgamma, float* dbeta, const float* add, int ldadd, void* workspace, size_t workspace_bytes,
                                 int N, long long P, int C, int G, int relu, void* stream);

/* nn.Upsample(scale_factor=2) (nearest, model.py:1995-2003, 1898-1911) fused with the skip addition of
 * UNet3D.forward (model.py:2029-2040): y (N,2D,2H,2W,C) = x[d/2,h/2,w/2] + skip (skip may be NULL); contiguous NDHWC.
 * Backward: dx = sum of the 8 children of dy (d skip = dy needs no kernel).                                       */
int mrdis_upsample2x_add_fwd(const float* x, const float* skip, float* y, int N, int D, int H, int W, int C, void* stream);
int mrdis_upsample2x_bwd(const float* dy, float* dx, int N, int D, int H, int W, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRDIS_H */
