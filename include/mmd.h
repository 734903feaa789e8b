/* libmmd - C ABI of the MI355X-native MM-Diffusion denoising hot path.
 *
 * One shared library (mm-diffusion_amd/lib/libmmd.so, built by hipcc --offload-arch=gfx950) loaded with
 * ctypes by the Python host mirror (mm-diffusion_amd/mm_diffusion/_hip.py).  The reference has no native
 * layer (pure PyTorch), so each entry point replaces a GROUP of ATen calls of the reference hot path; the
 * reference interface it replaces is cited per function (paths relative to /root/reference/mm_diffusion).
 *
 * Conventions
 *   - raw DEVICE pointers + sizes; row-major "channels-last" activations X[rows, C] with a row stride `ld`
 *     in ELEMENTS (so a column slice of a wider buffer is a valid tensor: skip concats are free views);
 *     video rows are ordered (n, f, h, w), audio rows (n, l)
 *   - dtype: MMD_F32 (0) or MMD_BF16 (1) = element type of activations / GEMM weights; all statistics,
 *     accumulation, softmax and bias arithmetic are fp32
 *   - every call only ENQUEUES work on `stream` (a hipStream_t); no allocation, no sync, no global state:
 *     the caller owns all buffers including workspaces; calls are re-entrant and graph-capturable
 *   - return 0 on success, <0 on error (MMD_ERR_*); mmd_last_error() gives the message (thread local)
 */
#ifndef MMD_H
#define MMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MMD_F32 0
#define MMD_BF16 1
#define MMD_OK 0
#define MMD_ERR_ARG (-1)
#define MMD_ERR_LAUNCH (-2)
#define MMD_ERR_UNSUPPORTED (-3)

int mmd_version(void);
const char* mmd_last_error(void);
/* test-suite aid: print the native backtrace of a fatal signal (SIGSEGV / SIGBUS / SIGABRT) to stderr, then chain to the handler
 * installed before (Python's faulthandler).  Never called by the product path. */
int mmd_debug_install_crash_handler(void);

/* --- HIP graph capture of one denoising step + stream-ordered timing (bench roofline leg) --- */
int mmd_graph_begin(void* stream);
int mmd_graph_end(void* stream, void** exec_out);
int mmd_graph_launch(void* exec, void* stream);
int mmd_graph_destroy(void* exec);
/* private launch streams of the host mirror (hipStreamNonBlocking): the video / audio chains and the capture stream */
int mmd_stream_create(void** stream_out);
int mmd_stream_sync(void* stream);
int mmd_stream_destroy(void* stream);
int mmd_event_create(void** ev);
int mmd_event_record(void* ev, void* stream);
int mmd_stream_wait_event(void* stream, void* ev);   /* fork/join of the video and audio launch streams (also under capture) */
int mmd_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);
int mmd_event_destroy(void* ev);

/* timestep_embedding + time_embed MLP (nn.py:192-210; multimodal_unet.py:791-795,1075).
 * t_kind: 0 int64, 1 int32, 2 float32.  out_silu[N,dim] = SiLU(time_embed(emb(t))) (the input every
 * ResBlock emb_layers applies its Linear to, unet:366-372); out_raw (nullable) = time_embed output. */
int mmd_temb_fwd(const void* t, int t_kind, int N, int dim, const float* W0, const float* b0, const float* W2,
                 const float* b2, float* out_silu, float* out_raw, void* stream);

/* y[N,J] = x[N,K] W[J,K]^T + b : all ResBlock emb_layers Linear(128 -> 2*Cout) batched into one call
 * over the row-concatenated weights (unet:366-372,454). */
int mmd_linear_fwd(const float* x, const float* W, const float* b, float* y, int N, int K, int J, void* stream);

/* GroupNorm32 statistics -> fused per-(slice,channel) affine (nn.py:16-33; FiLM unet:457-470).
 * Slice s normalises rows base(s) + j*tstride (j < Tn), base(s) = (s/inner)*outer_stride + (s%inner)*inner_stride.
 * a_out/b_out [S, C] fp32: y = x*a + b; mr_out (nullable) [S, 32, 2] = (mean, rstd) for the backward.  film (nullable) [S, >=2C] rows (scale | shift), row stride film_ld.
 * workspace: mmd_gn_workspace_bytes(dtype, C, S, Tn) bytes (may be 0: slice handled by one block). */
int64_t mmd_gn_workspace_bytes(int dtype, int C, int S, int Tn);
int mmd_gn_stats(int dtype, const void* x, int64_t ld, int C, int S, int Tn, int inner, int64_t outer_stride,
                 int64_t inner_stride, int64_t tstride, const float* gamma, const float* beta, const float* film,
                 int64_t film_ld, float eps, float* a_out, float* b_out, float* mr_out, void* workspace, void* stream);
/* The same fused affine from PRODUCER-side statistics: the GEMM that wrote x (mmd_conv_gemm_stats / mmd_gn_conv1x1_stats) left per
 * (64-row record, QUAD of 4 channels) float2 (sum, sum of squares) in rec[(row / 64) * rec_ld + quad]; rec points at the first quad of
 * the C normalised channels (C % 128 == 0: groups of whole quads); S contiguous slices of Tn rows, Tn % 64 == 0.
 * Replaces the statistics pass over x of GroupNorm32 (nn.py:16-33) for conv-fed norms. */
int mmd_gn_finalize_stats(const float* rec, int64_t rec_ld, int C, int S, int Tn, const float* gamma, const float* beta,
                          const float* film, int64_t film_ld, float eps, float* a_out, float* b_out, float* mr_out, void* stream);
/* y = act(x*a[slice(row)] + b[slice(row)]), act: 0 none, 1 SiLU (nn.SiLU after every GroupNorm32). */
int mmd_gn_apply(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int C, int S, int Tn, int inner,
                 int64_t outer_stride, int64_t inner_stride, int64_t tstride, const float* a, const float* b, int act,
                 void* stream);
/* x[m, :] += e[m / rows_per_sample, :]  (non-FiLM ResBlock h + emb_out, unet:473-477). */
int mmd_add_rowbias(int dtype, void* x, int64_t ld, int64_t rows, int C, int64_t rows_per_sample, const float* e,
                    int64_t e_ld, void* stream);
/* Its gradient w.r.t. e: out[s, c] += sum over the rows of sample s of dY[row, c] (S contiguous slices of Tn rows; out fp32 [S, ldo],
 * accumulated - the caller zeroes). */
int mmd_colsum_slices(int dtype, const void* dY, int64_t lddy, int S, int64_t Tn, int C, float* out, int64_t ldo, void* stream);

/* Implicit-GEMM convolution on the matrix cores:
 *   Y[m, co] = bias[co] + sum_tap sum_ci A[src(m,tap), ci] * W[co, tap*Cin + ci] (+ R[m, co])
 * rows m decompose as (n, p0, p1, p2) over (D0, D1, D2); tap t = taps[3t..3t+2] offsets (p0,p1,p2), out of
 * range -> zero padding.  Covers VideoConv 2d+1d spatial (D=(1,H,W), 9 taps) and temporal (D=(F,HW,1), 3 taps),
 * VideoConv '3d' k=1, AudioConv k=3 dilated (D=(L,1,1), taps (+-d,0,0)) and k=1, and every qkv/proj 1x1 conv
 * (unet:83-131,272,275,378,401,605-610).  W is [Cout][ntaps*Cin] in `dtype`; bias fp32 (nullable);
 * R (nullable) residual in `dtype`.  taps is a HOST pointer.  tile: 0 auto, 64 or 128 (register-staged
 * main loop), 129 (128x128 tile, direct-to-LDS global_load_lds main loop), 132 (129 with a four-slot LDS ring - three K steps of
 * DMA in flight, one block per CU - for launches with fewer tiles than the chip has block slots; Cin a multiple of 128 bytes),
 * 131 (bf16 convs with ntaps * Cin in {128, 256, 384, 512}: row strips stationary in registers, weights streamed through
 * LDS); these variants are bitwise identical.  130 = halo-tile
 * main loop for spatial 3x3 convs (chunk-major K order: equal to rounding); 133 = the same on 16 x 16 pixel patches (bf16, 8 waves,
 * one block per CU, three-slot weight ring; D1 % 16 == 0, D2 % 16 == 0), bitwise equal to 130. */
int mmd_conv_gemm(int dtype, const void* A, int64_t lda, const void* W, const float* bias, const void* R, int64_t ldr,
                  void* Y, int64_t ldy, int M, int Cout, int Cin, int ntaps, const int* taps, int D0, int D1, int D2, int tile,
                  void* stream);

/* 1x1 conv whose input GroupNorm32(+FiLM)(+SiLU) is applied on the way into LDS (no normalised tensor in HBM):
 *   Y = act(A * gn_a[s(m)] + gn_b[s(m)]) W^T + bias (+ R),  s(m) = m / rows_per_slice  (S contiguous slices, S*rows == M,
 *   rows_per_slice >= 128, Cin <= 256).  gn_a / gn_b [S, Cin] come from mmd_gn_stats.  Replaces the ResBlock tail
 *   norm -> SiLU -> out conv -> + skip (unet:373-388,457-483).  tile 131 (bf16): the normalisation is applied once per row
 *   strip in registers, so wide outputs (the qkv convs of the attention blocks, unet:272,605-606) fuse too; it needs
 *   rows_per_slice >= 256 (Cin 128 / 256) or >= 128 (Cin 384). */
int mmd_gn_conv1x1(int dtype, const void* A, int64_t lda, const float* gn_a, const float* gn_b, int act, int S,
                   int64_t rows_per_slice, const void* W, const float* bias, const void* R, int64_t ldr, void* Y, int64_t ldy,
                   int M, int Cout, int Cin, int tile, void* stream);

/* GroupNorm32(+SiLU) of S SHORT slices (Tn <= 16 rows; geometry as mmd_gn_stats) in one launch and one read of the tensor: the
 * temporal-attention norm over the frames of a pixel (unet:489-490 -> nn.py:16-33) = mmd_gn_stats + mmd_gn_apply.  Exact two-pass
 * statistics on register-resident data; channel groups must be whole quads (C % 128 == 0). */
int mmd_gn_small(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int C, int S, int Tn, int inner, int64_t outer_stride,
                 int64_t inner_stride, int64_t tstride, const float* gamma, const float* beta, float eps, int act, void* stream);

/* GroupNorm32(+FiLM)(+SiLU) of slices of a few hundred rows in ONE launch (nn.py:16-33; FiLM unet:457-470): one block per (group, slice),
 * the group's Tn x C / 32 elements register-resident (Tn * C / 128 <= 4096), exact two-pass fp32 statistics.  Writes the fused affine
 * a_out / b_out [S, C] (nullable pair: as mmd_gn_stats), the normalised tensor y (nullable: as mmd_gn_apply; act: 0 none, 1 SiLU), or
 * both = mmd_gn_stats (two launches on multi-block slices) + mmd_gn_apply for the slices whose rows are no multiple of the 64-row
 * producer records (the 400-row audio samples at ds8).  Geometry, film, mr_out as mmd_gn_stats; C % 128 == 0. */
int mmd_gn_group(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int C, int S, int Tn, int inner, int64_t outer_stride,
                 int64_t inner_stride, int64_t tstride, const float* gamma, const float* beta, const float* film, int64_t film_ld,
                 float eps, int act, float* a_out, float* b_out, float* mr_out, void* stream);

/* hipMemsetAsync(ptr, 0, bytes) on the stream (a memset node of a captured plan). */
int mmd_zero(void* ptr, int64_t bytes, void* stream);

/* The two GEMMs above with the GroupNorm statistics of their OUTPUT produced in the epilogue, for the norm that consumes Y next
 * (ResBlock in_layers / out_layers norms, attention norms, the heads: unet:339-340,374-375; nn.py:16-33): per (64-row record,
 * QUAD of 4 consecutive columns) the sum and the sum of squares of the values as stored, stats[(m / 64) * stats_ld + quad] =
 * float2 (per column until round 3: every group size of the model is a multiple of 4 and the per-quad fold costs the producers a
 * third of the instructions).  `stats` points at the first quad this launch writes (producers of a channel-concatenated tensor
 * fill column slices of one record buffer); Cout % 4 == 0.
 * M % 64 == 0; not with tile 130; tile 131 emits them for every K it accepts (K = 128 / 256: a wave owns whole 64-row records;
 * K = 384 / 512: two waves' half-records are paired through LDS).  The order in which a record's 64 rows are folded belongs
 * to the kernel family (tiles 128 / 129 share one, 131 has its own): a layer must be given the same family at every batch size
 * if its results are to be batch-invariant to the last bit.  mmd_gn_finalize_stats consumes the records. */
int mmd_conv_gemm_stats(int dtype, const void* A, int64_t lda, const void* W, const float* bias, const void* R, int64_t ldr,
                        void* Y, int64_t ldy, int M, int Cout, int Cin, int ntaps, const int* taps, int D0, int D1, int D2, int tile,
                        float* stats, int64_t stats_ld, void* stream);
int mmd_gn_conv1x1_stats(int dtype, const void* A, int64_t lda, const float* gn_a, const float* gn_b, int act, int S,
                         int64_t rows_per_slice, const void* W, const float* bias, const void* R, int64_t ldr, void* Y, int64_t ldy,
                         int M, int Cout, int Cin, int tile, float* stats, int64_t stats_ld, void* stream);

/* Spatial 3x3 conv whose input GroupNorm32(+FiLM)(+SiLU) is applied to the staged halo tile in LDS (tile 130, bf16, the nine
 * (0, dh, dw) taps, slices of whole frames): Y = conv3x3(act(A * gn_a[s(m)] + gn_b[s(m)])) + bias (+ R), zero padding of the
 * NORMALISED activation.  Replaces GroupNorm32 -> SiLU -> video_conv_spatial of the ResBlock in_layers (unet:339-340,83-99,
 * 457-458; nn.py:16-33) in one launch; bitwise equal to mmd_gn_apply followed by mmd_conv_gemm tile 130.  tile = 130 or 133. */
int mmd_gn_conv_gemm(int dtype, const void* A, int64_t lda, const float* gn_a, const float* gn_b, int act, int S,
                     int64_t rows_per_slice, const void* W, const float* bias, const void* R, int64_t ldr, void* Y, int64_t ldy,
                     int M, int Cout, int Cin, int ntaps, const int* taps, int D0, int D1, int D2, int tile, void* stream);

/* VideoConv '2d+1d' in ONE launch (round 4): Y = conv1d_t(conv2d_s(act(X * gn_a[s] + gn_b[s]))) - the per-frame 3x3 conv, then the
 * per-pixel k=3 conv along the 16 frames (unet:83-99), with the ResBlock in_layers GroupNorm32 + SiLU applied to the staged input
 * (unet:339-340,457-458; nn.py:16-33; gn_a == NULL: plain conv) and, optionally, the statistics of Y for the out_layers norm
 * (quad records as mmd_conv_gemm_stats writes them; 4 records per block: record (n * H * W / 16 + patch) * 4 + frame group, i.e.
 * records are only meaningful to norms whose slices are whole samples).  bf16; X rows [N * 16 * H * W, Cin] (row stride ldx), Y rows
 * [.., 128]; F == 16, Cout == 128, Cin % 32 == 0, H % 4 == 0, W % 4 == 0; norm slices = whole samples.  The spatial result is rounded
 * to bf16 with its bias before the temporal conv (what a two-launch path stores); zero padding pads the NORMALISED activation.
 * Wf: the weight image of mmd_vconv2d1d_pack (mmd_vconv2d1d_weight_bytes(Cin) bytes) built from the packed spatial [128][9 * Cin]
 * and temporal [128][3 * 128] matrices (K index = tap * C + ci). */
int64_t mmd_vconv2d1d_weight_bytes(int Cin);
int mmd_vconv2d1d_pack(const void* Ws, const void* Wt, void* out, int Cin, int Cout, void* stream);
int mmd_vconv2d1d(const void* X, int64_t ldx, const float* gn_a, const float* gn_b, int act, int S, int64_t rows_per_slice,
                  const void* Wf, const float* bias_s, const float* bias_t, void* Y, int64_t ldy, int N, int F, int H, int W,
                  int Cin, int Cout, float* stats, int64_t stats_ld, void* stream);

/* The temporal-attention block of the video stream in ONE launch (round 4): Y = X + proj_out(attention over the 16 frames of each
 * pixel of qkv(GroupNorm32(X))) - SingleModalAtten with temporal sequences (unet:246-287 as used at :485-493: norm = nn.py:16-33 over
 * (16 frames, C / 32 channels) of a pixel, qkv / proj_out 1x1 convs, QKVAttention unet:290-330 with 1 / sqrt(ch) scaling).  Replaces
 * mmd_gn_small + mmd_conv_gemm (qkv) + mmd_attn_small_fwd + mmd_conv_gemm (proj_out, residual): the normalised tensor, the qkv tensor
 * and the attention output never exist in HBM (q, k, v and the attention output are rounded to bf16 exactly where the unfused path
 * stores them).  bf16; X / Y rows (n, f, pixel) x C with row strides ldx / ldy, Y != X; built for F == 16, heads == 4, C == 256
 * (head width 64: the ds2 level; the 384 / 512-channel instances of round 4 were removed in round 5), HW % 8 == 0.  Wf: the image of mmd_tattn_pack (mmd_tattn_weight_bytes(C, with_pre) bytes) built
 * from the qkv weight [3 C, C] (rows q | k | v, head h = rows h ch .. of each third) and the proj_out weight [C, C], both bf16
 * row-major.  bias_qkv [3 C], bias_proj / gamma / beta [C] fp32.  stats (nullable): quad statistics records of Y for the GroupNorm
 * that consumes it, one per 64 rows in THIS kernel's row order inside a sample (record n HW / 4 + (pixel >> 2): the 16 frames of 4
 * consecutive pixels), so only norms over whole samples may finalize from them.
 * Optional front stage (A != NULL, Wf packed with Wpre [C, C]): the block's input is x = X + A Wpre^T + bias_pre, i.e. the proj_out
 * 1x1 conv + residual of the SPATIAL attention block that precedes the temporal one (unet:485-490) rides in the same launch; MID
 * [rows, C] (distinct from X, A, Y) receives x, rounded to bf16 like the tensor the unfused path stores, and is re-read as operand
 * and as the residual of the last stage.  A == NULL: Wpre == NULL at pack time, MID / bias_pre unused. */
int64_t mmd_tattn_weight_bytes(int C, int with_pre);
int mmd_tattn_pack(const void* Wpre, const void* Wqkv, const void* Wproj, void* out, int C, void* stream);
int mmd_tattn_block(const void* X, int64_t ldx, const void* A, int64_t lda, void* MID, int64_t ldm, const void* Wf,
                    const float* bias_pre, const float* bias_qkv, const float* bias_proj, const float* gamma, const float* beta,
                    float eps, void* Y, int64_t ldy, int N, int F, int HW, int C, int heads, float* stats, int64_t stats_ld,
                    void* stream);

/* The temporal half of VideoConv '2d+1d' (unet:83-99: the per-pixel k = 3 conv along the frames, after the per-frame 3x3 conv) with
 * the activations stationary in registers (round 4): rows are gathered per pixel, the 16 frames of a pixel are the 16 lanes of a DPP
 * row, and the operand of tap df is the x fragment shifted by df lanes (zeros shifted in = the conv's zero padding in time), so every
 * activation row is loaded once instead of once per tap.  Bitwise equal to mmd_conv_gemm with the temporal taps.  bf16; X / Y rows
 * (n, f, pixel) x Cin / Cout (Y != X); F == 16, Cin in {256, 384, 512}, Cout % 64 == 0 (<= 512), HW % 8 == 0.  Wf: the image of
 * mmd_tconv_pack (mmd_tconv_weight_bytes(Cin, Cout) bytes) built from the packed GEMM matrix [Cout][3 * Cin] (K = tap * Cin + ci, taps
 * df = -1, 0, +1) that mmd_conv_gemm takes.  stats (nullable): quad statistics records of Y, one per 64 rows in THIS kernel's row
 * order inside a sample (record n HW / 4 + (pixel >> 2): the 16 frames of 4 consecutive pixels) - for norms over whole samples. */
int64_t mmd_tconv_weight_bytes(int Cin, int Cout);
int mmd_tconv_pack(const void* W, void* out, int Cin, int Cout, void* stream);
int mmd_tconv(const void* X, int64_t ldx, const void* Wf, const float* bias, void* Y, int64_t ldy, int N, int F, int HW, int Cin,
              int Cout, float* stats, int64_t stats_ld, void* stream);

/* The audio half of a ResBlock's in_layers in one launch (round 6): GroupNorm32(+FiLM) + SiLU + AudioConv (Conv1d, kernel 3, dilation,
 * zero "same" padding) on channels-last rows of S = M / L samples (unet:339-346, 108-131; nn.py:16-33) - instead of mmd_gn_apply followed
 * by mmd_conv_gemm with the taps (-d, 0, +d).  bf16.  X [M, ldx] (Cin columns), W [Cout][3 Cin] (mmd_conv_gemm's weight layout),
 * ga / gb [S, Cin] = the fused affine written by mmd_gn_finalize_stats / mmd_gn_stats, act != 0: SiLU; a tap that leaves its sample
 * contributes zero (the padding is zero AFTER the norm).  stats (nullable): quad records of Y as mmd_conv_gemm_stats writes them
 * (M % 64 == 0).  Bitwise equal to the two launches it replaces (same K order, same rounding points). */
int mmd_aconv(const void* X, int64_t ldx, const void* W, const float* bias, const float* ga, const float* gb, int act, void* Y, int64_t ldy,
              int M, int L, int Cin, int Cout, int dil, float* stats, int64_t stats_ld, void* stream);

/* softmax(q k^T / sqrt(ch)) v over query groups with circular key windows - SingleModalQKVAttention
 * (unet:221-240) and the random-shift cross-modal QKVAttention (unet:507-564; window addressing unet:614-647).
 * For batch n, group g (< G): queries = Q rows n*q_rows_per_batch + g*q_per_group + [0, q_per_group) (the last
 * group extends to q_rows_per_batch); keys = KV rows n*k_rows_per_batch + ((g + shift)*k_per_group + j) mod
 * k_rows_per_batch, j < win*k_per_group.  shift_dev: device int (nullable = 0) so a captured graph can be
 * replayed with a new shift.  Head h uses columns {q,k,v}_off + h*ch.  impl: 0 auto, 1 force the VALU kernel, 2 the per-128-query
 * MFMA kernel (register-staged K / V), 3 the staged-window kernel, 4 the DMA-staged kernel (head width 64: K / V tiles by
 * buffer_load ... lds, V^T fragments by transposing LDS reads; bitwise equal to 2), 5 the software-pipelined kernel whose loop
 * iteration is one hand-written instruction stream (head width 64; bitwise equal to 2). */
int mmd_attn_fwd(int dtype, const void* Q, int64_t ldq, int q_off, const void* KV, int64_t ldkv, int k_off, int v_off, void* O,
                 int64_t ldo, int heads, int ch, int nb, int G, int64_t q_rows_per_batch, int q_per_group,
                 int64_t k_rows_per_batch, int k_per_group, int win, const int* shift_dev, int impl, void* stream);
/* Self-attention over short strided slices (temporal attention '(b h w) c f', unet:489-490): rows hold
 * [q(C)|k(C)|v(C)]; slice geometry as mmd_gn_stats; Tn <= 32. */
int mmd_attn_small_fwd(int dtype, const void* QKV, int64_t ld, void* O, int64_t ldo, int C, int heads, int S, int Tn, int inner,
                       int64_t outer_stride, int64_t inner_stride, int64_t tstride, void* stream);

/* Downsample (avg-pool, mode 0) / Upsample (nearest, mode 1) by (1, fh, fw) on rows (nf, h, w)
 * (unet:133-208: video (1,2,2); audio F=1,H=1,W=L, fw=4).  H, W describe the INPUT. */
int mmd_resample(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int C, int NF, int H, int W, int fh, int fw,
                 int mode, float scale, void* stream);   /* y = scale * resample(x): backward of one mode is the other mode, rescaled */
/* The same resampling (bf16, scale 1) with the GroupNorm statistics of the OUTPUT in the epilogue: stats / stats_ld as in
 * mmd_conv_gemm_stats - one (sum, sum of squares) record per 64 output rows and QUAD of channels, of the values as stored.  The norms
 * that follow a resample (the out_layers norm of a down ResBlock, every norm over an up ResBlock's output: unet:441-448 + nn.py:16-33)
 * then finalize with mmd_gn_finalize_stats instead of a statistics pass.  Output rows % 64 == 0, C % 8 == 0, C <= 2048. */
int mmd_resample_stats(const void* x, int64_t ldx, void* y, int64_t ldy, int C, int NF, int H, int W, int fh, int fw, int mode,
                       float* stats, int64_t stats_ld, void* stream);
/* 2-D strided copy (skip-connection concat th.cat, unet:1093-1094). */
int mmd_copy2d(const void* x, int64_t ldx_bytes, void* y, int64_t ldy_bytes, int64_t rows, int64_t row_bytes, void* stream);

/* Stem: API layout fp32 x[N,F,Cin,H,W] (audio F=1,H=1,W=L) -> channels-last rows; W fp32 [ntaps][Cin][Cout]
 * (InitialBlock, unet:680-694: spatial 3x3 half of the 2d+1d conv, and the audio k=3 conv). */
int mmd_stem_conv(int dtype, const float* x, const float* w, const float* bias, void* y, int64_t ldy, int N, int F, int Cin,
                  int H, int W, int Cout, int ntaps, const int* taps, void* stream);
/* Head: channels-last rows (after GN+SiLU) -> API layout fp32 y[N,F,Co,H,W]; W fp32 [ntaps][Cin][Co], Co <= 8
 * (video_out Conv3d 3x3x3 / audio_out Conv1d k3, unet:1003-1012). */
int mmd_head_conv(int dtype, const void* x, int64_t ldx, const float* w, const float* bias, float* y, int N, int F, int Cin,
                  int H, int W, int Co, int ntaps, const int* taps, void* stream);
/* The same head for few output channels as GEMM + gather (round 5; bf16, Cin = 128, ntaps * Co <= 96): mmd_head_gemm computes the
 * per-row products P[tap Co + co][m] = sum_ci W[tap][ci][co] act(x[m][ci] a + b) on the matrix cores - the GroupNorm32 + SiLU of
 * video_out.0/.1 (unet:1003-1006) applied in registers from the fused affine a / b [S, Cin] (S slices of gn_rows rows, gn_rows % 128
 * == 0), x read once, nothing normalised written - and mmd_head_gather sums them over the taps into the API layout
 * y[N, F, Co, H, W] (+ bias; zero padding outside (F, H, W)).  wimg: the bf16 (hi, lo) weight image in MFMA fragment order,
 * mmd_head_gemm_weight_bytes(Cin) bytes, packed by the host mirror (ops.head_gemm_pack); P: fp32 workspace [ntaps * Co][M]. */
int64_t mmd_head_gemm_weight_bytes(int Cin);
int64_t mmd_head_gemm_workspace_bytes(int64_t M, int ntaps, int Co);
int mmd_head_gemm(const void* x, int64_t ldx, int64_t M, int Cin, const float* gn_a, const float* gn_b, int S, int64_t gn_rows, int act,
                  const void* wimg, float* P, int NO, void* stream);
int mmd_head_gather(const float* P, const float* bias, float* y, int N, int F, int H, int W, int Co, int ntaps, const int* taps,
                    void* stream);

/* One DDPM ancestral step for one stream (p_mean_variance + p_sample, multimodal_gaussian_diffusion.py:231-343,
 * 415-474) on API-layout fp32 tensors x/noise/out [N,F,C,HW], model_out [N,F,Cm,HW] (Cm = 2C with flag 4).
 * tables fp32 [7][T]: sqrt_recip_ac, sqrt_recipm1_ac, post_c1, post_c2, fixed logvar, min_log, max_log;
 * t int64[N] device.  flags: 1 clip x0, 2 model predicts x0, 4 learned-range variance.  out (sample = mean +
 * [t!=0] exp(logvar/2) noise), x0_out, mean_out, logvar_out are each nullable (p_mean_variance alone: out = NULL). */
int mmd_ddpm_update(const float* x, const float* model_out, const float* noise, float* out, float* x0_out, float* mean_out,
                    float* logvar_out, const float* tables, const int64_t* t, int T, int N, int F, int C, int HW, int flags,
                    void* stream);
/* q_sample (gd:187-205): out = tab2[0][t] x0 + tab2[1][t] eps. */
int mmd_q_sample(const float* x0, const float* eps, float* out, const float* tab2, const int64_t* t, int T, int N,
                 int64_t per_sample, void* stream);

/* Per-sample loss terms of multimodal_training_losses for one stream (gd:1114-1203; vb term _vb_terms_bpd gd:1048-1092,
 * losses.py:12-77): mse_out[n] = mean((target - eps_hat)^2); with flag 4 also vb_out[n] (KL for t>0, decoder NLL at t==0,
 * in bits, frozen mean, clip off) * vb_scale.  Layouts as mmd_ddpm_update; deterministic two-stage reduction. */
int64_t mmd_loss_workspace_bytes(int N);
int mmd_loss_terms(const float* x0, const float* xt, const float* model_out, const float* target, const float* tables,
                   const int64_t* t, int T, int N, int F, int C, int HW, int flags, float vb_scale, float* mse_out, float* vb_out,
                   void* workspace, void* stream);

/* ---------------------------------------------------------------- training step: backward kernels (gd:1114-1203 backward)
 * conv wgrad: dW fp32 [Cout][ntaps*Cin] += dY^T gather(X) (same tap semantics as mmd_conv_gemm; caller zeroes dW), db
 * (nullable, fp32 [Cout]) += column sums of dY.  conv dgrad = mmd_conv_gemm(dY, W^T-packed, taps negated).  bf16 with >= 64 channels on
 * both sides: 128 x 128 MFMA tiles split over M with fp32 atomics (the accumulation order - and the last bit - varies from run to run,
 * like the reference's autograd); 9-tap convs stage row-major by descriptor DMA and read both operands with transposing LDS reads. */
int mmd_conv_wgrad(int dtype, const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, float* db, int M, int Cout, int Cin,
                   int ntaps, const int* taps, int D0, int D1, int D2, int torch_layout, void* stream);
/* Re-pack every conv weight in ONE launch after an optimizer step.  descs_dev: device array of n records
 *   { const float* src; void* fwd; void* bwd; int Cout, Cin, nt; int block_start; }   (block_start = running sum of mmd_pack_blocks(Cout, Cin, nt):
 *   one workgroup per tile of 32 output x (216 / nt) input channels x all taps, staged through LDS so that both operands are written in runs; nt <= 27)
 * src fp32 [Cout][Cin][nt] (torch conv layout) -> fwd [Cout][nt*Cin] (mmd_conv_gemm operand of the forward conv) and
 * bwd [Cin][nt*Cout] (operand of the data-gradient conv), both in `dtype`.  mmd_conv_wgrad with torch_layout = 1 accumulates
 * dW directly in [Cout][Cin][nt], i.e. straight into the parameter's .grad. */
int mmd_pack_blocks(int Cout, int Cin, int nt);     /* workgroups of one record (-1: unsupported tap count) */
int mmd_pack_conv_weights(int dtype, const void* descs_dev, int n, int total_blocks, void* stream);
/* Gradient counterpart, once per step: for every record (same struct; src = the parameter's fp32 .grad [Cout][Cin][nt], fwd = an
 * fp32 packed accumulation buffer [Cout][nt*Cin] that mmd_conv_wgrad filled with coalesced atomics) grad += packed, packed = 0. */
int mmd_unpack_conv_grads(const void* descs_dev, int n, int total_blocks, void* stream);
/* GroupNorm32(+FiLM)(+SiLU) backward; a, b, mr from the forward mmd_gn_stats; dgamma/dbeta accumulate; dfilm (nullable)
 * [S, >=2C] receives (dscale | dshift); workspace (S*C*2 + S*64) floats. */
int mmd_gn_bwd(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx, int64_t rows, int C, int S,
               int Tn, int inner, int64_t outer_stride, int64_t inner_stride, int64_t tstride, const float* a, const float* b,
               const float* mr, const float* gamma, const float* beta, const float* film, int64_t film_ld, int act, float* dgamma,
               float* dbeta, float* dfilm, int64_t dfilm_ld, float* workspace, void* stream);
/* The same with a CALLER-KEPT workspace whose first S*C*2 floats are zero on entry; they are zero again on exit (the parameter stage
 * clears what it read), so the fill launch in front of every norm's backward is gone (243 per training step, mtu:280-330).  One
 * workspace per stream; zero it once after allocation. */
int mmd_gn_bwd_ws0(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx, int64_t rows, int C, int S,
                   int Tn, int inner, int64_t outer_stride, int64_t inner_stride, int64_t tstride, const float* a, const float* b,
                   const float* mr, const float* gamma, const float* beta, const float* film, int64_t film_ld, int act, float* dgamma,
                   float* dbeta, float* dfilm, int64_t dfilm_ld, float* workspace, void* stream);
/* Attention backward for every attention of the model (strided + windowed row descriptor, see mmd_attn_bwd.hip). */
int mmd_attn_bwd(int dtype, const void* Q, int64_t ldq, int q_off, const void* KV, int64_t ldkv, int k_off, int v_off, const void* O,
                 int64_t ldo, const void* dO, int64_t lddo, void* dQ, int64_t lddq, int dq_off, void* dKV, int64_t lddkv, int dk_off,
                 int dv_off, float* lse_ws, float* dsum_ws, int heads, int ch, int nb, int G, int q_inner, int64_t q_outer,
                 int64_t q_istride, int64_t q_tstride, int q_total, int q_per_group, int k_inner, int64_t k_outer, int64_t k_istride,
                 int64_t k_tstride, int k_mod, int k_per_group, int win, const int* shift_dev, void* stream);
/* bf16 MFMA pair for contiguous-row attention: the training forward also returns the log2-domain log-sum-exp per (query
 * row, head); the backward recomputes P from it (no stored probabilities).  Same row / window arguments as mmd_attn_fwd. */
int mmd_attn_fwd_lse(int dtype, const void* Q, int64_t ldq, int q_off, const void* KV, int64_t ldkv, int k_off, int v_off, void* O,
                     int64_t ldo, int heads, int ch, int nb, int G, int64_t q_rows_per_batch, int q_per_group,
                     int64_t k_rows_per_batch, int k_per_group, int win, const int* shift_dev, float* lse2_out, void* stream);
int mmd_attn_bwd_mfma(const void* Q, int64_t ldq, int q_off, const void* KV, int64_t ldkv, int k_off, int v_off, const void* O,
                      int64_t ldo, const void* dO, int64_t lddo, void* dQ, int64_t lddq, int dq_off, void* dKV, int64_t lddkv,
                      int dk_off, int dv_off, const float* lse2, float* dsum_ws, int heads, int ch, int nb, int G,
                      int64_t q_rows_per_batch, int q_per_group, int64_t k_rows_per_batch, int k_per_group, int win,
                      const int* shift_dev, void* stream);
/* DDIM step for one stream (ddim_sample gd:821-901; flag 8 = ddim_reverse_sample gd:903-953).  tables as mmd_ddpm_update,
 * tab3 [3,T] fp32 = alphas_cumprod, alphas_cumprod_prev, alphas_cumprod_next; flags 1 clip, 2 model predicts x0, 4 learned sigma. */
int mmd_ddim_update(const float* x, const float* model_out, const float* noise, float* out, float* x0_out, const float* tables,
                    const float* tab3, const int64_t* t, int T, int N, int F, int C, int HW, int flags, float eta, void* stream);
/* out[n,:] = (ca[t_n] a + cb[t_n] b) cs[t_n]: the _predict_* / q_posterior helpers (gd:170-229,345-366); NULL table = 1, b nullable. */
int mmd_lincomb_t(const float* a, const float* b, float* out, const float* ca, const float* cb, const float* cs, const int64_t* t,
                  int N, int64_t per_sample, void* stream);
/* out = ca a + cb b + cc c (b, c nullable): solver update combinations (multimodal_dpm_solver_plus.py:520-1100). */
int mmd_lincomb(const float* a, float ca, const float* b, float cb, const float* c, float cc, float* out, int64_t n, void* stream);
/* y = (dst dtype)(x * scale), fp32 <-> bf16: the bf16 payload of the data-parallel gradient all-reduce (reference: DDP reduces
 * the fp32 / fp16 gradients it is given, multimodal_train_util.py:127-136) and its widening + 1/world scaling afterwards. */
int mmd_cast(const void* x, int src_dtype, void* y, int dst_dtype, float scale, int64_t n, void* stream);
/* backward of mmd_ddpm_update's sample w.r.t. x and the model output (gradient-guided sampling gd:722-817; fixed variance). */
int mmd_ddpm_update_bwd(const float* x, const float* model_out, const float* dsample, float* dx, float* dmodel_out, const float* tables,
                        const int64_t* t, int T, int N, int64_t per_sample, int flags, void* stream);
/* DPM-Solver++ dynamic thresholding (multimodal_dpm_solver_plus.py:419-440): out[n] = q-quantile of |x[n,:]| (torch.quantile,
 * linear interpolation), exact radix select; then x[n,:] = clamp(x, -s, s) / (s / max_val) with s = max(s_n, 1), in place. */
int mmd_abs_quantile(const float* x, int N, int64_t per_sample, float q, float* out, void* stream);
int mmd_clamp_scale(float* x, const float* s, float max_val, int N, int64_t per_sample, void* stream);
/* adaptive-step error term (dpm:1088-1149): out[n] += sum(((hi - lo) / max(atol, rtol max(|lo|, |prev|)))^2), fp64, caller zeroes. */
int mmd_dpm_err(const float* hi, const float* lo, const float* prev, float atol, float rtol, int N, int64_t per_sample, double* out,
                void* stream);
/* SR model input (image_unet.py:704-715): out [N,2C,H,W] = concat(x [N,C,H,W], bilinear(low [N,C,h,w] -> H x W)), fp32. */
int mmd_bilinear_concat(const float* x, const float* low, float* out, int N, int C, int H, int W, int h, int w, void* stream);
/* The same tensor as channels-last rows [(n, y, x), Cpad] in `dtype` (channels 2C..Cpad zero): feeds the SR stem as an implicit GEMM. */
int mmd_bilinear_concat_rows(int dtype, const float* x, const float* low, void* out, int N, int C, int H, int W, int h, int w, int Cpad,
                             void* stream);
/* Gradient of sum_n(dmse[n] mse[n] + dvb[n] vb[n]) of mmd_loss_terms w.r.t. the model output (same layout, fp32): the mean channels
 * get the mse gradient only (the vb term detaches the mean, gd:1147-1151), the variance channels the KL / decoder-NLL gradient. */
int mmd_loss_terms_bwd(const float* x0, const float* xt, const float* model_out, const float* target, const float* tables,
                       const int64_t* t, int T, int N, int F, int C, int HW, int flags, float vb_scale, const float* dmse,
                       const float* dvb, float* g_model_out, void* stream);
/* backward of mmd_attn_small_fwd (temporal attention): dQKV rows [dq | dk | dv], same slice geometry. */
int mmd_attn_small_bwd(int dtype, const void* QKV, int64_t ld, const void* dO, int64_t lddo, void* dQKV, int64_t ldd, int C, int heads,
                       int S, int Tn, int inner, int64_t outer_stride, int64_t inner_stride, int64_t tstride, void* stream);
/* out = silu(x) (dy NULL) or dy*silu'(x); d(mse loss)/d(out); AdamW (+EMA, nn.py:128-138) on flat fp32 buffers. */
int mmd_timestep_embedding(const void* t, int t_kind, int N, int dim, float* out, void* stream);   /* nn.py:192-210 */
int mmd_silu(int dtype, const void* x, const void* dy, void* out, int64_t n, void* stream);
int mmd_dropout(int dtype, const void* x, const uint8_t* mask, float scale, void* out, int64_t n, void* stream);   /* nn.Dropout, unet:376 */
int mmd_mse_grad(const float* out, const float* target, const float* w, float* g, int N, int64_t per_sample, void* stream);
int mmd_adamw_step(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1, float beta2, float eps,
                   float weight_decay, int step, float ema_rate, void* stream);

#ifdef __cplusplus
}
#endif
#endif
