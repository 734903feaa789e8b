// Fused VideoConv '2d+1d' (reference multimodal_unet.py:83-99: video_conv_spatial 3x3 per frame, then video_conv_temporal k=3 per
// pixel) with the GroupNorm32(+SiLU) of the ResBlock in_layers in front of it (unet:339-340,457-458; nn.py:16-33) and the statistics
// of its output for the out_layers norm behind it - one launch instead of {gn_apply | 3x3 conv | k=3 conv} and no intermediate in HBM.
//
// Block = a 4 x 4 pixel patch of ALL 16 frames of one sample: 256 output rows x 128 channels, 8 waves (4 frame groups x 2 column
// halves, a 64 x 64 output tile each, 2 x 2 v_mfma_f32_32x32x16_bf16 accumulators).  Because every frame of the patch lives in the
// block the temporal conv needs no halo in time and recomputes nothing:
//   phase 1 (spatial, K = 9 Cin): per 32-channel chunk the 6 x 6 x 16-frame halo of the patch is staged ONCE (36 KB, double-buffered,
//     rows of 64 bytes, row index hh * 96 + f * 6 + ww so that a tap is a uniform row shift dh * 96 + dw) and the nine taps read
//     shifted windows of it; a step = one tap ROW (three taps x 32 channels: 24 MFMAs per wave) whose 24 KB of weights arrive in a
//     three-slot ring two steps ahead.  GroupNorm(+SiLU) of the input is applied to the staged halo in place (the zero padding pads
//     the NORMALISED activation, so padding slots stay zero), interleaved with the MFMAs of the steps that do not read those rows:
//     halo rows hh 0..3 of chunk c + 1 are fetched at step 0 of chunk c and transformed during its step 2, rows hh 4..5 are fetched
//     at step 1 and transformed during step 0 of chunk c + 1, whose taps (dh = -1) only read rows hh 0..3.
//   transition: the 256 x 128 tile + spatial bias is rounded to bf16 (what the two-launch path stores) and written to LDS as the
//     temporal GEMM's operand image T[plane of 64 channels][18 frames x 16 pixels][128 B] (frames -1 and 16 = zero rows), aliasing
//     the halo stages.
//   phase 2 (temporal, K = 3 x 128): six steps (tap, 64-channel plane) of 16 MFMAs per wave, operand rows = T rows shifted by 16 per
//     frame; weights keep streaming through the same ring.
//   epilogue: straight from the accumulators (v_permlane32_swap pairs -> 16-byte row stores), quad statistics records folded by DPP.
// Every DMA is a buffer_load ... lds through a wave-uniform descriptor; the weights are read from an image packed once per layer
// (mmd_vconv2d1d_pack: step-major slabs, pre-swizzled) so a weight piece is a linear 1 KB copy.  Every step issues a static number
// of DMA instructions (dummies where there is nothing to fetch), so the waits are counted s_waitcnt vmcnt(N) + one raw s_barrier.
// K order: spatial (32-channel chunk, tap, k) - not the (64-channel chunk, tap, k) of tiles 130 / 133, so T may differ from their
// output in the last bf16 bit of a few elements; temporal (tap, channel) like every other main loop.  The layer that runs here is
// chosen by its geometry alone (ops.vconv_fused_ok), never by timing.
#include "mmd_common.h"
#include <type_traits>

struct VConvParams {
  const char* X; int64_t ldx;            // [N * 16 * H * W, Cin] bf16 rows
  const char* Wf; int wf_bytes;          // packed weight image: nchunk * 3 spatial slabs of 24 KB, then 6 temporal slabs of 16 KB
  const float* bias_s; const float* bias_t;
  char* Y; int64_t ldy;                  // [N * 16 * H * W, 128]
  int N, H, W, Cin;
  const float* gn_a; const float* gn_b;  // [S, Cin] fused affine of the input norm (nullptr: plain conv)
  int gn_act; int gn_S; int64_t gn_rows;
  float* stats; int64_t stats_ld;        // quad records of Y (nullptr: none): 4 records per block, index (n * H * W / 16 + patch) * 4 + frame group
};

#define VC_STAGE_B 36864                 // 576 halo rows x 64 B
#define VC_WSLOT_B 24576                 // 3 taps x 128 rows x 64 B
#define VC_WT_B 16384                    // 128 rows x 128 B
#define VC_TPLANE_B 36864                // 288 T rows x 128 B

__device__ __attribute__((aligned(16))) uint32_t g_vc_zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};

__device__ __forceinline__ uint32_t vc_pack2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 t = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, t);
}

template <int GNM, bool R2>                                  // GNM 0: plain conv, 1: fused input affine, 2: affine + SiLU; R2: two-slot weight ring
__global__ __launch_bounds__(512, 2) void vconv2d1d_kernel(const VConvParams p) {
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr bool GN = GNM != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                                   // phase 1: [2 stages][576 rows][64 B]; phase 2: T [2 planes][288 rows][128 B]
  constexpr int NSLOT = R2 ? 2 : 3;
  char* sW = smem + 2 * VC_STAGE_B;                  // [NSLOT slots][24 KB]
  float* sGN = (float*)(smem + 2 * VC_STAGE_B + NSLOT * VC_WSLOT_B);   // [2 chunk parities][a (32) | b (32)]
  float* sBias = sGN + 128;                          // [bias_s (128) | bias_t (128)]
  char* sDummy = (char*)(sBias + 256);               // 256 B: target of the DMA instructions that only keep the per-step count static

  // (not const: the persistent loop re-derives every lane constant per patch from laundered copies, see its top)
  int tid = threadIdx.x;
  int lane = tid & 63;
  int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wc = wave & 1, wr = wave >> 1;
  int half = lane >> 5, l31 = lane & 31;
  int par = 0;                                       // two-slot ring: the slot of the current patch's step 0

  // Persistent blocks (one per CU, grid = min(patches, CUs)): block b walks the patches L = b, b + grid, ... in the XCD-aware order
  // (consecutive patches - sharing halos - stay on one L2; L and b share their XCD while grid % 8 == 0).  While a patch's last two
  // temporal steps run the next patch's first two weight slabs are already on their way, and its first halo stage is requested
  // before the epilogue stores of this one: the ~2 us of launch + first-operand latency per patch hide behind work.
  const int PWc = p.W >> 2, ppf = PWc * (p.H >> 2);
  const int HW = p.H * p.W;
  const int nchunk = p.Cin >> 5;
  const int total = p.N * ppf;
  const uint32_t OOB = 0xfffffff0u;
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wf, 0, p.wf_bytes, 0x00020000);
  const int x_bytes = (int)(((int64_t)16 * HW - 1) * p.ldx * 2 + (int64_t)p.Cin * 2);
  // ---- per-patch state (setup): sample / patch origin, the activation descriptor of the sample, lane-constant halo offsets
  // (pieces of this wave: j = 0..2 rows hh 0..3 = pieces w, w + 8, w + 16; j = 3: piece 24 + w; j = 4: piece 32 + w, waves 0-3),
  // the in-place norm slots (slot i of this thread = 16 bytes at tid * 16 + i * 8192 of a stage; i = 4: waves 0-3 only)
  int n = 0, patch = 0, h0 = 0, w0 = 0;
  auto rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, x_bytes, 0x00020000);
  uint32_t h_off[5];
  unsigned gvalid = 0, glc = 0;
  const float* gn_src = nullptr;
  auto setup = [&](int L) {
    const int q = total >> 3, r = total & 7, xcd = L & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    n = wgid / ppf;
    patch = wgid - n * ppf;
    h0 = (patch / PWc) * 4;
    w0 = (patch % PWc) * 4;
    rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (int64_t)n * 16 * HW * p.ldx * 2), 0, x_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int g = j < 3 ? wave + 8 * j : (j == 3 ? 24 + wave : 32 + wave);
      const int row = 16 * g + (lane >> 2), pc = lane & 3;
      const int hh = row / 96, rem = row - hh * 96, f = rem / 6, ww = rem - f * 6;
      const int y = h0 - 1 + hh, x = w0 - 1 + ww;
      const bool ok = g < 36 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const int logical = pc ^ (hh & 3);
      h_off[j] = ok ? (uint32_t)((((int64_t)f * p.H + y) * p.W + x) * p.ldx * 2 + logical * 16) : OOB;
    }
    gvalid = 0;
    glc = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int s = tid + 512 * i, row = s >> 2, pc = s & 3;
      const int hh = row / 96, rem = row - hh * 96, ww = rem % 6;
      const bool ok = (unsigned)(h0 - 1 + hh) < (unsigned)p.H && (unsigned)(w0 - 1 + ww) < (unsigned)p.W;
      gvalid |= (ok ? 1u : 0u) << i;
      glc |= (unsigned)(pc ^ (hh & 3)) << (2 * i);
    }
    if (GN) {
      const int sidx = min((int)(((int64_t)n * 16 * HW) / p.gn_rows), p.gn_S - 1);
      gn_src = (lane < 32 ? p.gn_a : p.gn_b) + (int64_t)sidx * p.Cin + (lane & 31);
    }
  };
  // padding slots of both stages are zeroed once per patch (an out-of-range DMA lane may or may not write its zero, the norm leaves
  // them alone, and the temporal operand image of the previous patch lay over both stages)
  auto zero_padding = [&]() {
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if ((i < 4 || wave < 4) && !((gvalid >> i) & 1u)) {
        *(u32x4*)(sA + tid * 16 + i * 8192) = u32x4{0u, 0u, 0u, 0u};
        *(u32x4*)(sA + VC_STAGE_B + tid * 16 + i * 8192) = u32x4{0u, 0u, 0u, 0u};
      }
  };
  if (tid < 256) sBias[tid] = tid < 128 ? (p.bias_s ? p.bias_s[tid] : 0.f) : (p.bias_t ? p.bias_t[tid - 128] : 0.f);

  auto dma_dummy = [&]() { __builtin_amdgcn_global_load_lds((gptr_t)g_vc_zero, (lptr_t)sDummy, 4, 0, 0); };
  auto issue_h = [&](int stage, int c, int j) {            // one DMA instruction (j is a literal at every call site)
    const int g = j < 3 ? wave + 8 * j : (j == 3 ? 24 + wave : 32 + wave);
    if (j < 4 || wave < 4)                                   // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lptr_t)(sA + stage * VC_STAGE_B + g * 1024), 16, h_off[j], c * 64, 0, 0);
    else
      dma_dummy();
  };
  auto issue_ws = [&](int slot, int s1, int i) {           // piece i (of three per wave) of spatial step s1 = chunk * 3 + tap row: a linear 1 KB copy
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lptr_t)(sW + slot * VC_WSLOT_B + (wave + 8 * i) * 1024), 16, lane * 16,
                                             s1 * VC_WSLOT_B + (wave + 8 * i) * 1024, 0, 0);
  };
  auto issue_wt = [&](int slot, int s2, int i) {           // piece i (of two per wave) of temporal step s2 = tap * 2 + plane
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lptr_t)(sW + slot * VC_WSLOT_B + (wave + 8 * i) * 1024), 16, lane * 16,
                                             nchunk * 3 * VC_WSLOT_B + s2 * VC_WT_B + (wave + 8 * i) * 1024, 0, 0);
  };
  auto issue_gn = [&](int c) {                             // one DMA instruction: a | b of chunk c (32 + 32 floats) -> ring slot c & 1
    if (GN && wave == 0) __builtin_amdgcn_global_load_lds((gptr_t)(gn_src + c * 32), (lptr_t)(sGN + (c & 1) * 64), 4, 0, 0);
    else dma_dummy();
  };

  // ---- fragment addressing
  int ph, pw;
  int rbB[2];                                              // byte offset of halo row (ph + 1, f_b, pw + 1): tap (0, 0) of this lane's row
  int kw1, wl1;                                            // weight rows of 64 B: chunk key (co >> 2) & 3, lane offset
  int kw2, wl2, tl2;                                       // rows of 128 B (temporal weights, T): chunk key (row >> 1) & 7, lane offsets
  auto lane_constants = [&]() {
    ph = (l31 >> 2) & 3;
    pw = l31 & 3;
#pragma unroll
    for (int b = 0; b < 2; ++b) rbB[b] = (((ph + 1) * 96 + (wr * 4 + b * 2 + (l31 >> 4)) * 6 + pw + 1)) * 64;
    kw1 = (l31 >> 2) & 3;
    wl1 = (wc * 64 + l31) * 64;
    kw2 = (l31 >> 1) & 7;
    wl2 = (wc * 64 + l31) * 128;
    tl2 = (wr * 64 + l31 + 16) * 128;
  };
  lane_constants();

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- GroupNorm(+SiLU) of one 16-byte slot in place, in PIECES of d0 .. d0 + nd - 1 of its four dwords (two channels each) that
  // are placed behind the MFMA groups of a step: a wave hides ~5 other instructions in the shadow of one of its MFMAs
  // (profiles/r04_instruction_rates.txt: v_fma 5.1, v_exp / v_rcp 8.5 cycles per instruction against 32 per MFMA) - a quarter slot
  // (~16 VALU) per group of four MFMAs fits, half a slot (33) does not and its excess is paid in full.
  u32x4 tv;
  uint32_t ty[4];
  auto tr_piece = [&](int stage, int i, int d0, int nd) {
    char* q = sA + stage * VC_STAGE_B + tid * 16 + i * 8192;
    if (d0 == 0) tv = *(const u32x4*)q;
    const float* ap = sGN + stage * 64 + ((glc >> (2 * i)) & 3u) * 8;     // the stage's affine ring slot: a (32) | b (32)
#pragma unroll
    for (int d = d0; d < d0 + nd; ++d) {
      const float a0 = ap[2 * d], a1 = ap[2 * d + 1], b0 = ap[32 + 2 * d], b1 = ap[33 + 2 * d];
      const float w0 = __uint_as_float(tv[d] << 16) * a0 + b0, w1 = __uint_as_float(tv[d] & 0xffff0000u) * a1 + b1;
      ty[d] = vc_pack2(GNM == 2 ? silu_f(w0) : w0, GNM == 2 ? silu_f(w1) : w1);
    }
    if (d0 + nd == 4) {
      const u32x4 y = {ty[0], ty[1], ty[2], ty[3]};
      *(u32x4*)q = ((gvalid >> i) & 1u) ? y : tv;
    }
  };

  // ---- one spatial step: tap row J (dh = J - 1) of chunk c; MORE: another chunk follows.  Six sub-steps k = (tap dw, 16-channel k-step),
  // each {fragment reads two sub-steps ahead | 4 MFMAs | ONE DMA instruction | a quarter slot of the input norm}: the DMA issue (~100
  // cycles apiece beside LDS traffic) and the norm's VALU work run under the MFMAs instead of in front of them.
  // DMA groups (per wave, in issue order) and the counted waits they imply (a step's top wait leaves exactly the previous step's
  // group - minus what this step needs of it - in flight):
  //   J = 0: halo rows hh 0..3 of chunk c + 1 (3), then the weights of step s + 2 (3)      top of J = 1: vmcnt(3) - the halo pieces landed
  //   J = 1: weights of step s + 2 (3), halo rows hh 4..5 of chunk c + 1 (2)               top of J = 2: vmcnt(5)
  //   J = 2: weights of step s + 2 (3), affine rows of chunk c + 2 (GN: 1)                 top of J = 0: vmcnt(3 + GN)
  //   last chunk: J = 0 weights (3); J = 1 / J = 2 the temporal steps 0 / 1 (2 each)       tops: vmcnt(3), vmcnt(2), (temporal) vmcnt(2)
  // Norm slots (GN): 0, 1, 2 of the next stage in twelve quarters over the sub-steps of J = 1 and J = 2 (their pieces landed with
  // the top of J = 1), slots 3 (, 4) of THIS stage at J = 0 (rows hh 4..5: landed with this step's top wait, not read by the dh = -1 taps).
  // Two-slot ring (R2: 121.75 KB of LDS instead of 145.75, so that a K = 128 row-strip / elementwise workgroup of another launch queue fits
  // beside this one on the CU): the weights of step s + 1 go out FIRST in step s (into the slot step s - 1 left at this step's barrier;
  // slot of a patch's step g = (par + g) & 1, par flips per patch when the step count is odd), the halo pieces behind them:
  //   J = 0: weights of step s + 1 (3), then halo rows hh 0..3 of chunk c + 1 (3)          top of J = 1: vmcnt(3) - the weights landed
  //   J = 1: weights of step s + 1 (3), halo rows hh 4..5 of chunk c + 1 (2)               top of J = 2: vmcnt(2)
  //   J = 2: weights of step s + 1 (3), affine rows of chunk c + 2 (GN: 1)                 top of J = 0: vmcnt(GN)
  //   last chunk: J = 0 / 1 weights (3); J = 2 temporal step 0 (2)                         tops: vmcnt(0)
  // Norm slots (GN): a wave normalises only rows of its OWN DMA pieces (slot i = piece wave + 8 i), so slot 0 of the next stage starts in
  // the middle of J = 1 behind a counted wait of this wave alone (vmcnt(2): the two weight pieces issued in front of it stay in flight),
  // slots 1, 2 during J = 2 (two halves + four quarters), slots 3 (, 4) of THIS stage at J = 0 as before.
  auto spatial = [&](auto jtag, auto mtag, int c) {
    constexpr int J = decltype(jtag)::value;
    constexpr bool MORE = decltype(mtag)::value;
    if constexpr (R2) {
      if constexpr (J == 0) {
        if constexpr (GN) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else if constexpr (!MORE) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else if constexpr (J == 1) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    } else if constexpr (J == 0) {
      if constexpr (GN) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    } else if constexpr (J == 1) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else if constexpr (MORE) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // ... for every wave; every wave is past its fragment reads of the previous step
    asm volatile("" ::: "memory");
    const int s1n = c * 3 + J + (R2 ? 1 : 2);              // the spatial step whose weights go out now (3-slot ring: slot (J + 2) % 3, the previous step left it)
    const int slot_n = (par + c * 3 + J + 1) & 1;          // R2: its slot; this step reads slot_n ^ 1
    auto dma = [&](int k) {
      if constexpr (R2) {
        // this wave's halo pieces of chunk c + 1 (slot 0's rows among them) before its norm starts behind MFMA group 2: two weight pieces are
        // newer (placed in FRONT of the third so that a reordering of the two statements could only over-wait)
        if constexpr (GN && J == 1 && MORE) { if (k == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        if constexpr (J == 2 && !MORE) {
          if (k < 2) issue_wt(slot_n, 0, k);
        } else {
          if (k < 3) issue_ws(slot_n, s1n, k);
          else if constexpr (J == 0 && MORE) issue_h((c + 1) & 1, c + 1, k - 3);
          else if constexpr (J == 1 && MORE) { if (k < 5) issue_h((c + 1) & 1, c + 1, k); }
          else if constexpr (J == 2 && MORE) { if (GN && k == 3) { if (c + 2 < nchunk) issue_gn(c + 2); else dma_dummy(); } }
        }
      } else if constexpr (J == 0 && MORE) {
        if (k < 3) issue_h((c + 1) & 1, c + 1, k); else issue_ws(2, s1n, k - 3);
      } else if constexpr (J == 0) {
        if (k < 3) issue_ws(2, s1n, k);
      } else if constexpr (J == 1 && MORE) {
        if (k < 3) issue_ws(0, s1n, k); else if (k < 5) issue_h((c + 1) & 1, c + 1, k);
      } else if constexpr (J == 1) {
        if (k < 2) issue_wt(0, 0, k);
      } else if constexpr (MORE) {
        if (k < 3) issue_ws(1, s1n, k);
        else if (GN && k == 3) { if (c + 2 < nchunk) issue_gn(c + 2); else dma_dummy(); }
      } else {
        if (k < 2) issue_wt(1, 1, k);
      }
    };
    const bool tr_on = GN && (J == 0 ? c > 0 : MORE);      // block-uniform
    const int tstage = J == 0 ? c & 1 : (c + 1) & 1;       // the stage (= affine ring slot) whose slots this step normalises
    auto tr = [&](int k) {                                 // a piece of the norm behind MFMA group k
      if (!tr_on) return;
      if constexpr (J == 0) {                              // slot 3 in quarters (k = 0..3), slot 4 in halves (k = 4, 5; waves 0-3)
        if (k < 4) tr_piece(tstage, 3, k, 1);
        else if (wave < 4) tr_piece(tstage, 4, 2 * (k - 4), 2);
      } else if constexpr (R2) {
        if constexpr (J == 1) {                            // slot 0 in quarters over k = 2..5, behind this wave's own wait for its halo pieces (in dma(2))
          if (k >= 2) tr_piece(tstage, 0, k - 2, 1);
        } else {                                           // slot 1 in halves (k = 0, 1), slot 2 in quarters (k = 2..5)
          if (k < 2) tr_piece(tstage, 1, 2 * k, 2);
          else tr_piece(tstage, 2, k - 2, 1);
        }
      } else {                                             // slots 0, 1, 2 in twelve quarters over the twelve sub-steps of J = 1, 2
        const int qq = (J - 1) * 6 + k;
        tr_piece(tstage, qq >> 2, qq & 3, 1);
      }
    };
    const char* bW = sW + (R2 ? slot_n ^ 1 : J) * VC_WSLOT_B + wl1;   // 3-slot ring: slot of step 3 c + J = J
    const char* bA[2];
    const int key = (ph + J) & 3;                          // halo rows hh = ph + 1 + dh
    const char* st = sA + (c & 1) * VC_STAGE_B + (J - 1) * (96 * 64) - 64;   // tap (dh, dw = -1)
#pragma unroll
    for (int b = 0; b < 2; ++b) bA[b] = st + rbB[b];
    u32x4 fw[6][2], fa[6][2];
    auto rd = [&](int k) {                                 // sub-step k = tap dw (k >> 1) x 16-channel k-step (k & 1)
      const int dwi = k >> 1, k2 = k & 1;
#pragma unroll
      for (int a = 0; a < 2; ++a) fw[k][a] = *(const u32x4*)(bW + dwi * 8192 + a * 2048 + (((2 * k2 + half) ^ kw1) * 16));
#pragma unroll
      for (int b = 0; b < 2; ++b) fa[k][b] = *(const u32x4*)(bA[b] + dwi * 64 + (((2 * k2 + half) ^ key) * 16));
    };
    auto mm = [&](int k) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[k][a]), __builtin_bit_cast(bf16x8, fa[k][b]), acc[a][b], 0, 0, 0);
    };
    rd(0); rd(1);
    __builtin_amdgcn_sched_barrier(0);
    rd(2); mm(0); dma(0); tr(0);
    __builtin_amdgcn_sched_barrier(0);
    rd(3); mm(1); dma(1); tr(1);
    __builtin_amdgcn_sched_barrier(0);
    rd(4); mm(2); dma(2); tr(2);
    __builtin_amdgcn_sched_barrier(0);
    rd(5); mm(3); dma(3); tr(3);
    __builtin_amdgcn_sched_barrier(0);
    mm(4); dma(4); tr(4);
    __builtin_amdgcn_sched_barrier(0);
    mm(5); dma(5); tr(5);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- the first halo stage, the affine rows of chunks 0 / 1 and the weights of steps 0 / 1 of a patch
  auto issue_first_halo = [&]() {
#pragma unroll
    for (int j = 0; j < 5; ++j) issue_h(0, 0, j);
    if (GN) {
      issue_gn(0);
      if (nchunk > 1) issue_gn(1);
    }
  };
  int L = blockIdx.x;
  setup(L);
  zero_padding();
  issue_first_halo();
#pragma unroll
  for (int i = 0; i < 3; ++i) issue_ws(0, 0, i);
  if constexpr (!R2) {
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_ws(1, 1, i);
  }

  for (;;) {
  // Every lane / wave constant goes through an opaque copy once per patch: what is derived from them (fragment, slot and store
  // addresses) is then recomputed per patch instead of being hoisted out of this loop as ~60 loop-invariant registers - the
  // persistent form of the kernel otherwise needs all 256 VGPRs and spills (the one-patch form: 191).
  asm volatile("" : "+v"(tid), "+v"(lane), "+v"(half), "+v"(l31), "+s"(wave));
  wc = wave & 1;
  wr = wave >> 1;
  lane_constants();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the first operands of this patch (and the previous patch's stores)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (GN) {                                                // chunk 0 is normalised before the first MFMA
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (i < 4 || wave < 4) tr_piece(0, i, 0, 4);
    // (the first step's barrier orders these LDS writes before the first fragment reads)
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  for (int c = 0; c + 1 < nchunk; ++c) {
    spatial(std::integral_constant<int, 0>{}, std::true_type{}, c);
    spatial(std::integral_constant<int, 1>{}, std::true_type{}, c);
    spatial(std::integral_constant<int, 2>{}, std::true_type{}, c);
  }
  spatial(std::integral_constant<int, 0>{}, std::false_type{}, nchunk - 1);
  spatial(std::integral_constant<int, 1>{}, std::false_type{}, nchunk - 1);
  spatial(std::integral_constant<int, 2>{}, std::false_type{}, nchunk - 1);

  // ---- transition: T = bf16(acc + bias_s) as the temporal operand image (aliases the halo stages), frames -1 / 16 = zero rows
  __builtin_amdgcn_s_barrier();                            // every wave is past its last halo read
  asm volatile("" ::: "memory");
  {
    const int zr = tid >> 3, zc = tid & 7;                 // 64 rows x 8 chunks: rows 0..15 and 272..287 of both planes
    const int row = (zr & 15) + ((zr & 16) ? 272 : 0), plane = zr >> 5;
    *(u32x4*)(sA + plane * VC_TPLANE_B + row * 128 + zc * 16) = u32x4{0u, 0u, 0u, 0u};
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int trow = wr * 64 + b * 32 + l31 + 16;
      char* tb = sA + wc * VC_TPLANE_B + trow * 128;
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        // acc[4 q + j] = channel 8 q + 4 half + j of row l31; pair q = 2 j2 with q = 2 j2 + 1: this lane then holds the 8 consecutive
        // channels 16 j2 + 8 half .. + 8 of its row
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][b][8 * j2 + j]), __float_as_uint(acc[a][b][8 * j2 + 4 + j]), false, false);
          v[j] = __uint_as_float(sw[0]);
          v[4 + j] = __uint_as_float(sw[1]);
        }
        const int cl = a * 32 + 16 * j2 + 8 * half;        // channel inside the plane (= inside this wave column's 64)
        const f32x4 b0 = *(const f32x4*)(sBias + wc * 64 + cl), b1 = *(const f32x4*)(sBias + wc * 64 + cl + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
        *(u32x4*)(tb + ((((cl >> 3)) ^ ((trow >> 1) & 7)) * 16)) = Elt<__bf16>::pack(v);
      }
    }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- phase 2: temporal k = 3 over T.  Step s2 = tap * 2 + plane in ring slot s2 % 3 (3 nchunk spatial steps: the ring continues)
  const bool has_next = L + (int)gridDim.x < total;        // block-uniform
  auto temporal = [&](auto stag) {
    constexpr int S2 = decltype(stag)::value;
    // the newest group in flight: the weights of step S2 + 1 (two pieces) - behind step 4 the first slab of the NEXT patch (three)
    // (two-slot ring: the only group in flight is this step's own weights)
    if constexpr (R2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (S2 < 5) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    constexpr int tap = S2 >> 1, plane = S2 & 1;
    const int slot_t = (par + nchunk + S2) & 1;              // R2: 3 nchunk spatial steps came first
    const char* bW = sW + (R2 ? slot_t : S2 % 3) * VC_WSLOT_B + wl2;
    const char* bT = sA + plane * VC_TPLANE_B + tl2 + (tap - 1) * 2048;
    u32x4 fw[4][2], fa[4][2];
    auto rd = [&](int k) {
#pragma unroll
      for (int a = 0; a < 2; ++a) fw[k][a] = *(const u32x4*)(bW + a * 4096 + (((2 * k + half) ^ kw2) * 16));
#pragma unroll
      for (int b = 0; b < 2; ++b) fa[k][b] = *(const u32x4*)(bT + b * 4096 + (((2 * k + half) ^ kw2) * 16));
    };
    auto mm = [&](int k) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw[k][a]), __builtin_bit_cast(bf16x8, fa[k][b]), acc[a][b], 0, 0, 0);
    };
    auto dma = [&](int k) {
      if constexpr (R2) {
        if constexpr (S2 < 5) {
          if (k < 2) issue_wt(slot_t ^ 1, S2 + 1, k);
        } else {                                              // step 5: spatial slab 0 of the next patch (its slot 0 = this step's other slot)
          if (has_next) issue_ws(slot_t ^ 1, 0, k); else dma_dummy();
        }
      } else if constexpr (S2 + 2 < 6) {
        if (k < 2) issue_wt((S2 + 2) % 3, S2 + 2, k);
      } else {                                              // steps 4 / 5: spatial slabs 0 / 1 of the next patch into slots 0 / 1 (three pieces)
        if (has_next) issue_ws(S2 - 4, S2 - 4, k); else dma_dummy();
      }
    };
    rd(0); rd(1);
    __builtin_amdgcn_sched_barrier(0);
    rd(2); mm(0); dma(0);
    __builtin_amdgcn_sched_barrier(0);
    rd(3); mm(1); dma(1);
    __builtin_amdgcn_sched_barrier(0);
    mm(2); dma(2); mm(3);
    __builtin_amdgcn_sched_barrier(0);
  };
  temporal(std::integral_constant<int, 0>{});
  temporal(std::integral_constant<int, 1>{});
  temporal(std::integral_constant<int, 2>{});
  temporal(std::integral_constant<int, 3>{});
  temporal(std::integral_constant<int, 4>{});
  temporal(std::integral_constant<int, 5>{});

  // ---- epilogue from the accumulators: + bias_t -> bf16 -> 16-byte row stores; quad statistics of the values as stored.  The next
  // patch's first halo stage is requested first (the temporal operand image is dead once every wave is past its last fragment read)
  const int64_t mbase = (int64_t)n * 16 * HW + (int64_t)(h0 + ph) * p.W + w0 + pw;
  const int64_t rec = ((int64_t)n * (HW >> 4) + patch) * 4 + wr;
  if (has_next) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    setup(L + (int)gridDim.x);
    zero_padding();
    issue_first_halo();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int col = wc * 64 + a * 32 + 16 * j2 + 8 * half;
      const f32x4 b0 = *(const f32x4*)(sBias + 128 + col), b1 = *(const f32x4*)(sBias + 128 + col + 4);
      float u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][b][8 * j2 + j]), __float_as_uint(acc[a][b][8 * j2 + 4 + j]), false, false);
          v[j] = __uint_as_float(sw[0]);
          v[4 + j] = __uint_as_float(sw[1]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
        const u32x4 pk = Elt<__bf16>::pack(v);
        const int64_t m = mbase + (int64_t)(wr * 4 + b * 2 + (l31 >> 4)) * HW;
        *(u32x4*)(p.Y + (m * p.ldy + col) * 2) = pk;
        if (p.stats) {
          float rf[8];
          Elt<__bf16>::unpack(pk, rf);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            u[0] += rf[j];
            u[1] += rf[4 + j];
            u[2] += rf[j] * rf[j];
            u[3] += rf[4 + j] * rf[4 + j];
          }
        }
      }
      if (p.stats) {                                       // block-uniform: lanes 16 / 17 of each half hold quad 0 / 1 of the lane group's 8 channels
        const float t0 = halfwave_total(u[0]), t1 = halfwave_total(u[1]), t2 = halfwave_total(u[2]), t3 = halfwave_total(u[3]);
        if ((l31 >> 1) == 8) {
          float* d = p.stats + (rec * p.stats_ld + (col >> 2) + (l31 & 1)) * 2;
          d[0] = (l31 & 1) ? t1 : t0;
          d[1] = (l31 & 1) ? t3 : t2;
        }
      }
    }
  if (!has_next) break;
  L += (int)gridDim.x;
  par = (par + nchunk) & 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // trailing dummies
}

// ---- weight image.  Spatial slab s1 = chunk * 3 + dhi: [dwi][co][64 B], 16-byte chunk lc of row co stored at chunk lc ^ ((co >> 2) & 3);
// value = Ws[co][(dhi * 3 + dwi) * Cin + chunk * 32 + lc * 8 + e].  Temporal slab s2 = tap * 2 + plane: [co][128 B], chunk lc at
// lc ^ ((co >> 1) & 7); value = Wt[co][tap * 128 + plane * 64 + lc * 8 + e].
__global__ __launch_bounds__(256) void vconv_pack_kernel(const uint16_t* __restrict__ Ws, const uint16_t* __restrict__ Wt, uint16_t* __restrict__ out,
                                                         int Cin, int64_t total) {
  const int64_t nsp = (int64_t)(Cin >> 5) * 3 * (VC_WSLOT_B / 2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    uint16_t v;
    if (i < nsp) {
      const int s1 = (int)(i / (VC_WSLOT_B / 2)), r = (int)(i - (int64_t)s1 * (VC_WSLOT_B / 2));
      const int c = s1 / 3, dhi = s1 - c * 3;
      const int dwi = r / 4096, r2 = r - dwi * 4096, co = r2 >> 5, pc = (r2 >> 3) & 3, e = r2 & 7;
      const int lc = pc ^ ((co >> 2) & 3);
      v = Ws[(int64_t)co * 9 * Cin + (dhi * 3 + dwi) * Cin + c * 32 + lc * 8 + e];
    } else {
      const int64_t k = i - nsp;
      const int s2 = (int)(k / (VC_WT_B / 2)), r = (int)(k - (int64_t)s2 * (VC_WT_B / 2));
      const int tap = s2 >> 1, plane = s2 & 1;
      const int co = r >> 6, pc = (r >> 3) & 7, e = r & 7;
      const int lc = pc ^ ((co >> 1) & 7);
      v = Wt[(int64_t)co * 384 + tap * 128 + plane * 64 + lc * 8 + e];
    }
    out[i] = v;
  }
}

extern "C" int64_t mmd_vconv2d1d_weight_bytes(int Cin) { return (int64_t)(Cin >> 5) * 3 * VC_WSLOT_B + 6 * VC_WT_B; }

extern "C" int mmd_vconv2d1d_pack(const void* Ws, const void* Wt, void* out, int Cin, int Cout, void* stream) {
  MMD_REQUIRE(Ws && Wt && out, "vconv2d1d_pack: null pointer");
  MMD_REQUIRE(Cout == 128 && Cin > 0 && Cin % 32 == 0, "vconv2d1d_pack: needs 128 output channels and Cin %% 32 == 0 (got %d -> %d)", Cin, Cout);
  const int64_t total = mmd_vconv2d1d_weight_bytes(Cin) / 2;
  hipLaunchKernelGGL(vconv_pack_kernel, dim3(cdiv(total, 256 * 8)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)Ws, (const uint16_t*)Wt,
                     (uint16_t*)out, Cin, total);
  return mmd_check_launch("vconv2d1d_pack");
}

extern "C" int mmd_vconv2d1d(const void* X, int64_t ldx, const float* gn_a, const float* gn_b, int act, int S, int64_t rows_per_slice,
                             const void* Wf, const float* bias_s, const float* bias_t, void* Y, int64_t ldy, int N, int F, int H, int W,
                             int Cin, int Cout, float* stats, int64_t stats_ld, void* stream) {
  MMD_REQUIRE(X && Wf && Y, "vconv2d1d: null pointer");
  MMD_REQUIRE(F == 16 && Cout == 128 && Cin > 0 && Cin % 32 == 0 && N > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0,
              "vconv2d1d: needs 16 frames, 128 output channels, Cin %% 32 == 0 and frame sides that are multiples of 4 (got F=%d Cout=%d Cin=%d H=%d W=%d)",
              F, Cout, Cin, H, W);
  MMD_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= Cin && ldy >= Cout && ((uintptr_t)X | (uintptr_t)Wf | (uintptr_t)Y) % 16 == 0,
              "vconv2d1d: rows must be 16-byte aligned");
  MMD_REQUIRE(((int64_t)16 * H * W - 1) * ldx * 2 + (int64_t)Cin * 2 < 0x7fffffffLL, "vconv2d1d: one sample's rows must stay below 2 GB");
  {  // in-place / overlapping X and Y: other blocks read 3x3 halo rows a neighbouring patch has already overwritten (silent corruption)
    const int64_t M = (int64_t)N * 16 * H * W;
    const uintptr_t x0 = (uintptr_t)X, x1 = x0 + (uintptr_t)(((M - 1) * ldx + Cin) * 2), y0 = (uintptr_t)Y, y1 = y0 + (uintptr_t)(((M - 1) * ldy + Cout) * 2);
    MMD_REQUIRE(x1 <= y0 || y1 <= x0, "vconv2d1d: X and Y overlap (in-place / slices of one buffer are not supported)");
  }
  MMD_REQUIRE((gn_a == nullptr) == (gn_b == nullptr), "vconv2d1d: gn_a and gn_b come together");
  MMD_REQUIRE(!gn_a || (S > 0 && rows_per_slice > 0 && rows_per_slice % ((int64_t)16 * H * W) == 0 && (int64_t)S * rows_per_slice == (int64_t)N * 16 * H * W),
              "vconv2d1d: the fused input norm needs slices of whole samples covering the tensor (S=%d rows=%lld)", S, (long long)rows_per_slice);
  MMD_REQUIRE(!stats || ((uintptr_t)stats % 8 == 0 && stats_ld >= Cout / 4), "vconv2d1d: statistics records: 8-byte aligned, stats_ld >= Cout / 4");
  VConvParams p;
  p.X = (const char*)X; p.ldx = ldx; p.Wf = (const char*)Wf; p.wf_bytes = (int)mmd_vconv2d1d_weight_bytes(Cin);
  p.bias_s = bias_s; p.bias_t = bias_t; p.Y = (char*)Y; p.ldy = ldy; p.N = N; p.H = H; p.W = W; p.Cin = Cin;
  p.gn_a = gn_a; p.gn_b = gn_b; p.gn_act = act; p.gn_S = S; p.gn_rows = rows_per_slice;
  p.stats = stats; p.stats_ld = stats_ld;
  // MMD_VCONV_RING=2 selects the two-slot weight ring (121.75 KB of LDS instead of 145.75: a K = 128 row-strip or elementwise workgroup of
  // another launch queue fits beside it; weights one step ahead).  Bitwise equal, measured (profiles/r06_vconv_two_slot_ring.txt): the
  // kernel alone + 1 ... 3 %, the step unchanged (11.09 vs 11.04 ms, three alternating runs each) - with 2 x 200 of a SIMD's 512 VGPRs
  // held by this kernel's two waves only kernels of <= 112 VGPRs can join, LDS or not.  Default: three slots.
  const char* ring_env = getenv("MMD_VCONV_RING");           // read per call (a launch plan is recorded once): tests flip it in-process
  const bool ring2 = ring_env && atoi(ring_env) == 2;
  const size_t lds = 2 * VC_STAGE_B + (ring2 ? 2 : 3) * VC_WSLOT_B + 512 + 1024 + 256;
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    const void* fns[6] = {(const void*)vconv2d1d_kernel<0, false>, (const void*)vconv2d1d_kernel<1, false>, (const void*)vconv2d1d_kernel<2, false>,
                          (const void*)vconv2d1d_kernel<0, true>,  (const void*)vconv2d1d_kernel<1, true>,  (const void*)vconv2d1d_kernel<2, true>};
    for (int i = 0; i < 6; ++i) {
      const hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * VC_STAGE_B + (i < 3 ? 3 : 2) * VC_WSLOT_B + 512 + 1024 + 256));
      if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "vconv2d1d: set LDS attr: %s", hipGetErrorString(e));
    }
    attr_set = true;
  }
  int ncu = 0;
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, mmd_device_slot()) != hipSuccess || ncu <= 0) ncu = 256;
  const int total = N * (H / 4) * (W / 4);
  const int grid = total < ncu ? total : ncu;               // persistent blocks, one per CU
  const int gnm = !gn_a ? 0 : (act ? 2 : 1);
#define VC_LAUNCH(G, R) hipLaunchKernelGGL((vconv2d1d_kernel<G, R>), dim3(grid), dim3(512), lds, (hipStream_t)stream, p)
  if (ring2) { if (gnm == 0) VC_LAUNCH(0, true); else if (gnm == 1) VC_LAUNCH(1, true); else VC_LAUNCH(2, true); }
  else { if (gnm == 0) VC_LAUNCH(0, false); else if (gnm == 1) VC_LAUNCH(1, false); else VC_LAUNCH(2, false); }
#undef VC_LAUNCH
  return mmd_check_launch("vconv2d1d");
}
