// Fused temporal-attention block of the video stream (bf16, 4 heads, 16 frames; C = 256 channels - the ds2 level; the C = 384 / 512
// instances of rounds 4 were REMOVED in round 5: built, tested and measured slower than the four launches at ds4 / ds8): ONE launch for
//     y = x + proj_out( attention_over_frames( qkv( GroupNorm32(x) ) ) )
// i.e. SingleModalAtten with the rows of a pixel as the sequence (/root/reference/mm_diffusion/multimodal_unet.py:246-287, used at
// :485-493; GroupNorm32 = nn.py:16-33; QKVAttention = unet:290-330) - and, optionally, the proj_out + residual of the SPATIAL
// attention block in front of it.  Unfused this is four (five) launches per block - mmd_gn_small, the qkv 1x1 conv,
// mmd_attn_small_fwd, the proj_out 1x1 conv with the residual - that write and re-read the normalised tensor, the 3 C-wide qkv
// tensor and the attention output: 365 MB of traffic for 100 MB of input + residual + output at the ds2 level (134 us per block), and at
// the small levels (ds4 / ds8: 16384 / 4096 rows) five launches that are each bound by launch + first-operand latency.
//
// A sequence is the 16 frames of ONE pixel, so a workgroup that owns 8 pixels x 16 frames (128 rows) has everything the block needs;
// nothing but x, the weights and y touches memory.  The chain runs in REGISTERS: the accumulators of one v_mfma_f32_32x32x16_bf16 are
// packed (bias added, rounded to bf16 - the roundings of the unfused path's stored tensors) straight into the operand registers of
// the next, without a shuffle or an LDS transpose (tools/tattn_model.py checks the algebra lane by lane on the CPU, for all three head
// widths):
//   * a wave owns 32 rows = 2 pixels x 16 frames (row l31: pixel l31 >> 4, frame l31 & 15), loaded once as MFMA operand fragments
//     (lane (l31, half): channels 16 cg + 8 half .. + 8 of its row, cg < C / 16) - like the row-strip GEMM, the activations are
//     stationary and the weights stream through LDS;
//   * GroupNorm32 over (16 frames x C / 32 channels) of a pixel, two-pass like mmd_gn_small.  C = 256: a group is the lane's own
//     8-channel vector and the 16 frames are the 16 lanes of a DPP row -> two row reductions per vector;
//   * q, k   : D = W x^T (A = weight fragment, B = x): lane (row, half) ends up with channels 8 q + 4 half + j of its row (i = 4 q + j);
//     packing i = 8 s .. 8 s + 7 gives the operand of k-step s, and q and k carry the SAME channel in the same (half, element) slot,
//     which is all the contraction S^T = k q^T (A = k, B = q) needs;
//   * S^T    : lane (query row, half) holds its scores against keys 8 q + 4 half + j; the keys of the query's own pixel p = l31 >> 4
//     are i in [8 p, 8 p + 8) plus the same eight of the partner half-wave (v_permlane32_swap for max and sum): softmax in fp32,
//     exp2 with the scale folded in, P as a bf16 hi + lo pair (the numerics of mmd_attn_small_fwd), zero for the other pixel's keys;
//   * v^T    : the SAME two fragments with the operands swapped (A = x, B = weights) leave lane (channel l31, half) with the values of
//     keys 8 q + 4 half + j - the key order P has -, so O^T = v^T P needs no transpose either;
//   * O^T    : lane (query row, half) holds channels 8 q + 4 half + j of the head's 32-channel sub-tile: packed, it is the B operand of
//     the projection for the 16-channel block [CH h + 32 a + 16 s, + 16) in the order 8 (e >> 2) + 4 half + (e & 3); the packed
//     proj_out weight has its K columns in that order (mmd_tattn_pack), so the standard fragment read serves;
//   * out    : bias + residual (re-read: L2-hot) in the row-strip epilogue's form (v_permlane32_swap pairs the half-waves' 4-channel
//     groups into 16-byte stores) and, for the GroupNorm that consumes y, quad statistics records: the 64 rows of a wave PAIR (the 16
//     frames of 4 consecutive pixels) are one record - the odd wave parks its half in LDS, the even one adds it after the next
//     barrier and writes (record n HW / 4 + (pixel >> 2): the producer's own order inside a sample, like the fused VideoConv -
//     engine: perm_unit);
//   * front stage (A != NULL): x = X + A Wpre^T + bias_pre, the spatial block's proj_out + residual, with the same GEMM + epilogue;
//     the result goes to MID (bf16, what the unfused path stores), is re-read from there as this wave's operand fragments (the
//     epilogue's 16-byte pieces ARE fragments) and, at the end, as the residual.
// Weights: chunks of CCH output rows x C (CCH = 64 at C = 256, 32 above; <= 32 KB) in the order [front stage] | per head: q, k, v |
// proj_out, pre-packed as the LDS image (128-byte rows per 64-channel plane, 16-byte chunks XOR-swizzled by (row >> 1) & 7) so a
// chunk is linear 1 KB DMA pieces (buffer_load ... lds); two stages, the next chunk in flight under the current one's MFMAs, one raw
// s_barrier per chunk; the weight fragments of K step st + 1 are read before the MFMAs of step st.
// Workgroup = 4 waves = 128 rows.  C = 256: 72 KB of LDS, <= 256 VGPRs, TWO workgroups per CU with independent barriers (measured,
// tools/tattn_bench.py, 65536 rows: 50 us against 107 us for the four launches; 256-row workgroups of 8 waves with three stages:
// 49 us there but 38 against 28 us at 16384 rows; 4 waves x 64 rows at one wave per SIMD: 56 us).  (C = 384 / 512 - head widths 96 / 128 -
// needed ~350 / ~430 VGPRs, one wave per SIMD: ds4 50 vs 63 us alone and neutral in the step, ds8 74 vs 44 us; removed.)
// SQ counters at C = 256 (profiles/r04_tattn_pmc_sq.txt): ~7.6 VALU instructions per MFMA (bias, pack, norm, softmax, epilogue), MFMA
// pipe busy 25 % of the SIMD cycles: bound by VALU issue + dependency stalls at two waves per SIMD, not by the matrix pipe.
#include "mmd_common.h"

typedef __attribute__((address_space(3))) void* lptr_t;

struct TAttnParams {
  const char* X; int64_t ldx;
  const char* A; int64_t lda;          // front stage (nullable): the spatial block's attention output
  char* MID; int64_t ldm;              // front stage: receives x = X + pre(A) + bias_pre
  const float* bpre;
  const char* Wf; int wf_bytes;
  const float* bqkv; const float* bproj;
  const float* gamma; const float* beta;
  char* Y; int64_t ldy;
  int N, HW;
  float eps, sc;                       // sc = log2(e) / sqrt(C / 4)
  float* stats; int64_t stats_ld;
};

__device__ __forceinline__ void ta_mma(const u32x4& a, const u32x4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// total over the 16 lanes of a DPP row, in every lane of the row
__device__ __forceinline__ float ta_row16_total(float v) {
  v = dpp_add<0x128, 0xf>(v);      // row_ror:8
  v = dpp_add<0x124, 0xf>(v);      // row_ror:4
  v = dpp_add<0x122, 0xf>(v);      // row_ror:2
  return dpp_add<0x121, 0xf>(v);   // row_ror:1
}
__device__ __forceinline__ float ta_pair_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float ta_pair_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int C>
struct TACfg {
  static constexpr int CH = C / 4;                       // head width 64 / 96 / 128
  static constexpr int NK = C / 16;                      // k-steps of one GEMM over the channels
  static constexpr int KSP = C / 64;                     // 64-channel planes of a weight chunk
  static constexpr int NSUB = CH / 32;                   // 32-channel sub-tiles per head
  static constexpr int CCH = C == 256 ? 64 : 32;         // output rows per weight chunk
  static constexpr int NA = CCH / 32;
  static constexpr int PLANE_B = CCH * 128, STAGE_B = KSP * PLANE_B;
  static constexpr int NQC = CH / CCH;                   // chunks per (head, q | k | v)
  static constexpr int NPC = C / CCH;                    // chunks of a C x C projection
  static constexpr int QPG = C / 128;                    // channel quads per group
  static constexpr int TAB_F = 5 * C + 2 * C + 256 + 4 * 2 * (C / 4) + 4 * 2 * 32 * 2;   // floats after the weight stages
};

template <int C, bool PRE>
__global__ __launch_bounds__(256, (C == 256 ? 2 : 1)) void tattn_kernel(const TAttnParams p) {
  using G = TACfg<C>;
  constexpr int NW = 4, NST = 2;
  constexpr int CH = G::CH, NK = G::NK, NSUB = G::NSUB, CCH = G::CCH, NA = G::NA, NQC = G::NQC, NPC = G::NPC;
  constexpr int STAGE_B = G::STAGE_B, PLANE_B = G::PLANE_B;
  constexpr int NP = STAGE_B / 1024 / NW;                 // 1 KB DMA pieces per wave and chunk
  constexpr int CB = PRE ? NPC : 0;                       // chunks of the front stage
  constexpr int NCH = CB + 12 * NQC + NPC;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                                        // [2 stages][KSP planes][CCH rows][128 B]
  float* sB = (float*)(smem + NST * STAGE_B);             // [q C | k C | v C | proj C | front stage C] biases
  float* sG = sB + 5 * C;                                 // [gamma C | beta C]
  float* sR = sG + 2 * C;                                 // [2 parities][2 wave pairs][NA * 2][2 halves][2 quads][2] half-record statistics
  float* sQ = sR + 256;                                   // GroupNorm (C > 256): [4 waves][2 pixels][C / 4] quad totals
  float* sM = sQ + 4 * 2 * (C / 4);                       //                      [4 waves][2 pixels][32 groups][mean, rstd]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int bpn = p.HW >> 3;
  const int n = blockIdx.x / bpn, pix0 = (blockIdx.x - n * bpn) * 8;

  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wf, 0, p.wf_bytes, 0x00020000);
  auto issue = [&](int stage, int c) __attribute__((always_inline)) {      // chunk c -> stage: linear 1 KB pieces
#pragma unroll
    for (int i = 0; i < NP; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lptr_t)(sW + stage * STAGE_B + (i * NW + wave) * 1024), 16, lane * 16,
                                               c * STAGE_B + (i * NW + wave) * 1024, 0, 0);
  };
  issue(0, 0);

  // ---- the wave's rows as operand fragments
  const int64_t rowi = ((int64_t)n * 16 + (l31 & 15)) * p.HW + pix0 + 2 * wave + (l31 >> 4);
  u32x4 xa[NK];
  {
    const char* ap = PRE ? p.A + (rowi * p.lda + half * 8) * 2 : p.X + (rowi * p.ldx + half * 8) * 2;
#pragma unroll
    for (int cg = 0; cg < NK; ++cg) xa[cg] = *(const u32x4*)(ap + cg * 32);
  }
  for (int t = tid; t < C; t += 256) {
    sB[t] = p.bqkv[t]; sB[C + t] = p.bqkv[C + t]; sB[2 * C + t] = p.bqkv[2 * C + t]; sB[3 * C + t] = p.bproj[t];
    if (PRE) sB[4 * C + t] = p.bpre[t];
    sG[t] = p.gamma[t]; sG[C + t] = p.beta[t];
  }

  // ---- chunk pipeline
  const int xsw = (l31 >> 1) & 7;
  const char* fbase = sW + l31 * 128;
  int choff[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) choff[c4] = ((2 * c4 + half) ^ xsw) * 16;

  auto chunk_top = [&](int c) __attribute__((always_inline)) {      // chunk c landed for every wave; the other stage is free again
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + 1 < NCH) issue((c + 1) & 1, c + 1);
  };
  // acc[a] = the chunk's sub-tile a (32 weight rows) against the fragments B over K = C.  swap: A = rows, B = weights.
  // The weight fragments of K step st + 1 are requested BEFORE the MFMAs of step st (two register sets; sched_barrier pins the
  // order: left alone the compiler issues a step's reads, waits for them, issues its MFMAs - one LDS latency per step).
  auto gemm = [&](int c, const u32x4 (&B)[NK], f32x16 (&acc)[NA], bool swap) __attribute__((always_inline)) {
    const char* fb = fbase + (c & 1) * STAGE_B;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    u32x4 fw[2][NA];
    auto ldfw = [&](int buf, int st) __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < NA; ++a) fw[buf][a] = *(const u32x4*)(fb + (st >> 2) * PLANE_B + a * 4096 + choff[st & 3]);
    };
    ldfw(0, 0);
#pragma unroll
    for (int st = 0; st < NK; ++st) {
      if (st + 1 < NK) ldfw((st + 1) & 1, st + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        if (swap) ta_mma(B[st], fw[st & 1][a], acc[a]);
        else ta_mma(fw[st & 1][a], B[st], acc[a]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  f32x16 acc[NA];
  const int64_t rec = (int64_t)blockIdx.x * 2 + (wave >> 1);
  // C x C projection with the row-strip epilogue: Y = R + B W^T + bias, chunks c0 .. c0 + NPC (a runtime loop: nothing here is
  // indexed by the chunk).  stats: quad records of Y through the wave pairs.
  float keep[NA * 2][2];
  auto flush = [&](int j) __attribute__((always_inline)) {          // after the barrier that follows chunk j's epilogue
    if (p.stats && (wave & 1) == 0 && (l31 >> 1) == 8) {
#pragma unroll
      for (int aj = 0; aj < NA * 2; ++aj) {
        const float* o = sR + (((((j & 1) * 2 + (wave >> 1)) * (NA * 2) + aj) * 2 + half) * 2 + (l31 & 1)) * 2;
        const int col = CCH * j + 16 * aj + 8 * half;
        float* d = p.stats + (rec * p.stats_ld + (col >> 2) + (l31 & 1)) * 2;
        d[0] = keep[aj][0] + o[0];
        d[1] = keep[aj][1] + o[1];
      }
    }
  };
  auto projection = [&](int c0, const u32x4 (&B)[NK], const float* bias, const char* R, int64_t ldr, char* Y, int64_t ldy, bool stats)
                        __attribute__((always_inline)) {
    for (int j = 0; j < NPC; ++j) {
      chunk_top(c0 + j);
      if (stats && j > 0) flush(j - 1);
      gemm(c0 + j, B, acc, false);
      u32x4 rres[NA][2];
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) rres[a][j2] = *(const u32x4*)(R + (rowi * ldr + CCH * j + 32 * a + 16 * j2 + 8 * half) * 2);
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
          const int col = CCH * j + 32 * a + 16 * j2 + 8 * half;
          const f32x4 b0 = *(const f32x4*)(bias + col), b1 = *(const f32x4*)(bias + col + 4);
          float v[8], rf[8];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][8 * j2 + jj]), __float_as_uint(acc[a][8 * j2 + 4 + jj]), false, false);
            v[jj] = __uint_as_float(sw[0]);
            v[4 + jj] = __uint_as_float(sw[1]);
          }
          Elt<__bf16>::unpack(rres[a][j2], rf);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) { v[jj] = (v[jj] + b0[jj]) + rf[jj]; v[4 + jj] = (v[4 + jj] + b1[jj]) + rf[4 + jj]; }
          const u32x4 pk = Elt<__bf16>::pack(v);
          *(u32x4*)(Y + (rowi * ldy + col) * 2) = pk;
          if (stats && p.stats) {                           // statistics of the values as STORED: the lane's 8 channels = two quads
            float sf[8], u[4] = {0.f, 0.f, 0.f, 0.f};
            Elt<__bf16>::unpack(pk, sf);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              u[0] += sf[jj];
              u[1] += sf[4 + jj];
              u[2] += sf[jj] * sf[jj];
              u[3] += sf[4 + jj] * sf[4 + jj];
            }
            const float t0 = halfwave_total(u[0]), t1 = halfwave_total(u[1]), t2 = halfwave_total(u[2]), t3 = halfwave_total(u[3]);
            const float msum = (l31 & 1) ? t1 : t0, msq = (l31 & 1) ? t3 : t2;
            keep[a * 2 + j2][0] = msum;
            keep[a * 2 + j2][1] = msq;
            if ((wave & 1) && (l31 >> 1) == 8) {
              float* o = sR + (((((j & 1) * 2 + (wave >> 1)) * (NA * 2) + a * 2 + j2) * 2 + half) * 2 + (l31 & 1)) * 2;
              o[0] = msum;
              o[1] = msq;
            }
          }
        }
    }
  };

  // ---- front stage: x = X + A Wpre^T + bias_pre -> MID; the epilogue's 16-byte pieces are this lane's operand fragments of x
  if constexpr (PRE) {
    projection(0, xa, sB + 4 * C, p.X, p.ldx, p.MID, p.ldm, false);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* ap = p.MID + (rowi * p.ldm + half * 8) * 2;
#pragma unroll
    for (int cg = 0; cg < NK; ++cg) xa[cg] = *(const u32x4*)(ap + cg * 32);
  } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // the tables (bias, gamma / beta) are in LDS
  }
  const char* resid = PRE ? (const char*)p.MID : p.X;       // residual of the last stage
  const int64_t ldres = PRE ? p.ldm : p.ldx;

  // ---- GroupNorm32 over (16 frames, C / 32 channels) of a pixel, two-pass like mmd_gn_small, applied in place
  static_assert(C == 256, "groups of 8 channels: one 8-channel vector per group (the 384 / 512-channel instances - groups of 12 / 16 channels through a wave-private LDS table - were removed in round 5: measured slower than the four launches at ds4 / ds8)");
  {
#pragma unroll
    for (int cg = 0; cg < NK; ++cg) {
      float x[8];
      Elt<__bf16>::unpack(xa[cg], x);
      float s = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
      s = ta_row16_total(s);
      const float mean = s * (1.f / 128.f);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = x[e] - mean; q += d * d; }
      q = ta_row16_total(q);
      const float rstd = rsqrtf(q * (1.f / 128.f) + p.eps);
      const float* gp = sG + cg * 16 + half * 8;
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        const f32x4 g4 = *(const f32x4*)(gp + e), b4 = *(const f32x4*)(gp + C + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float av = rstd * g4[k];
          const float bv = b4[k] - mean * av;
          x[e + k] = x[e + k] * av + bv;
        }
      }
      u32x4 y = Elt<__bf16>::pack(x);
      asm volatile("" : "+v"(y.x), "+v"(y.y), "+v"(y.z), "+v"(y.w));     // pin: keep the arithmetic here, not sunk into the MFMA loop
      xa[cg] = y;
    }
  }

  // ---- per head: q, k (W x^T), v^T (operands swapped), attention, O^T -> operands of the projection
  u32x4 oo[NK];                                            // [h (CH / 16) + 2 a + s]
  const bool p1 = (l31 >> 4) != 0;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    u32x4 qo[NSUB][2], ko[NSUB][2], vo[NSUB][2];           // [32-channel sub-tile][s]
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < NQC; ++i) {
        const int c = CB + (3 * h + t) * NQC + i;
        chunk_top(c);
        gemm(c, xa, acc, t == 2);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          const int sub = i * NA + a;
          if (t < 2) {                                      // q / k: lane (row, half) holds channels 8 q + 4 half + j; the bias is a vector over i
            f32x4 b4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) b4[q] = *(const f32x4*)(sB + t * C + h * CH + sub * 32 + 8 * q + 4 * half);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = acc[a][8 * s + e] + b4[2 * s + (e >> 2)][e & 3];
              if (t == 0) qo[sub][s] = Elt<__bf16>::pack(v);
              else ko[sub][s] = Elt<__bf16>::pack(v);
            }
          } else {                                          // v^T: lane (channel l31, half) holds keys 8 q + 4 half + j; the bias is a lane scalar
            const float bs = sB[2 * C + h * CH + sub * 32 + l31];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = acc[a][8 * s + e] + bs;
              vo[sub][s] = Elt<__bf16>::pack(v);
            }
          }
        }
      }
    // attention of the head: two pixels side by side (block-diagonal P)
    f32x16 sT;
#pragma unroll
    for (int r = 0; r < 16; ++r) sT[r] = 0.f;
#pragma unroll
    for (int a = 0; a < NSUB; ++a)
#pragma unroll
      for (int s = 0; s < 2; ++s) ta_mma(ko[a][s], qo[a][s], sT);
    float own[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // (opaque copies: left to itself the compiler turns the select of two vector elements into ONE element with a variable
      // index - a 16-way compare / select chain per value, with the lane masks spilled through v_writelane)
      float lo = sT[i], hi = sT[8 + i];
      asm volatile("" : "+v"(lo), "+v"(hi));
      own[i] = p1 ? hi : lo;
    }
    float mx = fmaxf(fmaxf(fmaxf(own[0], own[1]), fmaxf(own[2], own[3])), fmaxf(fmaxf(own[4], own[5]), fmaxf(own[6], own[7])));
    mx = ta_pair_max(mx);
    const float msc = mx * p.sc;
    float e8[8], lo8[8], sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      e8[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(own[i], p.sc, -msc));
      sum += e8[i];
    }
    sum = ta_pair_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < 8; ++i) lo8[i] = e8[i] - bf16_bits_to_f32(f32_to_bf16_bits(e8[i]));
    const u32x4 ph = Elt<__bf16>::pack(e8), pl = Elt<__bf16>::pack(lo8), z = {0u, 0u, 0u, 0u};
    const u32x4 po[2] = {p1 ? z : ph, p1 ? ph : z}, pol[2] = {p1 ? z : pl, p1 ? pl : z};
#pragma unroll
    for (int a = 0; a < NSUB; ++a) {
      f32x16 oT;
#pragma unroll
      for (int r = 0; r < 16; ++r) oT[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        ta_mma(vo[a][s], po[s], oT);
        ta_mma(vo[a][s], pol[s], oT);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = oT[8 * s + e] * inv;
        oo[h * (CH / 16) + 2 * a + s] = Elt<__bf16>::pack(v);
      }
    }
  }

  // ---- proj_out + bias + residual (+ statistics records)
  projection(CB + 12 * NQC, oo, sB + 3 * C, resid, ldres, p.Y, p.ldy, true);
  if (p.stats) {                                            // block-uniform
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    flush(NPC - 1);
  }
}

// Weight image of mmd_tattn_block: chunks of [KSP planes][CCH rows][8 chunks of 16 B], chunk pc of a row holds logical chunk
// pc ^ ((row >> 1) & 7).  Order: [front stage: rows CCH j .. of ITS weight [C, C], plain K order] | per head h: the rows of
// q_h, k_h, v_h (rows h CH .. + CH of each third of the qkv weight [3 C, C]) | proj_out [C, C] with the 16 K columns of every
// 16-block in the order 8 (e >> 2) + 4 half + (e & 3).
__global__ __launch_bounds__(256) void tattn_pack_kernel(const uint16_t* __restrict__ Wpre, const uint16_t* __restrict__ Wqkv,
                                                         const uint16_t* __restrict__ Wproj, uint16_t* __restrict__ out, int C, int CCH, int with_pre) {
  const int CH = C / 4, KSP = C / 64, NQC = CH / CCH, NPC = C / CCH;
  const int per_chunk = KSP * CCH * 8;                     // 16-byte chunks per weight chunk
  const int nchunk = (with_pre ? NPC : 0) + 12 * NQC + NPC;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= (int64_t)nchunk * per_chunk) return;
  const int c = (int)(o / per_chunk) - (with_pre ? NPC : 0);     // c < 0: front stage
  const int r = (int)(o % per_chunk);
  const int pl = r / (CCH * 8), row = (r / 8) % CCH, pc = r & 7;
  const int lc = pc ^ ((row >> 1) & 7);
  const uint16_t* src;
  bool perm = false;
  if (c < 0) {
    src = Wpre + ((int64_t)(CCH * (c + NPC) + row)) * C;
  } else if (c < 12 * NQC) {
    const int hp = c / NQC, i = c % NQC;                   // hp = 3 h + part
    src = Wqkv + ((int64_t)((hp % 3) * C + (hp / 3) * CH + i * CCH + row)) * C;
  } else {
    src = Wproj + ((int64_t)(CCH * (c - 12 * NQC) + row)) * C;
    perm = true;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int col = perm ? 64 * pl + 16 * (lc >> 1) + 8 * (e >> 2) + 4 * (lc & 1) + (e & 3) : 64 * pl + 8 * lc + e;
    out[o * 8 + e] = src[col];
  }
}

static int tattn_cch(int C) { return C == 256 ? 64 : 32; }
static int tattn_nchunk(int C, int with_pre) {
  const int cch = tattn_cch(C), nqc = C / 4 / cch, npc = C / cch;
  return (with_pre ? npc : 0) + 12 * nqc + npc;
}

extern "C" int64_t mmd_tattn_weight_bytes(int C, int with_pre) {
  if (C != 256) return 0;
  return (int64_t)tattn_nchunk(C, with_pre) * (C / 64) * tattn_cch(C) * 128;
}

// Wqkv [3 C, C] / Wproj [C, C] / Wpre [C, C] (nullable: no front stage): bf16, row-major - the 1x1 conv weights of
// SingleModalAtten.qkv / .proj_out of the temporal block (unet:263-266) and .proj_out of the spatial block in front of it
extern "C" int mmd_tattn_pack(const void* Wpre, const void* Wqkv, const void* Wproj, void* out, int C, void* stream) {
  MMD_REQUIRE(Wqkv && Wproj && out, "tattn_pack: null pointer");
  MMD_REQUIRE(C == 256, "tattn_pack: built for C = 256 (got %d)", C);
  const int64_t chunks16 = mmd_tattn_weight_bytes(C, Wpre != nullptr) / 16;
  hipLaunchKernelGGL(tattn_pack_kernel, dim3((unsigned)((chunks16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)Wpre,
                     (const uint16_t*)Wqkv, (const uint16_t*)Wproj, (uint16_t*)out, C, tattn_cch(C), Wpre ? 1 : 0);
  return mmd_check_launch("tattn_pack");
}

template <int C, bool PRE>
static int launch_tattn(const TAttnParams& p, hipStream_t st) {
  const size_t lds = 2 * (size_t)TACfg<C>::STAGE_B + (size_t)TACfg<C>::TAB_F * sizeof(float);
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)tattn_kernel<C, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "tattn_block: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((tattn_kernel<C, PRE>), dim3(p.N * (p.HW / 8)), dim3(256), lds, st, p);
  return mmd_check_launch("tattn_block");
}

// X / Y: rows (n, f, pixel) x C bf16 (Y may not alias X: other workgroups' residual reads); Wf from mmd_tattn_pack; bias_qkv [3 C],
// bias_proj / gamma / beta [C] fp32; stats (nullable): quad records of Y, one per 64 rows in THIS kernel's row order inside a sample
// (the 16 frames of 4 consecutive pixels), stats[rec * stats_ld + quad] = (sum, sum of squares).
// Front stage (A != NULL; Wf packed with Wpre): the block's input is x = X + A Wpre^T + bias_pre - the spatial attention block's
// proj_out + residual - and MID [rows, C] receives it (scratch the kernel re-reads; also what the unfused path calls the spatial
// block's output).
extern "C" int mmd_tattn_block(const void* X, int64_t ldx, const void* A, int64_t lda, void* MID, int64_t ldm, const void* Wf,
                               const float* bias_pre, const float* bias_qkv, const float* bias_proj, const float* gamma,
                               const float* beta, float eps, void* Y, int64_t ldy, int N, int F, int HW, int C, int heads, float* stats,
                               int64_t stats_ld, void* stream) {
  MMD_REQUIRE(X && Wf && bias_qkv && bias_proj && gamma && beta && Y, "tattn_block: null pointer");
  MMD_REQUIRE(F == 16 && C == 256 && heads == 4, "tattn_block: built for 16 frames, 4 heads, 256 channels (got F=%d C=%d heads=%d)", F, C, heads);
  MMD_REQUIRE(N > 0 && HW > 0 && HW % 8 == 0, "tattn_block: the pixels of a frame must be a multiple of 8 (N=%d HW=%d)", N, HW);
  MMD_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C && ((uintptr_t)X | (uintptr_t)Y | (uintptr_t)Wf) % 16 == 0,
              "tattn_block: 16-byte aligned rows");
  MMD_REQUIRE(X != Y, "tattn_block: in-place is not supported (the residual is re-read)");
  MMD_REQUIRE(!A || (MID && bias_pre && lda % 8 == 0 && ldm % 8 == 0 && lda >= C && ldm >= C && ((uintptr_t)A | (uintptr_t)MID) % 16 == 0 &&
                     MID != Y && MID != X && MID != A),
              "tattn_block: the front stage needs A, a scratch MID (distinct from X, A and Y) and bias_pre, 16-byte aligned rows");
  MMD_REQUIRE(!stats || (stats_ld >= C / 4 && (uintptr_t)stats % 8 == 0), "tattn_block: statistics buffer");
  MMD_REQUIRE(eps > 0.f, "tattn_block: eps");
  TAttnParams p;
  p.X = (const char*)X; p.ldx = ldx; p.A = (const char*)A; p.lda = lda; p.MID = (char*)MID; p.ldm = ldm; p.bpre = bias_pre;
  p.Wf = (const char*)Wf; p.wf_bytes = (int)mmd_tattn_weight_bytes(C, A != nullptr);
  p.bqkv = bias_qkv; p.bproj = bias_proj; p.gamma = gamma; p.beta = beta;
  p.Y = (char*)Y; p.ldy = ldy; p.N = N; p.HW = HW; p.eps = eps;
  p.sc = 1.44269504088896f * (1.0f / sqrtf((float)(C / 4)));
  p.stats = stats; p.stats_ld = stats_ld;
  hipStream_t st = (hipStream_t)stream;
  return A ? launch_tattn<256, true>(p, st) : launch_tattn<256, false>(p, st);
}
