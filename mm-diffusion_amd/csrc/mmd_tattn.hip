// Fused temporal-attention block of the video stream (bf16, 256 channels, 4 heads of 64, 16 frames): ONE launch for
//     y = x + proj_out( attention_over_frames( qkv( GroupNorm32(x) ) ) )
// i.e. SingleModalAtten with the rows of a pixel as the sequence (/root/reference/mm_diffusion/multimodal_unet.py:246-287, used at
// :485-493; GroupNorm32 = nn.py:16-33; QKVAttention = unet:290-330).  Unfused this is four launches per block - mmd_gn_small, the qkv
// 1x1 conv, mmd_attn_small_fwd, the proj_out 1x1 conv with the residual - that write and re-read the normalised tensor, the 768-wide
// qkv tensor and the attention output: 365 MB of traffic for 100 MB of input + residual + output at the ds2 level, 134 us per block.
//
// A sequence is the 16 frames of ONE pixel, so a workgroup that owns 16 pixels x 16 frames (256 rows) has everything the block needs;
// nothing but x, the weights and y touches memory.  The chain runs in REGISTERS: the accumulators of one v_mfma_f32_32x32x16_bf16 are
// packed (bias added, rounded to bf16 - the roundings of the unfused path's stored tensors) straight into the operand registers of
// the next, without a shuffle or an LDS transpose (tools/tattn_model.py checks the algebra lane by lane on the CPU):
//   * a wave owns 32 rows = 2 pixels x 16 frames (row l31: pixel l31 >> 4, frame l31 & 15), loaded once as MFMA operand fragments
//     (lane (l31, half): channels 16 cg + 8 half .. + 8 of its row, cg < 16) - like the row-strip GEMM, the activations are stationary
//     and the weights stream through LDS;
//   * GroupNorm32 over (16 frames x 8 channels) of a pixel: the 8 channels are the lane's own vector, the 16 frames are the 16 lanes
//     of a DPP row -> two row reductions per vector (two-pass: mean, then centred squares), affine applied in place;
//   * q, k   : D = W x^T (A = weight fragment, B = x): lane (row, half) ends up with channels 8 q + 4 half + j of its row (i = 4 q + j);
//     packing i = 8 s .. 8 s + 7 gives the operand of k-step s, and q and k carry the SAME channel in the same (half, element) slot,
//     which is all the contraction S^T = k q^T (A = k, B = q) needs;
//   * S^T    : lane (query row, half) holds its scores against keys 8 q + 4 half + j; the keys of the query's own pixel p = l31 >> 4
//     are i in [8 p, 8 p + 8) plus the same eight of the partner half-wave (v_permlane32_swap for max and sum): softmax in fp32,
//     exp2 with the scale folded in, P as a bf16 hi + lo pair (the numerics of mmd_attn_small_fwd), zero for the other pixel's keys;
//   * v^T    : the SAME two fragments with the operands swapped (A = x, B = weights) leave lane (channel l31, half) with the values of
//     keys 8 q + 4 half + j - the key order P has -, so O^T = v^T P needs no transpose either;
//   * O^T    : lane (query row, half) holds channels 8 q + 4 half + j of the head's 32-channel sub-tile: packed, it is the B operand of
//     the projection for the 16-channel block [64 h + 32 a + 16 s, + 16) in the order 8 (e >> 2) + 4 half + (e & 3); the packed
//     proj_out weight has its K columns in that order (mmd_tattn_pack), so the standard fragment read serves;
//   * out    : bias + residual (x re-read: L2-hot) in the row-strip epilogue's form (v_permlane32_swap pairs the half-waves' 4-channel
//     groups into 16-byte stores) and, for the GroupNorm that consumes y, quad statistics records: the 64 rows of a wave PAIR (the 16
//     frames of 4 consecutive pixels) are one record - the odd wave parks its half in LDS, the even one adds it after the next
//     barrier and writes (record n HW / 4 + (pixel >> 2): the producer's own order inside a sample, like the fused VideoConv -
//     engine: perm_unit).
// Weights: 16 chunks of 64 output rows x 256 K (q_h, k_h, v_h for h = 0..3, then four quarters of proj_out), pre-packed as the LDS
// image (128-byte rows per 64-channel plane, 16-byte chunks XOR-swizzled by (row >> 1) & 7) so a chunk is 32 linear 1 KB DMA pieces
// (buffer_load ... lds), the next chunk in flight under the current one's MFMAs, one raw s_barrier per chunk; the weight fragments of
// K step st + 1 are read before the MFMAs of step st.
// Workgroup = 4 waves = 8 pixels x 16 frames = 128 rows, 72 KB of LDS (two weight stages), <= 256 VGPRs: TWO workgroups per CU with
// independent barriers, grid = N HW / 8 (512 workgroups at the ds2 level of the headline batch: one round).  Measured (round 4,
// tools/tattn_bench.py, 65536 rows): 50 us against 107 us for the four launches; the other shapes tried - 256-row workgroups of
// 8 waves with three stages (MMD_TATTN_CFG=1: 49 us there, but 38 against 28 us at 16384 rows) and 4 waves x 64 rows at one wave per
// SIMD (each fragment read feeds two MFMAs, 406 VGPRs: 56 us - nothing overlaps a wave's own VALU phases) - are no faster.
// SQ counters (profiles/r04_tattn_pmc_sq.txt): ~7.6 VALU instructions per MFMA (bias, pack, norm, softmax, epilogue), MFMA pipe busy
// 25 % of the SIMD cycles: the kernel is bound by VALU issue + dependency stalls at two waves per SIMD, not by the matrix pipe.
#include "mmd_common.h"

typedef __attribute__((address_space(3))) void* lptr_t;

#define TA_STAGE_B 32768
#define TA_PLANE_B 8192
#define TA_NCHUNK 16

struct TAttnParams {
  const char* X; int64_t ldx;
  // optional front stage (A != nullptr): the block's input is x = X + pre(A) + bias_pre - the proj_out 1x1 conv + residual of the
  // SPATIAL attention block that precedes the temporal one (unet:485-490) -, written to MID (bf16: what the unfused path stores) and
  // re-read from there as the residual of the last stage
  const char* A; int64_t lda;
  char* MID; int64_t ldm;
  const float* bpre;
  const char* Wf; int wf_bytes;
  const float* bqkv; const float* bproj;
  const float* gamma; const float* beta;
  char* Y; int64_t ldy;
  int N, HW;
  float eps, sc;                       // sc = log2(e) / sqrt(64)
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

// RF = 32-row fragments per wave (1 in every built instance; 2 = the 64-row waves of the measurement above), NW = waves per workgroup
// (rows per workgroup = 32 RF NW), NST = weight stages in LDS.
template <int RF, int NW, int NST, bool PRE>
__global__ __launch_bounds__(64 * NW, (NST == 2 ? 2 : 1)) void tattn_kernel(const TAttnParams p) {
  constexpr int CB = PRE ? 4 : 0;                         // chunks of the front stage
  constexpr int NCH = TA_NCHUNK + CB;
  constexpr int NP = 32 / NW;                             // 1 KB DMA pieces per wave and chunk
  constexpr int PPB = 2 * RF * NW;                        // pixels per workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                                        // [3 stages][4 planes][64 rows][128 B]
  float* sB = (float*)(smem + NST * TA_STAGE_B);          // [q 256 | k 256 | v 256 | proj 256] biases
  float* sG = sB + 1280;                                  // (sB[1024 ..]: bias of the front stage)  [gamma 256 | beta 256]
  float* sR = sG + 512;                                   // RF = 1: [2 parities][4 wave pairs][4 (a, j2)][2 halves][2 quads][2] half-record statistics

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int bpn = p.HW / PPB;
  const int n = blockIdx.x / bpn, pix0 = (blockIdx.x - n * bpn) * PPB;

  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wf, 0, p.wf_bytes, 0x00020000);
  auto issue = [&](int stage, int c) __attribute__((always_inline)) {                    // chunk c -> stage: eight linear 1 KB pieces per wave
#pragma unroll
    for (int i = 0; i < NP; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lptr_t)(sW + stage * TA_STAGE_B + (i * NW + wave) * 1024), 16, lane * 16,
                                               c * TA_STAGE_B + (i * NW + wave) * 1024, 0, 0);
  };
  issue(0, 0);
  if (NST == 3) issue(1, 1);

  // ---- the wave's rows as operand fragments
  int64_t rowi[RF];
  u32x4 xa[RF][16];
#pragma unroll
  for (int f = 0; f < RF; ++f) {
    rowi[f] = ((int64_t)n * 16 + (l31 & 15)) * p.HW + pix0 + 2 * RF * wave + 2 * f + (l31 >> 4);
    const char* ap = PRE ? p.A + (rowi[f] * p.lda + half * 8) * 2 : p.X + (rowi[f] * p.ldx + half * 8) * 2;
#pragma unroll
    for (int cg = 0; cg < 16; ++cg) xa[f][cg] = *(const u32x4*)(ap + cg * 32);
  }
  for (int t = tid; t < 256; t += 64 * NW) {
    const int tid = t;
    const float b0 = p.bqkv[tid], b1 = p.bqkv[256 + tid], b2 = p.bqkv[512 + tid], b3 = p.bproj[tid];
    const float g0 = p.gamma[tid], g1 = p.beta[tid];
    sB[tid] = b0; sB[256 + tid] = b1; sB[512 + tid] = b2; sB[768 + tid] = b3;
    sG[tid] = g0; sG[256 + tid] = g1;
    if (PRE) sB[1024 + tid] = p.bpre[tid];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- chunk pipeline
  const int xsw = (l31 >> 1) & 7;
  const char* fbase = sW + l31 * 128;
  int choff[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) choff[c4] = ((2 * c4 + half) ^ xsw) * 16;

  auto chunk_top = [&](int c) __attribute__((always_inline)) {      // chunk c landed for every wave; the oldest stage is free again
    if (NST == 3 && c <= NCH - 4) {                            // (one younger chunk of NP DMA instructions stays in flight)
      if constexpr (NP == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (c + NST - 1 < NCH) issue((c + NST - 1) % NST, c + NST - 1);
  };
  // acc[a][f] (+)= the chunk's sub-tile a (32 weight rows) against fragment f of B over K = 256.  SWAP: A = rows, B = weights.
  // The weight fragments of K step st + 1 are requested BEFORE the MFMAs of step st (two register sets; sched_barrier pins the
  // order: left alone the compiler issues a step's two reads, waits for them, issues its MFMAs - one LDS latency per step).
  auto gemm = [&](int stage, const u32x4 (&B)[RF][16], f32x16 (&acc)[2][RF], bool swap) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][f][r] = 0.f;
    u32x4 fw[2][2];
    auto ldfw = [&](int buf, int st) __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
        fw[buf][a] = *(const u32x4*)(fbase + stage * TA_STAGE_B + (st >> 2) * TA_PLANE_B + a * 4096 + choff[st & 3]);
    };
    ldfw(0, 0);
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      if (st + 1 < 16) ldfw((st + 1) & 1, st + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int f = 0; f < RF; ++f) {
          if (swap) ta_mma(B[f][st], fw[st & 1][a], acc[a][f]);
          else ta_mma(fw[st & 1][a], B[f][st], acc[a][f]);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  u32x4 oo[RF][16];                                        // the attention output as B operands of the projection: [f][4 h + 2 a + s]
  f32x16 acc[2][RF];
  const char* resid = PRE ? (const char*)p.MID : p.X;      // residual of the last stage
  const int64_t ldres = PRE ? p.ldm : p.ldx;

  // ---- front stage: x = X + pre(A) + bias: four chunks of 64 output channels; the epilogue's 8-consecutive-channel form IS the
  // operand layout of x (k-step 4 j + 2 a + j2), so the rounded result goes to MID and straight into the fragments
  if constexpr (PRE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      chunk_top(j);
      gemm(j % NST, xa, acc, false);
      u32x4 rres[2][2][RF];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
          for (int f = 0; f < RF; ++f)
            rres[a][j2][f] = *(const u32x4*)(p.X + (rowi[f] * p.ldx + 64 * j + 32 * a + 16 * j2 + 8 * half) * 2);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
          const int col = 64 * j + 32 * a + 16 * j2 + 8 * half;
          const f32x4 b0 = *(const f32x4*)(sB + 1024 + col), b1 = *(const f32x4*)(sB + 1024 + col + 4);
#pragma unroll
          for (int f = 0; f < RF; ++f) {
            float v[8];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][f][8 * j2 + jj]), __float_as_uint(acc[a][f][8 * j2 + 4 + jj]),
                                                               false, false);
              v[jj] = __uint_as_float(sw[0]);
              v[4 + jj] = __uint_as_float(sw[1]);
            }
            float rf[8];
            Elt<__bf16>::unpack(rres[a][j2][f], rf);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { v[jj] = (v[jj] + b0[jj]) + rf[jj]; v[4 + jj] = (v[4 + jj] + b1[jj]) + rf[4 + jj]; }
            const u32x4 pk = Elt<__bf16>::pack(v);
            *(u32x4*)(p.MID + (rowi[f] * p.ldm + col) * 2) = pk;
            oo[f][4 * j + 2 * a + j2] = pk;                  // (parked in the registers of the attention output, free until the first head)
          }
        }
    }
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
      for (int cg = 0; cg < 16; ++cg) xa[f][cg] = oo[f][cg];
  }

  // ---- GroupNorm32 over (16 frames, 8 channels) of a pixel, two-pass like mmd_gn_small, applied in place
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int cg = 0; cg < 16; ++cg) {
      float x[8];
      Elt<__bf16>::unpack(xa[f][cg], x);
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
        const f32x4 g4 = *(const f32x4*)(gp + e), b4 = *(const f32x4*)(gp + 256 + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float av = rstd * g4[k];
          const float bv = b4[k] - mean * av;
          x[e + k] = x[e + k] * av + bv;
        }
      }
      u32x4 y = Elt<__bf16>::pack(x);
      asm volatile("" : "+v"(y.x), "+v"(y.y), "+v"(y.z), "+v"(y.w));     // pin: keep the arithmetic here, not sunk into the MFMA loop
      xa[f][cg] = y;
    }

#pragma unroll
  for (int h = 0; h < 4; ++h) {
    u32x4 qo[RF][2][2], ko[RF][2][2], vo[RF][2][2];        // [f][a][s]
    // q and k: lane (row, half) holds channels 8 q + 4 half + j (i = 4 q + j); bias is a vector over i
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int c = CB + 3 * h + t;
      chunk_top(c);
      gemm(c % NST, xa, acc, false);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        f32x4 b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) b4[q] = *(const f32x4*)(sB + t * 256 + h * 64 + a * 32 + 8 * q + 4 * half);
#pragma unroll
        for (int f = 0; f < RF; ++f)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[a][f][8 * s + e] + b4[2 * s + (e >> 2)][e & 3];
            if (t == 0) qo[f][a][s] = Elt<__bf16>::pack(v);
            else ko[f][a][s] = Elt<__bf16>::pack(v);
          }
      }
    }
    // v^T: operands swapped - lane (channel l31, half) holds keys 8 q + 4 half + j; bias is a lane scalar
    {
      const int c = CB + 3 * h + 2;
      chunk_top(c);
      gemm(c % NST, xa, acc, true);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const float bs = sB[512 + h * 64 + a * 32 + l31];
#pragma unroll
        for (int f = 0; f < RF; ++f)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[a][f][8 * s + e] + bs;
            vo[f][a][s] = Elt<__bf16>::pack(v);
          }
      }
    }
    // attention of the head: per fragment, two pixels side by side (block-diagonal P)
    const bool p1 = (l31 >> 4) != 0;
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      f32x16 sT;
#pragma unroll
      for (int r = 0; r < 16; ++r) sT[r] = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int s = 0; s < 2; ++s) ta_mma(ko[f][a][s], qo[f][a][s], sT);
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
      for (int a = 0; a < 2; ++a) {
        f32x16 oT;
#pragma unroll
        for (int r = 0; r < 16; ++r) oT[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          ta_mma(vo[f][a][s], po[s], oT);
          ta_mma(vo[f][a][s], pol[s], oT);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = oT[8 * s + e] * inv;
          oo[f][4 * h + 2 * a + s] = Elt<__bf16>::pack(v);
        }
      }
    }
  }

  // ---- projection: four chunks of 64 output channels; epilogue in the row-strip kernel's form
  // A statistics record = 64 rows = the 16 frames of 4 consecutive pixels: one wave (RF = 2), or a pair of waves (RF = 1: the odd wave
  // parks its half-record in LDS, the even one adds it after the next barrier and writes - a fixed order)
  const int64_t rec = (int64_t)blockIdx.x * (PPB / 4) + (RF == 2 ? wave : wave >> 1);
  float keep[4][2];                                        // RF = 1, even waves: own half-record of the previous chunk, per (a, j2)
  auto flush = [&](int j) __attribute__((always_inline)) { // (RF = 1) after the barrier that follows chunk 12 + j's epilogue
    if (p.stats && (wave & 1) == 0 && (l31 >> 1) == 8) {
#pragma unroll
      for (int aj = 0; aj < 4; ++aj) {
        const float* o = sR + (((((j & 1) * (NW / 2) + (wave >> 1)) * 4 + aj) * 2 + half) * 2 + (l31 & 1)) * 2;
        const int col = 64 * j + 16 * aj + 8 * half;
        float* d = p.stats + (rec * p.stats_ld + (col >> 2) + (l31 & 1)) * 2;
        d[0] = keep[aj][0] + o[0];
        d[1] = keep[aj][1] + o[1];
      }
    }
  };
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = CB + 12 + j;
    chunk_top(c);
    if (RF == 1 && j > 0) flush(j - 1);
    u32x4 rres[2][2][RF];                                  // [a][j2][f]: the residual (RF = 2: requested before the MFMAs; RF = 1 has
                                                           // no registers to park it in - the SIMD's other wave covers the latency)
    auto load_res = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
          for (int f = 0; f < RF; ++f)
            rres[a][j2][f] = *(const u32x4*)(resid + (rowi[f] * ldres + 64 * j + 32 * a + 16 * j2 + 8 * half) * 2);
    };
    if (RF == 2) load_res();
    gemm(c % NST, oo, acc, false);
    if (RF == 1) load_res();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int col = 64 * j + 32 * a + 16 * j2 + 8 * half;
        const f32x4 b0 = *(const f32x4*)(sB + 768 + col), b1 = *(const f32x4*)(sB + 768 + col + 4);
        float u[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < RF; ++f) {
          float v[8];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][f][8 * j2 + jj]), __float_as_uint(acc[a][f][8 * j2 + 4 + jj]),
                                                             false, false);
            v[jj] = __uint_as_float(sw[0]);
            v[4 + jj] = __uint_as_float(sw[1]);
          }
          float rf[8];
          Elt<__bf16>::unpack(rres[a][j2][f], rf);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) { v[jj] = (v[jj] + b0[jj]) + rf[jj]; v[4 + jj] = (v[4 + jj] + b1[jj]) + rf[4 + jj]; }
          const u32x4 pk = Elt<__bf16>::pack(v);
          *(u32x4*)(p.Y + (rowi[f] * p.ldy + col) * 2) = pk;
          if (p.stats) {                                    // statistics of the values as STORED: the lane's 8 channels = two quads
            float sf[8];
            Elt<__bf16>::unpack(pk, sf);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              u[0] += sf[jj];
              u[1] += sf[4 + jj];
              u[2] += sf[jj] * sf[jj];
              u[3] += sf[4 + jj] * sf[4 + jj];
            }
          }
        }
        if (p.stats) {                                      // block-uniform
          const float t0 = halfwave_total(u[0]), t1 = halfwave_total(u[1]), t2 = halfwave_total(u[2]), t3 = halfwave_total(u[3]);
          const float msum = (l31 & 1) ? t1 : t0, msq = (l31 & 1) ? t3 : t2;
          if (RF == 2) {
            if ((l31 >> 1) == 8) {
              float* d = p.stats + (rec * p.stats_ld + (col >> 2) + (l31 & 1)) * 2;
              d[0] = msum;
              d[1] = msq;
            }
          } else {
            keep[a * 2 + j2][0] = msum;
            keep[a * 2 + j2][1] = msq;
            if ((wave & 1) && (l31 >> 1) == 8) {
              float* o = sR + (((((j & 1) * (NW / 2) + (wave >> 1)) * 4 + a * 2 + j2) * 2 + half) * 2 + (l31 & 1)) * 2;
              o[0] = msum;
              o[1] = msq;
            }
          }
        }
      }
  }
  if (RF == 1) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    flush(3);
  }
}

// Weight image of mmd_tattn_block: (4 +) 16 chunks x [4 planes][64 rows][8 chunks of 16 B], chunk pc of a row holds logical chunk
// pc ^ ((row >> 1) & 7).  With a front stage, first: rows 64 j .. of ITS weight [256, 256] (plain K order).  Then chunks 3 h +
// {0, 1, 2}: rows h 64 .. + 64 of the q / k / v third of the qkv weight [768, 256]; then chunks 12 + j: rows 64 j .. of proj_out
// [256, 256] with the 16 K columns of every 16-block in the order 8 (e >> 2) + 4 half + (e & 3).
__global__ __launch_bounds__(256) void tattn_pack_kernel(const uint16_t* __restrict__ Wpre, const uint16_t* __restrict__ Wqkv,
                                                         const uint16_t* __restrict__ Wproj, uint16_t* __restrict__ out, int nchunk) {
  const int o = blockIdx.x * 256 + threadIdx.x;            // one 16-byte chunk
  if (o >= nchunk * 2048) return;
  const int cb = nchunk - TA_NCHUNK;
  const int c = (o >> 11) - cb, r = o & 2047;              // c < 0: front stage
  const int pl = r >> 9, row = (r >> 3) & 63, pc = r & 7;
  const int lc = pc ^ ((row >> 1) & 7);
  const uint16_t* src;
  if (c < 0) src = Wpre + ((int64_t)(64 * (c + cb) + row)) * 256;
  else if (c < 12) src = Wqkv + ((int64_t)((c % 3) * 256 + (c / 3) * 64 + row)) * 256;
  else src = Wproj + ((int64_t)(64 * (c - 12) + row)) * 256;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int col = c < 12 ? 64 * pl + 8 * lc + e : 64 * pl + 16 * (lc >> 1) + 8 * (e >> 2) + 4 * (lc & 1) + (e & 3);
    out[(int64_t)o * 8 + e] = src[col];
  }
}

extern "C" int64_t mmd_tattn_weight_bytes(int with_pre) { return (int64_t)(TA_NCHUNK + (with_pre ? 4 : 0)) * TA_STAGE_B; }

// Wqkv [768, 256] / Wproj [256, 256] / Wpre [256, 256] (nullable: no front stage): bf16, row-major - the 1x1 conv weights of
// SingleModalAtten.qkv / .proj_out of the temporal block (unet:263-266) and .proj_out of the spatial block in front of it
extern "C" int mmd_tattn_pack(const void* Wpre, const void* Wqkv, const void* Wproj, void* out, void* stream) {
  MMD_REQUIRE(Wqkv && Wproj && out, "tattn_pack: null pointer");
  const int nchunk = TA_NCHUNK + (Wpre ? 4 : 0);
  hipLaunchKernelGGL(tattn_pack_kernel, dim3(nchunk * 2048 / 256), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)Wpre,
                     (const uint16_t*)Wqkv, (const uint16_t*)Wproj, (uint16_t*)out, nchunk);
  return mmd_check_launch("tattn_pack");
}

// X / Y: rows (n, f, pixel) x C bf16 (Y may not alias X: other workgroups' residual reads); Wf from mmd_tattn_pack; bias_qkv [768],
// bias_proj / gamma / beta [256] fp32; stats (nullable): quad records of Y, one per 64 rows in THIS kernel's row order inside a sample
// (the 16 frames of 4 consecutive pixels), stats[rec * stats_ld + quad] = (sum, sum of squares).
// Front stage (A != NULL; Wf packed with Wpre): the block's input is x = X + A Wpre^T + bias_pre - the spatial attention block's
// proj_out + residual - and MID [rows, C] receives it (scratch the kernel re-reads; also what the unfused path calls the spatial
// block's output).
extern "C" int mmd_tattn_block(const void* X, int64_t ldx, const void* A, int64_t lda, void* MID, int64_t ldm, const void* Wf,
                               const float* bias_pre, const float* bias_qkv, const float* bias_proj, const float* gamma,
                               const float* beta, float eps, void* Y, int64_t ldy, int N, int F, int HW, int C, int heads, float* stats,
                               int64_t stats_ld, void* stream) {
  MMD_REQUIRE(X && Wf && bias_qkv && bias_proj && gamma && beta && Y, "tattn_block: null pointer");
  MMD_REQUIRE(F == 16 && C == 256 && heads == 4, "tattn_block: built for 16 frames, 256 channels, 4 heads (got F=%d C=%d heads=%d)", F, C, heads);
  MMD_REQUIRE(N > 0 && HW > 0 && HW % 16 == 0, "tattn_block: the pixels of a frame must be a multiple of 16 (N=%d HW=%d)", N, HW);
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
  p.Wf = (const char*)Wf; p.wf_bytes = (TA_NCHUNK + (A ? 4 : 0)) * TA_STAGE_B;
  p.bqkv = bias_qkv; p.bproj = bias_proj; p.gamma = gamma; p.beta = beta;
  p.Y = (char*)Y; p.ldy = ldy; p.N = N; p.HW = HW; p.eps = eps;
  p.sc = 1.44269504088896f * (1.0f / sqrtf(64.f));
  p.stats = stats; p.stats_ld = stats_ld;
  // MMD_TATTN_CFG (read once; A/B - the arithmetic per row is the same): 0 = 128-row workgroups of 4 waves, two per CU, two weight
  // stages (default); 1 = 256-row workgroups of 8 waves, three stages
  static const int cfg = [] { const char* e = getenv("MMD_TATTN_CFG"); return e ? atoi(e) : 0; }();
  constexpr size_t TAB = (1280 + 512 + 256) * sizeof(float);
  const size_t lds3 = 3 * (size_t)TA_STAGE_B + TAB, lds2 = 2 * (size_t)TA_STAGE_B + TAB;
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)tattn_kernel<1, 8, 3, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tattn_kernel<1, 8, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tattn_kernel<1, 4, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)tattn_kernel<1, 4, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "tattn_block: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipStream_t st = (hipStream_t)stream;
  if (cfg == 1) {
    if (A) hipLaunchKernelGGL((tattn_kernel<1, 8, 3, true>), dim3(N * (HW / 16)), dim3(512), lds3, st, p);
    else hipLaunchKernelGGL((tattn_kernel<1, 8, 3, false>), dim3(N * (HW / 16)), dim3(512), lds3, st, p);
  } else {
    if (A) hipLaunchKernelGGL((tattn_kernel<1, 4, 2, true>), dim3(N * (HW / 8)), dim3(256), lds2, st, p);
    else hipLaunchKernelGGL((tattn_kernel<1, 4, 2, false>), dim3(N * (HW / 8)), dim3(256), lds2, st, p);
  }
  return mmd_check_launch("tattn_block");
}
