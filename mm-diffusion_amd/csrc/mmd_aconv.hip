// AudioConv of a ResBlock's in_layers with the GroupNorm32 + SiLU in front of it, in ONE launch (round 6).
//
//   reference: in_layers = normalization(channels), SiLU(), AudioConv(channels, out_channels, 3, dilation = 2^k)   (unet:339-346;
//              AudioConv = Conv1d, kernel 3, padding "same", dilation d: unet:108-131; GroupNorm32: nn.py:16-33)
//   y[m, co] = bias[co] + sum_{tap in 0..2} sum_ci W[co][tap * Cin + ci] * act(x[m + (tap - 1) d, ci] * a[s, ci] + b[s, ci])
//   rows m = (sample s, position l) on channels-last rows; a tap that leaves the sample [0, L) contributes zero (the padding is zero
//   AFTER the norm, as in the reference, where the conv pads the normalised tensor).
//
// Until round 5 this was gn_apply (one read + one write of the tensor, a launch of its own: 51 per step on the audio chain) followed
// by the implicit GEMM: the row-strip kernel at 128 channels (K = 384 fits its registers), the tiled direct-to-LDS loops at 256 - 1024
// input channels (K = 768 .. 3072; 19 - 20 us alone, 44 - 66 us beside the video chain).  The audio levels have FEW rows (102400 ..
// 1600 at batch 4) and a wide K, so the mapping is the row-strip kernel's turned by ninety degrees:
//   * a wave owns 32 rows and ALL accumulators of its workgroup's column range (64 or 128 columns: 32 / 64 registers) for the whole K loop;
//   * K is streamed in steps of one 64-channel chunk of one tap: the wave loads its 32 x 64 activations of the step straight from global
//     memory into MFMA B-operand fragments (the row shifted by (tap - 1) d, clamped + masked outside the sample) one step ahead,
//     applies the fused affine + SiLU in registers (affine rows of the at most two samples of a workgroup in LDS), and multiplies with
//     the step's weight slab [columns][64] that the four waves share through a two-stage LDS ring (global_load_lds, the row-strip
//     kernel's swizzled image);
//   * epilogue from the accumulators as in the row-strip kernel: v_permlane32_swap pairs -> 8 consecutive channels per lane, bias,
//     16-byte stores, quad statistics (sum, sum of squares per 64-row record and channel quad) of the values as stored.
// The normalisation is redone per tap and per column range (3 x 1 .. 8 times): ~8 VALU instructions per element, which makes this a
// VALU-bound kernel - by design: it runs beside the video chain's MFMA-bound launches and no longer round-trips the normalised tensor.
// K order (tap-major, channels ascending, 16-channel MFMA steps) and every rounding point (the normalised value is rounded to bf16 like
// gn_apply's output) equal gn_apply + conv_gemm: the output is bitwise equal to the two launches it replaces.
#include "mmd_common.h"

struct AConvParams {
  const char* X; int64_t ldx;
  const char* W;                       // [Cout][3 * Cin] bf16 (pack_conv_weight layout)
  const float* bias;                   // [Cout]
  const float* ga; const float* gb;    // [S][Cin] fused GroupNorm(+FiLM) affine per sample
  char* Y; int64_t ldy;
  float* stats; int64_t stats_ld;      // optional quad records of Y
  int M, L, Cin, Cout, dil, act;
};

template <int CSUB>                    // 32-column sub-tiles per workgroup: 2 (64 columns) or 4 (128 columns)
__global__ __launch_bounds__(256, 2) void aconv_kernel(const AConvParams p, const int nsplit) {
  constexpr int CS = 32 * CSUB;
  constexpr int STAGE_B = CS * 128;    // one weight slab: CS rows x 64 channels
  constexpr int GP = CS / 32;          // weight DMA instructions per wave and step (8 rows each)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                                         // [2 stages][CS rows][128 B], 16-byte chunks XOR-swizzled by (row >> 1) & 7
  float* sBias = (float*)(smem + 2 * STAGE_B);             // [CS]
  float* sRec = sBias + CS;                                // [4 waves][CSUB * 2][2 halves][2 quads][2] half-record statistics
  float* sAB = sRec + 4 * CSUB * 2 * 2 * 2 * 2;            // [2 samples][a | b][Cin]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  // the nsplit workgroups of one row block get consecutive ids inside one XCD's contiguous range: they share the block's rows in that L2
  const int nwg = gridDim.x;
  int wgid;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int sp = wgid % nsplit, mt = wgid / nsplit;
  const int cbase = sp * CS, m0 = mt * 128;
  const int K = 3 * p.Cin, nck = p.Cin >> 6, nsteps = 3 * nck;

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int lrow = lane >> 3, pc = lane & 7;
  const char* w_ptr[GP];
#pragma unroll
  for (int ih = 0; ih < GP; ++ih) {
    const int row = 8 * (ih * 4 + wave) + lrow;            // row of the slab this lane fetches 16 bytes of
    const int logical = pc ^ ((row >> 1) & 7);
    w_ptr[ih] = p.W + ((int64_t)(cbase + row) * K + logical * 8) * 2;
  }
  auto issue_w = [&](int stage, int s) {                   // step s = (tap, chunk): K offset s * 64 (tap-major: tap * Cin + chunk * 64)
#pragma unroll
    for (int ih = 0; ih < GP; ++ih)
      __builtin_amdgcn_global_load_lds((gptr_t)(w_ptr[ih] + (int64_t)s * 128), (lptr_t)(sW + stage * STAGE_B + (ih * 4 + wave) * 1024), 16, 0, 0);
  };
  // this lane's row, its sample and position; the workgroup's rows touch at most two samples (L >= 128)
  const int row = m0 + wave * 32 + l31;
  const bool rok = row < p.M;
  const int rowc = rok ? row : p.M - 1;
  const int smp = rowc / p.L, pos = rowc - smp * p.L;
  const int s0 = m0 / p.L;
  const int gsel = min(smp - s0, 1) * 2 * p.Cin;
  auto load_x = [&](int s, u32x4 (&x)[4], bool& ok) {
    const int tap = s / nck, ck = s - tap * nck;
    const int sh = (tap - 1) * p.dil;
    ok = (unsigned)(pos + sh) < (unsigned)p.L;
    const int64_t src = ok ? (int64_t)rowc + sh : (int64_t)rowc;       // (a row outside the sample: read the lane's own row, masked below)
    const char* ap = p.X + (src * p.ldx + ck * 64 + half * 8) * 2;
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) x[cg] = *(const u32x4*)(ap + cg * 32);
  };

  issue_w(0, 0);
  u32x4 xc[4], xn[4];
  bool okc, okn = false;
  load_x(0, xc, okc);
  // bias of the column range and the affine rows of the two samples -> LDS
  if (tid < CS) sBias[tid] = p.bias ? p.bias[cbase + tid] : 0.f;
  {
    const int nS = p.M / p.L;
    for (int i = tid; i < 4 * p.Cin; i += 256) {
      const int sl = i / (2 * p.Cin), ab = (i / p.Cin) & 1, c = i % p.Cin;
      const int sidx = min(s0 + sl, nS - 1);
      sAB[i] = (ab ? p.gb : p.ga)[(int64_t)sidx * p.Cin + c];
    }
  }
  f32x16 acc[CSUB];
#pragma unroll
  for (int a = 0; a < CSUB; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                         // slab 0 landed; bias / affine tables visible

  const int xsw = (l31 >> 1) & 7;
  for (int s = 0; s < nsteps; ++s) {
    const int st = s & 1;
    if (s + 1 < nsteps) {                                  // next slab and next activations in flight under this step
      issue_w(st ^ 1, s + 1);
      load_x(s + 1, xn, okn);
    }
    // fused affine + SiLU of this step's 32 rows x 64 channels, rounded to bf16 where gn_apply stores the tensor; zero outside the sample
    const int ck = s % nck;
    u32x4 xb[4];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
      float x[8];
      Elt<__bf16>::unpack(xc[cg], x);
      const float* ap = sAB + gsel + ck * 64 + cg * 16 + half * 8;
      const float* bp = ap + p.Cin;
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        const f32x4 av = *(const f32x4*)(ap + e), bv = *(const f32x4*)(bp + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float y = x[e + k] * av[k] + bv[k];
          x[e + k] = p.act ? silu_f(y) : y;
        }
      }
      u32x4 y = Elt<__bf16>::pack(x);
#pragma unroll
      for (int k = 0; k < 4; ++k) y[k] = okc ? y[k] : 0u;
      xb[cg] = y;
    }
    const char* bW = sW + st * STAGE_B + l31 * 128;
#pragma unroll
    for (int cg = 0; cg < 4; ++cg)
#pragma unroll
      for (int a = 0; a < CSUB; ++a) {
        const u32x4 fw = *(const u32x4*)(bW + a * 32 * 128 + (((2 * cg + half) ^ xsw) * 16));
        acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw), __builtin_bit_cast(bf16x8, xb[cg]), acc[a], 0, 0, 0);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // next slab landed; every wave is past its reads of this stage
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) xc[cg] = xn[cg];
    okc = okn;
  }

  // ---- epilogue: acc[a][4 q + j] = channel 32 a + 8 q + 4 half + j of row l31.  Pair q = 2 j2 (vdst) with q = 2 j2 + 1 (src): afterwards
  //      this lane holds the 8 consecutive channels 32 a + 16 j2 + 8 half .. + 8 of its row
  const bool wave_ok = (int64_t)m0 + wave * 32 < p.M;      // wave-uniform (M % 64 == 0 when statistics are on: wave pairs are whole records)
  const int64_t rec = ((int64_t)m0 + wave * 32) / 64;
  float srec[CSUB][2][2];
#pragma unroll
  for (int a = 0; a < CSUB; ++a)
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const int cb = a * 32 + 16 * j2 + 8 * half;
      const f32x4 b0 = *(const f32x4*)(sBias + cb), b1 = *(const f32x4*)(sBias + cb + 4);
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][8 * j2 + j]), __float_as_uint(acc[a][8 * j2 + 4 + j]), false, false);
        v[j] = __uint_as_float(sw[0]);
        v[4 + j] = __uint_as_float(sw[1]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
      const u32x4 pk = Elt<__bf16>::pack(v);
      if (rok) *(u32x4*)(p.Y + ((int64_t)rowc * p.ldy + cbase + cb) * 2) = pk;
      if (p.stats) {                                       // statistics of the values as STORED: the lane's 8 channels = two quads
        float rf[8], u[4] = {0.f, 0.f, 0.f, 0.f};
        Elt<__bf16>::unpack(pk, rf);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u[0] += rf[j];
          u[1] += rf[4 + j];
          u[2] += rf[j] * rf[j];
          u[3] += rf[4 + j] * rf[4 + j];
        }
        // lanes 16 / 17 of each half end up with the (sum, sum of squares) of quad 0 / 1 of the lane group's 8 channels
        const float t0 = halfwave_total(u[0]), t1 = halfwave_total(u[1]), t2 = halfwave_total(u[2]), t3 = halfwave_total(u[3]);
        const float msum = (l31 & 1) ? t1 : t0, msq = (l31 & 1) ? t3 : t2;
        srec[a][j2][0] = msum;
        srec[a][j2][1] = msq;
        if ((wave & 1) && (l31 >> 1) == 8) {               // a wave holds HALF a record (32 rows): the odd wave parks its half
          float* d = sRec + (((wave * CSUB + a) * 2 + j2) * 2 + half) * 4 + (l31 & 1) * 2;
          d[0] = msum;
          d[1] = msq;
        }
      }
    }
  if (p.stats) {
    __syncthreads();
    if (wave_ok && (wave & 1) == 0 && (l31 >> 1) == 8) {   // the even wave of a pair adds (own + partner) in this fixed order
#pragma unroll
      for (int a = 0; a < CSUB; ++a)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
          const float* o = sRec + ((((wave + 1) * CSUB + a) * 2 + j2) * 2 + half) * 4 + (l31 & 1) * 2;
          const int col = cbase + a * 32 + 16 * j2 + 8 * half;
          float* d = p.stats + (rec * p.stats_ld + (col >> 2) + (l31 & 1)) * 2;
          d[0] = srec[a][j2][0] + o[0];
          d[1] = srec[a][j2][1] + o[1];
        }
    }
  }
}

template <int CSUB>
static int launch_aconv(const AConvParams& p, hipStream_t st) {
  constexpr int CS = 32 * CSUB;
  const int rowblocks = cdiv(p.M, 128), nsplit = p.Cout / CS;
  const size_t lds = 2 * (size_t)CS * 128 + (size_t)CS * 4 + 4 * CSUB * 2 * 2 * 2 * 2 * 4 + 4 * (size_t)p.Cin * 4;
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    const size_t lds_max = 2 * (size_t)CS * 128 + (size_t)CS * 4 + 4 * CSUB * 2 * 2 * 2 * 2 * 4 + 4 * (size_t)2048 * 4;
    hipError_t e = hipFuncSetAttribute((const void*)aconv_kernel<CSUB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "aconv: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL((aconv_kernel<CSUB>), dim3(rowblocks * nsplit), dim3(256), lds, st, p, nsplit);
  return mmd_check_launch("aconv");
}

// GroupNorm32(+FiLM)(+SiLU) -> Conv1d(k = 3, dilation, zero "same" padding) on channels-last rows of S = M / L samples (bf16).
// X [M, ldx] (Cin columns), W [Cout][3 Cin] (mmd_conv_gemm's weight layout, taps -d, 0, +d), ga / gb [S, Cin] = the fused affine of
// mmd_gn_finalize_stats / mmd_gn_stats, Y [M, ldy]; stats (optional): quad records of Y as mmd_conv_gemm_stats writes them (M % 64 == 0).
extern "C" int mmd_aconv(const void* X, int64_t ldx, const void* W, const float* bias, const float* ga, const float* gb, int act, void* Y,
                         int64_t ldy, int M, int L, int Cin, int Cout, int dil, float* stats, int64_t stats_ld, void* stream) {
  MMD_REQUIRE(X && W && ga && gb && Y, "aconv: null pointer");
  MMD_REQUIRE(M > 0 && L >= 128 && M % L == 0, "aconv: M = %d rows must be whole samples of L = %d >= 128 rows", M, L);
  MMD_REQUIRE(Cin >= 64 && Cin % 64 == 0 && Cin <= 2048 && Cout % 64 == 0 && Cout > 0, "aconv: Cin %% 64 == 0 (<= 2048), Cout %% 64 == 0 (got %d -> %d)", Cin, Cout);
  MMD_REQUIRE(dil >= 1, "aconv: dilation %d", dil);
  MMD_REQUIRE(ldx >= Cin && ldy >= Cout && ldx % 8 == 0 && ldy % 8 == 0 && ((uintptr_t)X | (uintptr_t)W | (uintptr_t)Y) % 16 == 0,
              "aconv: rows must be 16-byte aligned");
  MMD_REQUIRE(!stats || (M % 64 == 0 && stats_ld >= Cout / 4), "aconv: statistics need M %% 64 == 0 and a record row of >= Cout / 4 quads");
  const char *x0 = (const char*)X, *x1 = x0 + ((int64_t)(M - 1) * ldx + Cin) * 2, *y0 = (const char*)Y, *y1 = y0 + ((int64_t)(M - 1) * ldy + Cout) * 2;
  MMD_REQUIRE(x1 <= y0 || y1 <= x0, "aconv: X and Y overlap (a tap reads rows other workgroups write)");
  AConvParams p;
  p.X = x0; p.ldx = ldx; p.W = (const char*)W; p.bias = bias; p.ga = ga; p.gb = gb; p.Y = (char*)Y; p.ldy = ldy;
  p.stats = stats; p.stats_ld = stats_ld; p.M = M; p.L = L; p.Cin = Cin; p.Cout = Cout; p.dil = dil; p.act = act;
  // 128 columns per workgroup (each weight fragment feeds more MFMAs, the normalisation is redone for fewer column ranges) when that
  // still gives the chip one workgroup per CU; else 64
  const int rowblocks = cdiv(M, 128);
  if (Cout % 128 == 0 && (int64_t)rowblocks * (Cout / 128) >= 256) return launch_aconv<4>(p, (hipStream_t)stream);
  return launch_aconv<2>(p, (hipStream_t)stream);
}
