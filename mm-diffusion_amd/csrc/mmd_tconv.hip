// The temporal half of VideoConv '2d+1d' - the per-pixel k = 3 conv along the 16 frames
// (/root/reference/mm_diffusion/multimodal_unet.py:83-99: video_conv_temporal, Conv1d over t after the per-frame 3x3 conv) - for the
// levels where the fused 2d+1d kernel does not apply (ds2 / ds4 / ds8: 256 / 384 / 512 channels), bf16.
//
// As an implicit GEMM over rows in (n, f, pixel) order this conv has K = 3 C: the tiled direct-to-LDS loop stages every activation row
// three times (once per tap) and runs at 1.4 TB/s / 550 TFLOP/s on the ds2 shape.  But the conv is LOCAL TO A PIXEL, and with the rows of
// a wave arranged as in the fused temporal-attention block - 32 rows = 2 pixels x 16 frames, row l31 = (pixel l31 >> 4, frame l31 & 15) -
// the 16 frames of a pixel are the 16 lanes of a DPP row: the operand of tap df is the wave's own x fragment shifted by df lanes inside
// the row, zeros shifted in at the ends (v_mov_b32_dpp row_shr:1 / row_shl:1 with bound_ctrl) - exactly the conv's zero padding in time.
// So the activations are stationary in registers (loaded ONCE, like the row-strip GEMM), the weights stream through LDS, and the tap
// shift costs four DPP moves per two MFMAs:
//     acc[a] += W[cb, tap][a] * shift_tap(x)      for tap = 0, 1, 2 (K order = the tiled loops': tap-major, then channel),
// one weight chunk = (column block cb of CC output channels, tap): KS planes x CC rows x 128 B, pre-packed as the swizzled LDS image
// (linear 1 KB DMA pieces), two stages, one raw s_barrier per chunk, fragments of K step st + 1 read before the MFMAs of step st.
// Epilogue from the accumulators (v_permlane32_swap -> 16-byte stores), bias, optional quad statistics records for the GroupNorm that
// consumes y (one record = the 64 rows of a wave pair = the 16 frames of 4 consecutive pixels: the producer's own order inside a sample,
// like mmd_vconv2d1d / mmd_tattn_block).  Same K order, same epilogue arithmetic as mmd_conv_gemm: the output is bitwise equal to it.
// Workgroup = 4 waves = 8 pixels x 16 frames = 128 rows x a range of column blocks (the column split fills the chip at the small
// levels: 4096 rows are 32 row blocks); <= 64 KB of LDS + tables, <= 256 VGPRs: two workgroups per CU.
#include "mmd_common.h"
#include <cstdlib>

typedef __attribute__((address_space(3))) void* lptr_t;

struct TConvParams {
  const char* X; int64_t ldx;
  const char* Wf; int wf_bytes;
  const float* bias;
  char* Y; int64_t ldy;
  int N, HW, Cout;
  int nsplit;                          // column split: workgroup = (row block, one of nsplit ranges of column blocks)
  float* stats; int64_t stats_ld;
};

__device__ __forceinline__ void tc_mma(const u32x4& a, const u32x4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the fragment of the row one frame earlier (CTRL = row_shr:1, 0x111) / later (row_shl:1, 0x101); zeros outside the clip
template <int CTRL>
__device__ __forceinline__ u32x4 tc_shift(const u32x4& v) {
  u32x4 r;
  r.x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, CTRL, 0xf, 0xf, true);
  r.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, CTRL, 0xf, 0xf, true);
  r.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.z, CTRL, 0xf, 0xf, true);
  r.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.w, CTRL, 0xf, 0xf, true);
  return r;
}

// KS = Cin / 64 (planes of 64 input channels), CC = output channels per chunk (64: two 32-row sub-tiles, 32: one)
template <int KS, int CC>
__global__ __launch_bounds__(256, 2) void tconv_kernel(const TConvParams p) {
  constexpr int NW = 4, NA = CC / 32, NK = 4 * KS;
  constexpr int PLANE_B = CC * 128, STAGE_B = KS * PLANE_B;
  constexpr int NP = STAGE_B / 1024 / NW;                 // 1 KB DMA pieces per wave and chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                                        // [2 stages][KS planes][CC rows][128 B]
  float* sB = (float*)(smem + 2 * STAGE_B);               // [Cout <= 512] bias
  float* sR = sB + 512;                                   // [2 parities][2 wave pairs][NA * 2][2 halves][2 quads][2] half-record statistics

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int bpn = p.HW >> 3;
  const int rb = blockIdx.x / p.nsplit, sy = blockIdx.x - rb * p.nsplit;      // the splits of a row block are neighbours: its rows stay in one L2
  const int n = rb / bpn, pix0 = (rb - n * bpn) * 8;
  const int ncb = p.Cout / CC / p.nsplit, cb0 = sy * ncb, nchunk = 3 * ncb;   // this workgroup's column blocks cb0 .. cb0 + ncb

  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wf, 0, p.wf_bytes, 0x00020000);
  auto issue = [&](int stage, int c) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NP; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lptr_t)(sW + stage * STAGE_B + (i * NW + wave) * 1024), 16, lane * 16,
                                               (cb0 * 3 + c) * STAGE_B + (i * NW + wave) * 1024, 0, 0);
  };
  issue(0, 0);

  const int64_t rowi = ((int64_t)n * 16 + (l31 & 15)) * p.HW + pix0 + 2 * wave + (l31 >> 4);
  u32x4 xa[NK];
  {
    const char* ap = p.X + (rowi * p.ldx + half * 8) * 2;
#pragma unroll
    for (int cg = 0; cg < NK; ++cg) xa[cg] = *(const u32x4*)(ap + cg * 32);
  }
  for (int t = tid; t < p.Cout; t += 256) sB[t] = p.bias ? p.bias[t] : 0.f;

  const int xsw = (l31 >> 1) & 7;
  const char* fbase = sW + l31 * 128;
  int choff[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) choff[c4] = ((2 * c4 + half) ^ xsw) * 16;

  const int64_t rec = (int64_t)rb * 2 + (wave >> 1);
  float keep[NA * 2][2];
  auto flush = [&](int cb) __attribute__((always_inline)) {          // after the barrier that follows column block cb's epilogue
    if (p.stats && (wave & 1) == 0 && (l31 >> 1) == 8) {
#pragma unroll
      for (int aj = 0; aj < NA * 2; ++aj) {
        const float* o = sR + (((((cb & 1) * 2 + (wave >> 1)) * (NA * 2) + aj) * 2 + half) * 2 + (l31 & 1)) * 2;
        const int col = (cb0 + cb) * CC + 16 * aj + 8 * half;
        float* d = p.stats + (rec * p.stats_ld + (col >> 2) + (l31 & 1)) * 2;
        d[0] = keep[aj][0] + o[0];
        d[1] = keep[aj][1] + o[1];
      }
    }
  };

  f32x16 acc[NA];
  int c = 0;
  for (int cb = 0; cb < ncb; ++cb) {
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap, ++c) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // chunk c landed (and, at c = 0, x and the bias table)
      __builtin_amdgcn_s_barrier();
      if (c + 1 < nchunk) issue((c + 1) & 1, c + 1);
      if (tap == 0 && cb > 0) flush(cb - 1);
      const char* fb = fbase + (c & 1) * STAGE_B;
      u32x4 fw[2][NA];
      auto ldfw = [&](int buf, int st) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < NA; ++a) fw[buf][a] = *(const u32x4*)(fb + (st >> 2) * PLANE_B + a * 4096 + choff[st & 3]);
      };
      ldfw(0, 0);
#pragma unroll
      for (int st = 0; st < NK; ++st) {
        if (st + 1 < NK) ldfw((st + 1) & 1, st + 1);
        const u32x4 b = tap == 0 ? tc_shift<0x111>(xa[st]) : (tap == 2 ? tc_shift<0x101>(xa[st]) : xa[st]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < NA; ++a) tc_mma(fw[st & 1][a], b, acc[a]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // epilogue of the column block: acc[a][4 q + j] = channel 8 q + 4 half + j of row l31
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int col = (cb0 + cb) * CC + 32 * a + 16 * j2 + 8 * half;
        const f32x4 b0 = *(const f32x4*)(sB + col), b1 = *(const f32x4*)(sB + col + 4);
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[a][8 * j2 + jj]), __float_as_uint(acc[a][8 * j2 + 4 + jj]), false, false);
          v[jj] = __uint_as_float(sw[0]) + b0[jj];
          v[4 + jj] = __uint_as_float(sw[1]) + b1[jj];
        }
        const u32x4 pk = Elt<__bf16>::pack(v);
        *(u32x4*)(p.Y + (rowi * p.ldy + col) * 2) = pk;
        if (p.stats) {                                      // block-uniform: statistics of the values as stored, two quads per lane
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
            float* o = sR + (((((cb & 1) * 2 + (wave >> 1)) * (NA * 2) + a * 2 + j2) * 2 + half) * 2 + (l31 & 1)) * 2;
            o[0] = msum;
            o[1] = msq;
          }
        }
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  flush(ncb - 1);
}

// Weight image of mmd_tconv: chunk (cb, tap) = [KS planes][CC rows][8 chunks of 16 B] (chunk pc of a row holds logical chunk
// pc ^ ((row >> 1) & 7)) from the packed GEMM matrix W [Cout][3 * Cin] (K index = tap * Cin + ci)
__global__ __launch_bounds__(256) void tconv_pack_kernel(const uint16_t* __restrict__ W, uint16_t* __restrict__ out, int Cin, int Cout, int CC) {
  const int KS = Cin / 64;
  const int per_chunk = KS * CC * 8;                       // 16-byte chunks per weight chunk
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= (int64_t)3 * (Cout / CC) * per_chunk) return;
  const int c = (int)(o / per_chunk), r = (int)(o % per_chunk);
  const int cb = c / 3, tap = c % 3;
  const int pl = r / (CC * 8), row = (r / 8) % CC, pc = r & 7;
  const int lc = pc ^ ((row >> 1) & 7);
  const uint16_t* src = W + ((int64_t)(cb * CC + row)) * 3 * Cin + (int64_t)tap * Cin + 64 * pl + 8 * lc;
#pragma unroll
  for (int e = 0; e < 8; ++e) out[o * 8 + e] = src[e];
}

static int tconv_cc(int Cin) { return Cin == 256 ? 64 : 32; }

extern "C" int64_t mmd_tconv_weight_bytes(int Cin, int Cout) { return (int64_t)Cout * 3 * Cin * 2; }

// W: the packed temporal GEMM matrix [Cout][3 * Cin] bf16 (K index = tap * Cin + ci, taps df = -1, 0, +1: what mmd_conv_gemm takes)
extern "C" int mmd_tconv_pack(const void* W, void* out, int Cin, int Cout, void* stream) {
  MMD_REQUIRE(W && out, "tconv_pack: null pointer");
  MMD_REQUIRE((Cin == 256 || Cin == 384 || Cin == 512) && Cout > 0 && Cout % 64 == 0 && Cout <= 512, "tconv_pack: Cin in {256, 384, 512}, Cout %% 64 == 0, <= 512 (got %d -> %d)", Cin, Cout);
  const int64_t chunks16 = (int64_t)Cout * 3 * Cin / 8;
  hipLaunchKernelGGL(tconv_pack_kernel, dim3((unsigned)((chunks16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)W,
                     (uint16_t*)out, Cin, Cout, tconv_cc(Cin));
  return mmd_check_launch("tconv_pack");
}

template <int KS, int CC>
static int launch_tconv(const TConvParams& p, hipStream_t st) {
  const size_t lds = 2 * (size_t)(KS * CC * 128) + (512 + 128) * sizeof(float);
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)tconv_kernel<KS, CC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "tconv: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  // column split (results do not depend on it): the smallest divisor of the column-block count that gives the chip ONE workgroup per CU
  // (two until round 5: see the row-strip GEMM's split, mmd_gemm.hip - the second slot of a CU is the other launch chain's)
  const int rowblocks = p.N * (p.HW / 8), ncb = p.Cout / CC;
  static const int want_blocks = [] {                     // tuning switch (read once): MMD_TCONV_BLOCKS
    const char* e = getenv("MMD_TCONV_BLOCKS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 256;
  }();
  int nsplit = 1;
  for (int d = 1; d <= ncb; ++d)
    if (ncb % d == 0) {
      nsplit = d;
      if ((int64_t)rowblocks * d >= want_blocks) break;
    }
  TConvParams q = p;
  q.nsplit = nsplit;
  hipLaunchKernelGGL((tconv_kernel<KS, CC>), dim3(rowblocks * nsplit), dim3(256), lds, st, q);
  return mmd_check_launch("tconv");
}

// Y[(n, f, pixel), :] = bias + sum over df of X[(n, f + df, pixel), :] W_df^T (zero outside the 16 frames).  X / Y: rows (n, f, pixel),
// bf16, row strides ldx / ldy (Y != X).  stats (nullable): quad records of Y, one per 64 rows in THIS kernel's row order inside a sample
// (the 16 frames of 4 consecutive pixels: record n HW / 4 + (pixel >> 2)).
extern "C" int mmd_tconv(const void* X, int64_t ldx, const void* Wf, const float* bias, void* Y, int64_t ldy, int N, int F, int HW,
                         int Cin, int Cout, float* stats, int64_t stats_ld, void* stream) {
  MMD_REQUIRE(X && Wf && Y, "tconv: null pointer");
  MMD_REQUIRE(F == 16 && (Cin == 256 || Cin == 384 || Cin == 512) && Cout > 0 && Cout % 64 == 0 && Cout <= 512,
              "tconv: built for 16 frames, Cin in {256, 384, 512}, Cout %% 64 == 0, <= 512 (got F=%d Cin=%d Cout=%d)", F, Cin, Cout);
  MMD_REQUIRE(N > 0 && HW > 0 && HW % 8 == 0, "tconv: the pixels of a frame must be a multiple of 8 (N=%d HW=%d)", N, HW);
  MMD_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= Cin && ldy >= Cout && ((uintptr_t)X | (uintptr_t)Y | (uintptr_t)Wf) % 16 == 0 && X != Y,
              "tconv: 16-byte aligned rows, Y != X");
  MMD_REQUIRE(!stats || (stats_ld >= Cout / 4 && (uintptr_t)stats % 8 == 0), "tconv: statistics buffer");
  TConvParams p;
  p.X = (const char*)X; p.ldx = ldx; p.Wf = (const char*)Wf; p.wf_bytes = Cout * 3 * Cin * 2; p.bias = bias;
  p.Y = (char*)Y; p.ldy = ldy; p.N = N; p.HW = HW; p.Cout = Cout; p.nsplit = 1; p.stats = stats; p.stats_ld = stats_ld;
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 256) return launch_tconv<4, 64>(p, st);
  if (Cin == 384) return launch_tconv<6, 32>(p, st);
  return launch_tconv<8, 32>(p, st);
}
