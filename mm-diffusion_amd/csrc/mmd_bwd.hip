// Backward kernels of the training step (multimodal_training_losses backward, reference gd:1114-1203 through the U-Net).
//
//   conv wgrad   dW[co, tap*Cin + ci] += sum_m dY[m, co] * X[src(m, tap), ci]        (implicit GEMM, reduction over rows)
//   conv dgrad   = mmd_conv_gemm on dY with the transposed weight and mirrored taps   (no new kernel)
//   colsum       db[c] += sum_m dY[m, c]
//   GroupNorm    backward of y = act((x - mu) rstd gamma (1 + scale) ... ) in the fused-affine form of mmd_norm.hip
//   elementwise  SiLU forward/backward, d(mse)/d(out), AdamW (+EMA)
// Accumulation into dW / db uses fp32 L2 atomics over row splits (like the vendor conv backward, run-to-run bit
// differences of the last ulp are possible); everything else is deterministic.
#include <cstdlib>

#include "mmd_common.h"

// ============================================================================= conv wgrad (implicit GEMM-TN on MFMA)
struct WgradParams {
  const char* dY; int64_t lddy;      // [M, Cout]
  const char* X; int64_t ldx;        // [rows, Cin]
  float* dW;                         // fp32, accumulated with atomics: [Cout][ntaps*Cin] (packed) or [Cout][Cin][ntaps] (torch conv layout)
  float* db;                         // wgrad128 only: bias gradient (column sums of dY) accumulated by the (tap 0, ci tile 0) blocks
  int torch_layout;
  int M, Cout, Cin, ntaps;
  int D0, D1, D2;
  int rows_per_split, splits, xcd_order;
  int taps[27 * 3];
};

// Block = 64 (co) x 64 (ci of ONE tap) output tile x one row split; 4 waves, each a 32x32 MFMA tile.
// The reduction index is the row m, so both operands are needed "column-wise" (8 consecutive rows of one channel):
// tiles are staged row-major in LDS (coalesced 16-B loads) and the fragments are gathered with 2-byte LDS reads
// (bf16) / 4-byte reads (fp32), which are conflict free because adjacent lanes read adjacent channels.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  constexpr int MT = 64;                       // rows per staged chunk
  constexpr int LDT = 64 * ES + 16;            // bytes per staged row (64 channels + pad)
  __shared__ __attribute__((aligned(16))) char sdy[MT * LDT];
  __shared__ __attribute__((aligned(16))) char sx[MT * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave & 1, wj = wave >> 1;     // wave tile: co 32*wi, ci 32*wj
  const int half = lane >> 5, l31 = lane & 31;
  const int ci_tiles = (p.Cin + 63) / 64;
  const int kt = blockIdx.y;                   // (tap, ci tile)
  const int tap = kt / ci_tiles, ci0 = (kt % ci_tiles) * 64;
  const int co0 = blockIdx.x * 64;
  const int o0 = p.taps[tap * 3], o1 = p.taps[tap * 3 + 1], o2 = p.taps[tap * 3 + 2];
  const int D12 = p.D1 * p.D2;
  const int64_t roff = (int64_t)o0 * D12 + o1 * p.D2 + o2;
  const int m_begin = blockIdx.z * p.rows_per_split;
  const int m_end = min(m_begin + p.rows_per_split, p.M);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  constexpr int VPR = 64 / EPV;                // 16-byte vecs per staged row
  for (int mc = m_begin; mc < m_end; mc += MT) {
    __syncthreads();
    for (int i = tid; i < MT * VPR; i += 256) {
      const int r = i / VPR, v = i % VPR;
      const int m = mc + r;
      u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
      if (m < m_end) {
        if (co0 + v * EPV < p.Cout) a = *(const u32x4*)(p.dY + ((int64_t)m * p.lddy + co0 + v * EPV) * ES);
        const int p2 = m % p.D2, p1 = (m / p.D2) % p.D1, p0 = (m / D12) % p.D0;
        if (ci0 + v * EPV < p.Cin && (unsigned)(p0 + o0) < (unsigned)p.D0 && (unsigned)(p1 + o1) < (unsigned)p.D1 &&
            (unsigned)(p2 + o2) < (unsigned)p.D2)
          b = *(const u32x4*)(p.X + (((int64_t)m + roff) * p.ldx + ci0 + v * EPV) * ES);
      }
      *(u32x4*)(sdy + r * LDT + v * 16) = a;
      *(u32x4*)(sx + r * LDT + v * 16) = b;
    }
    __syncthreads();
    if constexpr (EPV == 8) {
#pragma unroll
      for (int ks = 0; ks < MT / 16; ++ks) {        // 16 rows per MFMA: lane (channel l31, half) takes rows ks*16 + 8*half + j
        bf16x8 fa, fb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int r = ks * 16 + 8 * half + j;
          fa[j] = *(const __bf16*)(sdy + r * LDT + (wi * 32 + l31) * 2);
          fb[j] = *(const __bf16*)(sx + r * LDT + (wj * 32 + l31) * 2);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
      }
    } else {
#pragma unroll 8
      for (int ks = 0; ks < MT / 2; ++ks) {         // 2 rows per MFMA: lane half takes row 2*ks + half
        const int r = ks * 2 + half;
        const float fa = *(const float*)(sdy + r * LDT + (wi * 32 + l31) * 4);
        const float fb = *(const float*)(sx + r * LDT + (wj * 32 + l31) * 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc, 0, 0, 0);
      }
    }
  }
  // D[row = co][col = ci]: lane holds ci = l31, co = (r&3) + 8*(r>>2) + 4*half
  const int64_t K = (int64_t)p.Cin * p.ntaps;
  const int ci = ci0 + wj * 32 + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (co < p.Cout && ci < p.Cin)
      atomicAdd(p.dW + (int64_t)co * K + (p.torch_layout ? (int64_t)ci * p.ntaps + tap : (int64_t)tap * p.Cin + ci), acc[r]);
  }
}

// bf16 weight gradient, 128 (co) x 128 (ci of ONE tap) tile per block, 4 waves each 64 x 64 (2 x 2 MFMA tiles).
// The reduction index is the row m, so both MFMA operands need 8 CONSECUTIVE ROWS of one channel per lane.  The 64-tile kernel
// above stages row-major and gathers every fragment with eight 2-byte LDS reads (16 reads per MFMA); here the 64-row chunk is
// transposed while it is stored to LDS ([channel][64 rows], 136-byte pitch: the 32-rows x 2-vectors lane pattern of the
// attention V^T staging, conflict free) and a fragment is two ds_read_b64 - 4 LDS reads per 4 MFMAs.  The next chunk's global
// loads are in flight under the MFMAs (register prefetch).
#define WG_PITCH 136
__global__ __launch_bounds__(256, 2) void wgrad128_bf16_kernel(const WgradParams p) {
  __shared__ __attribute__((aligned(16))) char sA[128 * WG_PITCH];      // dY^T chunk: [co][m]
  __shared__ __attribute__((aligned(16))) char sB[128 * WG_PITCH];      // X^T  chunk: [ci][m]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave & 1, wj = wave >> 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int ci_tiles = (p.Cin + 127) / 128;
  const int kt = blockIdx.y;
  const int tap = kt / ci_tiles, ci0 = (kt % ci_tiles) * 128;
  const int co0 = blockIdx.x * 128;
  const int o0 = p.taps[tap * 3], o1 = p.taps[tap * 3 + 1], o2 = p.taps[tap * 3 + 2];
  const int D12 = p.D1 * p.D2;
  const int64_t roff = (int64_t)o0 * D12 + o1 * p.D2 + o2;
  const int m_begin = blockIdx.z * p.rows_per_split;
  const int m_end = min(m_begin + p.rows_per_split, p.M);

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // staging slots of this thread: 4 x (row j, 8-channel vector v) per operand
  int sj[4], sv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    sj[i] = (id & 31) + 32 * ((id >> 6) & 1);
    sv[i] = 2 * (id >> 7) + ((id >> 5) & 1);
  }
  u32x4 ra[4], rb[4];
  auto load_chunk = [&](int mc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mc + sj[i];
      u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
      if (m < m_end) {
        if (co0 + sv[i] * 8 < p.Cout) a = *(const u32x4*)(p.dY + ((int64_t)m * p.lddy + co0 + sv[i] * 8) * 2);
        const int p2 = m % p.D2, p1 = (m / p.D2) % p.D1, p0 = (m / D12) % p.D0;
        if (ci0 + sv[i] * 8 < p.Cin && (unsigned)(p0 + o0) < (unsigned)p.D0 && (unsigned)(p1 + o1) < (unsigned)p.D1 &&
            (unsigned)(p2 + o2) < (unsigned)p.D2)
          b = *(const u32x4*)(p.X + (((int64_t)m + roff) * p.ldx + ci0 + sv[i] * 8) * 2);
      }
      ra[i] = a;
      rb[i] = b;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint16_t* da = (uint16_t*)(sA + (8 * sv[i]) * WG_PITCH + 2 * sj[i]);
      uint16_t* db = (uint16_t*)(sB + (8 * sv[i]) * WG_PITCH + 2 * sj[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        da[e * (WG_PITCH / 2)] = (uint16_t)((ra[i][e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        db[e * (WG_PITCH / 2)] = (uint16_t)((rb[i][e >> 1] >> ((e & 1) * 16)) & 0xffffu);
      }
    }
  };

  const bool do_db = p.db != nullptr && kt == 0;
  float dbacc = 0.f;
  load_chunk(m_begin);
  for (int mc = m_begin; mc < m_end; mc += 64) {
    __syncthreads();                 // previous chunk fully consumed
    store_chunk();
    __syncthreads();
    if (mc + 64 < m_end) load_chunk(mc + 64);
    if (do_db) {                     // bias gradient from the staged dY^T tile: thread = (channel, 32-row half); rows past m_end are zero
      const char* q = sA + (tid >> 1) * WG_PITCH + (tid & 1) * 64;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const u32x2 v = *(const u32x2*)(q + 8 * u);
        dbacc += __uint_as_float(v[0] << 16) + __uint_as_float(v[0] & 0xffff0000u) + __uint_as_float(v[1] << 16) + __uint_as_float(v[1] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      u32x4 fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const char* q = sA + (wi * 64 + a * 32 + l31) * WG_PITCH + (16 * ks + 8 * half) * 2;
        const u32x2 v0 = *(const u32x2*)q, v1 = *(const u32x2*)(q + 8);
        fa[a] = u32x4{v0[0], v0[1], v1[0], v1[1]};
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const char* q = sB + (wj * 64 + b * 32 + l31) * WG_PITCH + (16 * ks + 8 * half) * 2;
        const u32x2 v0 = *(const u32x2*)q, v1 = *(const u32x2*)(q + 8);
        fb[b] = u32x4{v0[0], v0[1], v1[0], v1[1]};
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[a]), __builtin_bit_cast(bf16x8, fb[b]), acc[a][b], 0, 0, 0);
    }
  }
  if (do_db) {
    dbacc += __shfl_xor(dbacc, 1, 64);
    const int co = co0 + (tid >> 1);
    if ((tid & 1) == 0 && co < p.Cout) atomicAdd(p.db + co, dbacc);
  }
  // D[row = co][col = ci]: lane holds ci = l31, co = (r&3) + 8*(r>>2) + 4*half of its 32x32 tile
  const int64_t K = (int64_t)p.Cin * p.ntaps;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ci = ci0 + wj * 64 + b * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wi * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < p.Cout && ci < p.Cin)
          atomicAdd(p.dW + (int64_t)co * K + (p.torch_layout ? (int64_t)ci * p.ntaps + tap : (int64_t)tap * p.Cin + ci), acc[a][b][r]);
      }
    }
}


// bf16 weight gradient, round 3: the same 128 (co) x 128 (ci of ONE tap) tile per block, 4 waves each 64 x 64, but nothing is transposed
// by the kernel.  The reduction index is the row m, so both MFMA operands want 8 consecutive ROWS of one channel per lane - exactly what
// ds_read_b64_tr_b16 returns from a ROW-MAJOR tile.  A 64-row chunk of dY and of X therefore goes global -> LDS by buffer-descriptor DMA
// (no VGPR staging, no ds_write; a padding row or a row past the split is an out-of-range offset and lands as zeros), in the tile format
// of the attention kernel's V ([plane of 64 channels][64 rows][128 B], 16-byte chunk ^ (((row >> 1) & 1) << 2)), double buffered with one
// barrier per chunk; a fragment is two transposing reads.  wgrad128_bf16_kernel above spent 64 ds_write_b16 + 12 integer divisions per
// thread and chunk on what the DMA and two float reciprocals do here.  The column sums (bias gradient) ride in the (tap 0, ci tile 0) blocks (round 6).
__device__ __forceinline__ int wg_fdiv(int n, int d, float rd) {       // n / d for 0 <= n < 2^24 (the launcher checks M)
  int q = (int)((float)n * rd);
  const int r = n - q * d;
  q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
  return q;
}

__global__ __launch_bounds__(256, 2) void wgrad_tr_bf16_kernel(const WgradParams p) {
  constexpr int PLANE_B = 64 * 128, OP_B = 2 * PLANE_B, STAGE_B = 2 * OP_B;
  extern __shared__ __attribute__((aligned(16))) char smem_w[];        // [2 stages][dY | X][2 planes][64 rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave & 1, wj = wave >> 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int ci_tiles = (p.Cin + 127) / 128;
  // round 6: XCD-aware block order.  The (co tile, ci tile, tap) blocks of one row split read the SAME rows of dY (per co tile) and of X
  // (per ci tile, shifted by the tap): 9 - 27 blocks per split for the 3 x 3 convs.  Dispatched in grid order they land on all eight XCDs
  // (workgroup id % 8) and every L2 fetches its own copy.  The grid is one-dimensional and padded to 8 x (blocks per split) x ceil(splits / 8):
  // XCD k works through the splits k, k + 8, ..., the blocks of a split back to back on one L2 (1 - 3 MB of rows per split).
  int kt_blk, co_blk, split;
  {
    const int nx = (p.Cout + 127) / 128, per = nx * ci_tiles * p.ntaps;
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3;
    split = p.xcd_order ? xcd + 8 * (j / per) : bid / per;
    const int inner = p.xcd_order ? j % per : bid % per;
    co_blk = inner % nx;
    kt_blk = inner / nx;
    if (split >= p.splits) return;                         // padding blocks (before any barrier)
  }
  const int tap = kt_blk / ci_tiles, ci0 = (kt_blk % ci_tiles) * 128;
  const int co0 = co_blk * 128;
  const int o0 = p.taps[tap * 3], o1 = p.taps[tap * 3 + 1], o2 = p.taps[tap * 3 + 2];
  const int D12 = p.D1 * p.D2;
  const int roff = o0 * D12 + o1 * p.D2 + o2;
  const bool shifted = (o0 | o1 | o2) != 0;
  const float rD2 = 1.f / (float)p.D2, rD1 = 1.f / (float)p.D1, rD0 = 1.f / (float)p.D0;
  const int m_begin = split * p.rows_per_split;
  const int m_end = min(m_begin + p.rows_per_split, p.M);

  typedef __attribute__((address_space(3))) void* lptr_t;
  const auto rsrcY = __builtin_amdgcn_make_buffer_rsrc((void*)p.dY, 0, (int)((int64_t)p.M * p.lddy * 2), 0x00020000);
  const auto rsrcX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (int)((int64_t)p.M * p.ldx * 2), 0x00020000);
  // DMA: wave w stages row groups {w, w + 4} of every plane; lane L of a group covers row 8 g + L / 8, physical chunk L % 8
  const int lrow = lane >> 3, pc = lane & 7;
  int drow[2];
  uint32_t ycol[2][2], xcol[2][2];            // [row group][plane]: byte column of the lane's (logical) chunk, or ~0 past the channel count
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (wave + 4 * i) + lrow;
    drow[i] = row;
    const int lc = pc ^ (((row >> 1) & 1) << 2);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const int co = co0 + 64 * pl + 8 * lc, ci = ci0 + 64 * pl + 8 * lc;
      ycol[i][pl] = co < p.Cout ? (uint32_t)(co * 2) : 0xffffffffu;
      xcol[i][pl] = ci < p.Cin ? (uint32_t)(ci * 2) : 0xffffffffu;
    }
  }
  const uint32_t ldyb = (uint32_t)(p.lddy * 2), ldxb = (uint32_t)(p.ldx * 2);
  auto issue = [&](int stage, int mc) {
    char* sb = smem_w + stage * STAGE_B;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mc + drow[i];
      const bool okm = m < m_end;
      bool okx = okm;
      if (shifted) {
        const int q1 = wg_fdiv(m, p.D2, rD2), p2 = m - q1 * p.D2;
        const int q2 = wg_fdiv(q1, p.D1, rD1), p1 = q1 - q2 * p.D1;
        const int q3 = wg_fdiv(q2, p.D0, rD0), p0 = q2 - q3 * p.D0;
        okx = okm && (unsigned)(p0 + o0) < (unsigned)p.D0 && (unsigned)(p1 + o1) < (unsigned)p.D1 && (unsigned)(p2 + o2) < (unsigned)p.D2;
      }
      const uint32_t yrow = (uint32_t)m * ldyb, xrow = (uint32_t)(m + roff) * ldxb;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const uint32_t yo = okm && ycol[i][pl] != 0xffffffffu ? yrow + ycol[i][pl] : 0xfffffff0u;
        const uint32_t xo = okx && xcol[i][pl] != 0xffffffffu ? xrow + xcol[i][pl] : 0xfffffff0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (lptr_t)(sb + pl * PLANE_B + (wave + 4 * i) * 1024), 16, yo, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lptr_t)(sb + OP_B + pl * PLANE_B + (wave + 4 * i) * 1024), 16, xo, 0, 0, 0);
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // transposing reads: this lane supplies row base + 4 half + (lane & 15) / 4, byte column 32 ((lane >> 4) & 1) + 8 (lane & 3) of a 64-byte
  // channel tile, and receives the four rows base + 4 half + [0, 4) of channel l31 of that tile
  const int vrow0 = 4 * half + ((lane & 15) >> 2);
  const int vcolb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  typedef __attribute__((address_space(3))) s16x4* lp4;
  auto frag = [&](const char* plane, int rbase, int ct) -> bf16x8 {
    s16x4 lo, hi;
    {
      const int row = rbase + vrow0;
      const int ch = ((vcolb >> 4) + 4 * ct) ^ (((row >> 1) & 1) << 2);
      lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(plane + row * 128 + ch * 16 + (vcolb & 15)));
    }
    {
      const int row = rbase + 8 + vrow0;
      const int ch = ((vcolb >> 4) + 4 * ct) ^ (((row >> 1) & 1) << 2);
      hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(plane + row * 128 + ch * 16 + (vcolb & 15)));
    }
    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, both);
  };

  // bias gradient (round 6: db != NULL): the (tap 0, ci tile 0) blocks sum the columns of the dY stage they have in LDS anyway - thread =
  // (channel tid & 127, row half tid >> 7): a wave reads 128 contiguous bytes of a row per step (the chunk swizzle permutes 16-byte pieces
  // inside them), rows past the split are zero-filled by the DMA.  No colsum launch: the 1x1 / k = 3 convs can use this kernel too.
  const bool do_db = p.db != nullptr && kt_blk == 0;
  float dbacc = 0.f;
  const int dbc = tid & 127, dbh = tid >> 7;
  int stage = 0;
  if (m_begin < m_end) issue(0, m_begin);
  for (int mc = m_begin; mc < m_end; mc += 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this chunk's DMA (the only one in flight) has landed
    __builtin_amdgcn_s_barrier();                           // ... for every wave; everyone is past its reads of the other stage
    asm volatile("" ::: "memory");
    if (mc + 64 < m_end) issue(stage ^ 1, mc + 64);         // in flight under this chunk's MFMAs
    const char* pa = smem_w + stage * STAGE_B + wi * PLANE_B;
    const char* pb = smem_w + stage * STAGE_B + OP_B + wj * PLANE_B;
    if (do_db) {                                            // block-uniform
      const char* q = smem_w + stage * STAGE_B + (dbc >> 6) * PLANE_B + (dbc & 7) * 2;
      const int lc = (dbc & 63) >> 3;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const int row = 32 * dbh + r;
        const uint16_t v = *(const uint16_t*)(q + row * 128 + ((lc ^ (((row >> 1) & 1) << 2)) * 16));
        dbacc += __uint_as_float((uint32_t)v << 16);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                        // 16 rows per k-step
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = frag(pa, 16 * ks, a);
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = frag(pb, 16 * ks, b);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    stage ^= 1;
  }
  if (do_db && co0 + dbc < p.Cout) atomicAdd(p.db + co0 + dbc, dbacc);
  // D[row = co][col = ci]: lane holds ci = l31, co = (r&3) + 8*(r>>2) + 4*half of its 32x32 tile
  const int64_t K = (int64_t)p.Cin * p.ntaps;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ci = ci0 + wj * 64 + b * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wi * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < p.Cout && ci < p.Cin)
          atomicAdd(p.dW + (int64_t)co * K + (p.torch_layout ? (int64_t)ci * p.ntaps + tap : (int64_t)tap * p.Cin + ci), acc[a][b][r]);
      }
    }
}

// db[c] += sum over rows of dY[m, c]
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const char* __restrict__ dy, int64_t ld, int M, int C, float* __restrict__ out,
                                                     int rows_per_block) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  __shared__ float s_sum[256 * EPV];
  const int tid = threadIdx.x;
  const int CV = C / EPV, RPP = 256 / CV;
  const int col = tid % CV, rl = tid / CV;
  float sum[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) sum[e] = 0.f;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(m0 + rows_per_block, M);
  if (rl < RPP) {
    int m = m0 + rl;
    for (; m + 3 * RPP < m1; m += 4 * RPP) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const u32x4*)(dy + ((int64_t)(m + u * RPP) * ld + col * EPV) * ES);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float f[EPV];
        Elt<T>::unpack(v[u], f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) sum[e] += f[e];
      }
    }
    for (; m < m1; m += RPP) {
      float f[EPV];
      Elt<T>::unpack(*(const u32x4*)(dy + ((int64_t)m * ld + col * EPV) * ES), f);
#pragma unroll
      for (int e = 0; e < EPV; ++e) sum[e] += f[e];
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) s_sum[rl * C + col * EPV + e] = sum[e];
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int r = 0; r < RPP; ++r) a += s_sum[r * C + c];
    atomicAdd(out + c, a);
  }
}

// ============================================================================= GroupNorm backward
// Forward (mmd_norm.hip): v = x*a[s,c] + b[s,c], y = act(v) with a = rstd*gamma*(1+sc), b = (beta - mu*rstd*gamma)(1+sc) + sh.
// With z = (x - mu) rstd:   P[s,c] = sum_rows dv,  Q[s,c] = sum_rows dv*z   (dv = dy * act'(v))
//   dshift = P, dscale = gamma*Q + beta*P, dbeta += (1+sc) P, dgamma += (1+sc) Q
//   dx = rstd * ( (1+sc) gamma dv - mean_g[(1+sc) gamma P]/... )   -> per (slice, group) m1 = sum_c g_c P_c / cnt, m2 = sum_c g_c Q_c / cnt
//   dx = rstd * (g_c dv - m1 - z m2),  g_c = (1+sc_c) gamma_c
struct GnBwdGeom {
  int S, Tn, inner;
  int64_t outer_stride, inner_stride, tstride;
};
__device__ __forceinline__ int64_t gnb_base(const GnBwdGeom& g, int s) {
  return (int64_t)(s / g.inner) * g.outer_stride + (int64_t)(s % g.inner) * g.inner_stride;
}

// stage 1: per (slice, row chunk) partial P, Q per channel -> atomics into PQ[S][C][2] (fp32)
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const char* __restrict__ x, int64_t ldx, const char* __restrict__ dy, int64_t lddy,
                                                            int C, GnBwdGeom g, const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ mr, int act, int R, float* __restrict__ PQ) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  __shared__ float s_p[256 * EPV], s_q[256 * EPV];
  const int s = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int CV = C / EPV, RPP = 256 / CV;
  const int col = tid % CV, rl = tid / CV;
  const int cpg = C / 32;
  const int64_t base = gnb_base(g, s);
  float P[EPV], Q[EPV], av[EPV], bv[EPV], mu[EPV], rs[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    const int c = col * EPV + e;
    P[e] = Q[e] = 0.f;
    av[e] = a[(int64_t)s * C + c];
    bv[e] = b[(int64_t)s * C + c];
    mu[e] = mr[((int64_t)s * 32 + c / cpg) * 2];
    rs[e] = mr[((int64_t)s * 32 + c / cpg) * 2 + 1];
  }
  const int j0 = chunk * R, j1 = min(j0 + R, g.Tn);
  if (rl < RPP) {
    auto body = [&](const u32x4& vx, const u32x4& vd) {
      float fx[EPV], fd[EPV];
      Elt<T>::unpack(vx, fx);
      Elt<T>::unpack(vd, fd);
#pragma unroll
      for (int e = 0; e < EPV; ++e) {
        float dv = fd[e];
        if (act) {
          const float v = fx[e] * av[e] + bv[e];
          const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-v));      // v_rcp_f32 (1 ulp), like silu_f of the forward: the IEEE division is ~10 more VALU instructions per element
          dv *= sg * (1.f + v * (1.f - sg));
        }
        P[e] += dv;
        Q[e] += dv * (fx[e] - mu[e]) * rs[e];
      }
    };
    int j = j0 + rl;
    for (; j + RPP < j1; j += 2 * RPP) {
      const int64_t r0 = base + (int64_t)j * g.tstride, r1 = base + (int64_t)(j + RPP) * g.tstride;
      const u32x4 x0 = *(const u32x4*)(x + (r0 * ldx + col * EPV) * ES);
      const u32x4 d0 = *(const u32x4*)(dy + (r0 * lddy + col * EPV) * ES);
      const u32x4 x1 = *(const u32x4*)(x + (r1 * ldx + col * EPV) * ES);
      const u32x4 d1 = *(const u32x4*)(dy + (r1 * lddy + col * EPV) * ES);
      body(x0, d0);
      body(x1, d1);
    }
    if (j < j1) {
      const int64_t r0 = base + (int64_t)j * g.tstride;
      body(*(const u32x4*)(x + (r0 * ldx + col * EPV) * ES), *(const u32x4*)(dy + (r0 * lddy + col * EPV) * ES));
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) { s_p[rl * C + col * EPV + e] = P[e]; s_q[rl * C + col * EPV + e] = Q[e]; }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float pa = 0.f, qa = 0.f;
    for (int r = 0; r < RPP; ++r) { pa += s_p[r * C + c]; qa += s_q[r * C + c]; }
    atomicAdd(PQ + ((int64_t)s * C + c) * 2, pa);
    atomicAdd(PQ + ((int64_t)s * C + c) * 2 + 1, qa);
  }
}

// stage 2: parameter / FiLM gradients and the per-(slice, group) means m1, m2   (one block per slice)
__global__ __launch_bounds__(256) void gn_bwd_params_kernel(float* __restrict__ PQ, int C, int Tn, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ film, int64_t film_ld,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dfilm,
                                                            int64_t dfilm_ld, float* __restrict__ m12, int rezero) {
  // round 6: grid (4, S) - a block owns 8 groups = 8 cpg <= 256 channels, one per thread (it was one block per slice looping over C)
  __shared__ float s_g1[256], s_g2[256];
  const int s = blockIdx.y, g0 = blockIdx.x * 8, tid = threadIdx.x;
  const int cpg = C / 32, nch = 8 * cpg;
  if (tid < nch) {
    const int c = g0 * cpg + tid;
    const float P = PQ[((int64_t)s * C + c) * 2], Q = PQ[((int64_t)s * C + c) * 2 + 1];
    if (rezero) {                        // mmd_gn_bwd_ws0: the accumulators go back to zero for the next call on this workspace
      PQ[((int64_t)s * C + c) * 2] = 0.f;
      PQ[((int64_t)s * C + c) * 2 + 1] = 0.f;
    }
    const float sc1 = film ? 1.f + film[(int64_t)s * film_ld + c] : 1.f;
    const float gc = sc1 * gamma[c];
    s_g1[tid] = gc * P;
    s_g2[tid] = gc * Q;
    atomicAdd(dgamma + c, sc1 * Q);
    atomicAdd(dbeta + c, sc1 * P);
    if (dfilm) {
      dfilm[(int64_t)s * dfilm_ld + c] = gamma[c] * Q + beta[c] * P;        // d scale
      dfilm[(int64_t)s * dfilm_ld + C + c] = P;                             // d shift
    }
  }
  __syncthreads();
  if (tid < 8) {
    float a1 = 0.f, a2 = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a1 += s_g1[c]; a2 += s_g2[c]; }
    const float cnt = (float)Tn * (float)cpg;
    m12[((int64_t)s * 32 + g0 + tid) * 2] = a1 / cnt;
    m12[((int64_t)s * 32 + g0 + tid) * 2 + 1] = a2 / cnt;
  }
}

// stage 3: dx = rstd * (g_c dv - m1 - z m2) = k1 dv + k2 x + k3 with per-(slice, channel) coefficients held in registers;
// block = (row chunk, slice), thread = (channel vector, row lane), two rows in flight.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const char* __restrict__ x, int64_t ldx, const char* __restrict__ dy, int64_t lddy,
                                                           char* __restrict__ dx, int64_t lddx, int C, GnBwdGeom g,
                                                           const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ mr,
                                                           const float* __restrict__ m12, const float* __restrict__ gamma,
                                                           const float* __restrict__ film, int64_t film_ld, int act, int R) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  const int s = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int CV = C / EPV, RPP = 256 / CV, cpg = C / 32;
  const int col = tid % CV, rl = tid / CV;
  if (rl >= RPP) return;
  const int64_t base = gnb_base(g, s);
  float av[EPV], bv[EPV], k1[EPV], k2[EPV], k3[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) {
    const int c = col * EPV + e, gi = c / cpg;
    av[e] = a[(int64_t)s * C + c];
    bv[e] = b[(int64_t)s * C + c];
    const float mu = mr[((int64_t)s * 32 + gi) * 2], rs = mr[((int64_t)s * 32 + gi) * 2 + 1];
    const float m1 = m12[((int64_t)s * 32 + gi) * 2], m2 = m12[((int64_t)s * 32 + gi) * 2 + 1];
    const float gc = (film ? 1.f + film[(int64_t)s * film_ld + c] : 1.f) * gamma[c];
    k1[e] = rs * gc;
    k2[e] = -rs * rs * m2;
    k3[e] = rs * (mu * rs * m2 - m1);
  }
  const int j0 = chunk * R, j1 = min(j0 + R, g.Tn);
  auto body = [&](const u32x4& vx, const u32x4& vd, int64_t row) {
    float fx[EPV], fd[EPV], out[EPV];
    Elt<T>::unpack(vx, fx);
    Elt<T>::unpack(vd, fd);
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      float dv = fd[e];
      if (act) {
        const float v = fx[e] * av[e] + bv[e];
        const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-v));      // v_rcp_f32 (1 ulp), like silu_f of the forward: the IEEE division is ~10 more VALU instructions per element
        dv *= sg * (1.f + v * (1.f - sg));
      }
      out[e] = k1[e] * dv + k2[e] * fx[e] + k3[e];
    }
    *(u32x4*)(dx + (row * lddx + col * EPV) * ES) = Elt<T>::pack(out);
  };
  int j = j0 + rl;
  for (; j + RPP < j1; j += 2 * RPP) {
    const int64_t r0 = base + (int64_t)j * g.tstride, r1 = base + (int64_t)(j + RPP) * g.tstride;
    const u32x4 x0 = *(const u32x4*)(x + (r0 * ldx + col * EPV) * ES);
    const u32x4 d0 = *(const u32x4*)(dy + (r0 * lddy + col * EPV) * ES);
    const u32x4 x1 = *(const u32x4*)(x + (r1 * ldx + col * EPV) * ES);
    const u32x4 d1 = *(const u32x4*)(dy + (r1 * lddy + col * EPV) * ES);
    body(x0, d0, r0);
    body(x1, d1, r1);
  }
  if (j < j1) {
    const int64_t r0 = base + (int64_t)j * g.tstride;
    body(*(const u32x4*)(x + (r0 * ldx + col * EPV) * ES), *(const u32x4*)(dy + (r0 * lddy + col * EPV) * ES), r0);
  }
}

// ============================================================================= elementwise
template <typename T>
__global__ __launch_bounds__(256) void silu_kernel(const char* __restrict__ x, const char* __restrict__ dy, char* __restrict__ out, int64_t nvec) {
  constexpr int EPV = Elt<T>::EPV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float f[EPV], d[EPV];
    Elt<T>::unpack(((const u32x4*)x)[i], f);
    if (dy) Elt<T>::unpack(((const u32x4*)dy)[i], d);
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-f[e]));
      f[e] = dy ? d[e] * sg * (1.f + f[e] * (1.f - sg)) : f[e] * sg;
    }
    ((u32x4*)out)[i] = Elt<T>::pack(f);
  }
}

// inverted dropout: out = x * mask * scale (mask bytes 0/1, drawn by the caller); the backward is the same kernel on dy
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const char* __restrict__ x, const uint8_t* __restrict__ mask, float scale,
                                                      char* __restrict__ out, int64_t nvec) {
  constexpr int EPV = Elt<T>::EPV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float f[EPV];
    Elt<T>::unpack(((const u32x4*)x)[i], f);
#pragma unroll
    for (int e = 0; e < EPV; ++e) f[e] = mask[i * EPV + e] ? f[e] * scale : 0.f;
    ((u32x4*)out)[i] = Elt<T>::pack(f);
  }
}

// d loss / d out for loss = sum_n w[n] * mean_n((target - out)^2):  g = 2 (out - target) * w[n] / per
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ out, const float* __restrict__ target,
                                                       const float* __restrict__ w, float* __restrict__ g, int64_t per, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    g[i] = 2.f * (out[i] - target[i]) * w[i / per] / (float)per;
}

// AdamW step (torch.optim.AdamW semantics: decoupled weight decay) + optional EMA update (nn.py:128-138)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    float* __restrict__ ema, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
                                                    float bc1, float bc2, float ema_rate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float w = p[i] * (1.f - lr * wd);
    const float gi = g[i];
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    w -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    p[i] = w;
    if (ema) ema[i] = ema[i] * ema_rate + w * (1.f - ema_rate);
  }
}

// ============================================================================= C-ABI
static inline int ew_grid_b(int64_t total) { return (int)min((int64_t)4096, (total + 255) / 256); }

// dW (fp32 [Cout][ntaps*Cin], caller zeroes it) += dY^T * gather(X); db (nullable, fp32 [Cout], zeroed) += colsum(dY).
extern "C" int mmd_conv_wgrad(int dtype, const void* dY, int64_t lddy, const void* X, int64_t ldx, float* dW, float* db, int M, int Cout,
                              int Cin, int ntaps, const int* taps, int D0, int D1, int D2, int torch_layout, void* stream) {
  const int epv = dtype == MMD_BF16 ? 8 : 4;
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "conv_wgrad: bad dtype");
  MMD_REQUIRE(dY && X && dW && taps && M > 0 && ntaps >= 1 && ntaps <= 27, "conv_wgrad: bad argument");
  MMD_REQUIRE(Cin % epv == 0 && Cout % epv == 0 && lddy % epv == 0 && ldx % epv == 0, "conv_wgrad: channel counts / strides must be 16-byte multiples");
  MMD_REQUIRE(((uintptr_t)dY | (uintptr_t)X) % 16 == 0, "conv_wgrad: unaligned pointer");
  WgradParams p;
  p.dY = (const char*)dY; p.lddy = lddy; p.X = (const char*)X; p.ldx = ldx; p.dW = dW; p.db = nullptr; p.torch_layout = torch_layout;
  p.M = M; p.Cout = Cout; p.Cin = Cin; p.ntaps = ntaps; p.D0 = D0; p.D1 = D1; p.D2 = D2;
  for (int i = 0; i < ntaps * 3; ++i) p.taps[i] = taps[i];
  hipStream_t st = (hipStream_t)stream;
  static const bool use128 = getenv("MMD_WGRAD_TILE64") == nullptr;      // A/B switch for tools/wgrad_bench.py
  if (use128 && dtype == MMD_BF16 && Cout >= 64 && Cin >= 64) {          // transposed-staging 128x128 kernel
    const int tiles = cdiv(Cout, 128) * cdiv(Cin, 128) * ntaps;
    // Row splits per launch.  Every split adds Cout * Cin * ntaps atomics on the same addresses; the blocks of a split share its rows of dY / X.
    // Round 6 (XCD-aware block order in wgrad_tr_bf16_kernel: a split count that is a multiple of 8 gives every XCD whole splits, the blocks of
    // a split back to back on ONE L2; other counts run in plain grid order): the candidates are what the block targets 256 ... 2048 give, and
    // the choice is a three-term cost fitted to a sweep of the training shapes (profiles/r06_wgrad_xcd_order.txt: it picks the measured best
    // or within 3 % of it on all 13): rounds of blocks on the 64 (per XCD) / 512 (chip) block slots x (rows per split + 300) + 40 x splits.
    // MMD_WGRAD_BLOCKS: one fixed target (the sweep).
    auto derive = [&](int target, int& sp, int& rps, int& xo) {
      sp = max(1, min(cdiv(M, 256), target / max(tiles, 1)));
      rps = cdiv(cdiv(M, sp), 64) * 64;
      sp = cdiv(M, rps);
      xo = 0;
      for (int s8 = sp / 8 * 8; s8 >= 8; s8 -= 8) {
        const int r = cdiv(cdiv(M, s8), 64) * 64;
        if (cdiv(M, r) % 8 == 0) { rps = r; sp = cdiv(M, r); xo = 1; break; }
      }
    };
    int splits = 1, xo = 0;
    p.rows_per_split = M;
    {
      static const int env_target = getenv("MMD_WGRAD_BLOCKS") ? atoi(getenv("MMD_WGRAD_BLOCKS")) : 0;
      static const int targets[] = {256, 384, 512, 768, 1024, 1536, 2048};
      int64_t best = -1;
      for (int t : targets) {
        int sp, rps, x;
        derive(env_target > 0 ? env_target : t, sp, rps, x);
        const int64_t rounds = x ? cdiv((int64_t)tiles * (sp / 8), 64) : cdiv((int64_t)tiles * sp, 512);
        const int64_t cost = rounds * (rps + 300) + 40 * (int64_t)sp;
        if (best < 0 || cost < best) { best = cost; splits = sp; p.rows_per_split = rps; xo = x; }
      }
    }
    p.xcd_order = xo;
    // DMA-staged kernel (round 3; MMD_WGRAD_TR=0: the transposed-staging kernel): 32-bit byte offsets, float-reciprocal row positions
    static const bool use_tr = [] { const char* e = getenv("MMD_WGRAD_TR"); return !(e && e[0] == '0'); }();
    // measured (tools/wgrad_bench.py, batch 8): 3x3 ds1 128->128 436 -> 372 us, ds2 256->256 405 -> 251, ds4 384->384 240 -> 180, ds8 130 -> 110;
    // the 1x1 / k=3 convs lose what the separate colsum launch costs (1x1 ds1: 82 -> 127 us), so they stay on the older kernel
    // round 6: the bias gradient rides in the kernel (no colsum launch), so every tap count uses it; MMD_WGRAD_TR=9: the 9-tap convs only
    static const int tr_min_taps = [] { const char* e = getenv("MMD_WGRAD_TR"); return e && e[0] == '9' ? 9 : 1; }();
    if (use_tr && ntaps >= tr_min_taps && M < (1 << 24) && (int64_t)M * lddy * 2 < 0x7fffffffLL && (int64_t)M * ldx * 2 < 0x7fffffffLL) {
      const size_t lds = 2 * 4 * 64 * 128;
      static bool attr_done[MMD_MAX_DEVICES] = {};
      bool& attr_set = attr_done[mmd_device_slot()];
      if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_tr_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "conv_wgrad: set LDS attr: %s", hipGetErrorString(e));
        attr_set = true;
      }
      p.db = db;                                                 // column sums ride in the (tap 0, ci tile 0) blocks: no colsum launch
      p.splits = splits;
      hipLaunchKernelGGL(wgrad_tr_bf16_kernel, dim3(p.xcd_order ? 8 * tiles * cdiv(splits, 8) : tiles * splits), dim3(256), lds, st, p);
      return mmd_check_launch("conv_wgrad");
    } else {
      p.db = db;                                                 // column sums ride in the (tap 0, ci tile 0) blocks: no colsum launch
      hipLaunchKernelGGL(wgrad128_bf16_kernel, dim3(cdiv(Cout, 128), cdiv(Cin, 128) * ntaps, splits), dim3(256), 0, st, p);
      return mmd_check_launch("conv_wgrad");
    }
  } else {
    const int tiles = cdiv(Cout, 64) * cdiv(Cin, 64) * ntaps;
    int splits = max(1, min(cdiv(M, 256), 2048 / max(tiles, 1)));
    p.rows_per_split = cdiv(cdiv(M, splits), 64) * 64;
    splits = cdiv(M, p.rows_per_split);
    dim3 grid(cdiv(Cout, 64), cdiv(Cin, 64) * ntaps, splits);
    if (dtype == MMD_BF16) hipLaunchKernelGGL(wgrad_kernel<__bf16>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(wgrad_kernel<float>, grid, dim3(256), 0, st, p);
  }
  int rc = mmd_check_launch("conv_wgrad");
  if (rc || !db) return rc;
  const int rpb = M >= 65536 ? 256 : 128;
  const int es = dtype == MMD_BF16 ? 2 : 4;
  for (int c0 = 0; c0 < Cout; c0 += 256 * epv) {       // a block covers <= 256 16-byte column vectors: wide fp32 outputs (qkv 1536) go in slabs
    const int cw = Cout - c0 < 256 * epv ? Cout - c0 : 256 * epv;
    const char* src = (const char*)dY + (int64_t)c0 * es;
    if (dtype == MMD_BF16) hipLaunchKernelGGL(colsum_kernel<__bf16>, dim3(cdiv(M, rpb)), dim3(256), 0, st, src, lddy, M, cw, db + c0, rpb);
    else hipLaunchKernelGGL(colsum_kernel<float>, dim3(cdiv(M, rpb)), dim3(256), 0, st, src, lddy, M, cw, db + c0, rpb);
  }
  return mmd_check_launch("colsum");
}

// out[s, c] += sum over the Tn rows of slice s of dY[row, c] (S contiguous slices): the gradient of a per-sample row bias - the
// non-FiLM ResBlock's h + emb_out (unet:473-477).  `out` [S, ldo] fp32 is ACCUMULATED (caller zeroes).
extern "C" int mmd_colsum_slices(int dtype, const void* dY, int64_t lddy, int S, int64_t Tn, int C, float* out, int64_t ldo, void* stream) {
  const int epv = dtype == MMD_BF16 ? 8 : 4, es = dtype == MMD_BF16 ? 2 : 4;
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "colsum_slices: bad dtype");
  MMD_REQUIRE(dY && out && S > 0 && Tn > 0 && C > 0 && C % epv == 0 && lddy % epv == 0 && ((uintptr_t)dY) % 16 == 0, "colsum_slices: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int rpb = Tn >= 65536 ? 256 : 128;
  for (int s = 0; s < S; ++s)
    for (int c0 = 0; c0 < C; c0 += 256 * epv) {
      const int cw = C - c0 < 256 * epv ? C - c0 : 256 * epv;
      const char* src = (const char*)dY + ((int64_t)s * Tn * lddy + c0) * es;
      if (dtype == MMD_BF16) hipLaunchKernelGGL(colsum_kernel<__bf16>, dim3(cdiv(Tn, rpb)), dim3(256), 0, st, src, lddy, (int)Tn, cw, out + (int64_t)s * ldo + c0, rpb);
      else hipLaunchKernelGGL(colsum_kernel<float>, dim3(cdiv(Tn, rpb)), dim3(256), 0, st, src, lddy, (int)Tn, cw, out + (int64_t)s * ldo + c0, rpb);
    }
  return mmd_check_launch("colsum_slices");
}

__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.f;
}

// GroupNorm(+FiLM)(+SiLU) backward.  a, b [S,C] and mr [S,32,2] (mean, rstd) come from the forward mmd_gn_stats.
// dgamma / dbeta fp32 [C] are ACCUMULATED (caller zeroes); dfilm (nullable) [S, >= 2C] receives (dscale | dshift);
// workspace: (S*C*2 + S*64) floats, zeroed by this call.
static int gn_bwd_impl(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx, int64_t rows, int C,
                       int S, int Tn, int inner, int64_t outer_stride, int64_t inner_stride, int64_t tstride, const float* a,
                       const float* b, const float* mr, const float* gamma, const float* beta, const float* film, int64_t film_ld,
                       int act, float* dgamma, float* dbeta, float* dfilm, int64_t dfilm_ld, float* workspace, void* stream, bool ws0) {
  const int epv = dtype == MMD_BF16 ? 8 : 4;
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "gn_bwd: bad dtype");
  MMD_REQUIRE(x && dy && dx && a && b && mr && gamma && beta && dgamma && dbeta && workspace, "gn_bwd: null pointer");
  MMD_REQUIRE(C % 32 == 0 && C % epv == 0 && C / epv <= 256 && C <= 1024, "gn_bwd: unsupported channel count %d", C);
  GnBwdGeom g{S, Tn, inner, outer_stride, inner_stride, tstride};
  hipStream_t st = (hipStream_t)stream;
  float* PQ = workspace;
  float* m12 = workspace + (int64_t)S * C * 2;
  // zeroed by a kernel, not hipMemsetAsync: memset nodes of a captured graph were observed to lose their ordering against the
  // neighbouring kernel nodes on replay (train_graph.py), a fill kernel is an ordinary node of the chain
  if (!ws0) {
    hipLaunchKernelGGL(zero_f32_kernel, dim3(ew_grid_b((int64_t)S * C * 2)), dim3(256), 0, st, PQ, (int64_t)S * C * 2);
    if (int zrc = mmd_check_launch("gn_bwd_zero")) return zrc;
  }
  const int rpp = max(1, 256 / (C / epv));
  int R = 4 * rpp;
  while ((int64_t)S * cdiv(Tn, R) > 1280 && R < 1024) R *= 2;
  dim3 grid(cdiv(Tn, R), S);
  if (dtype == MMD_BF16)
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<__bf16>, grid, dim3(256), 0, st, (const char*)x, ldx, (const char*)dy, lddy, C, g, a, b, mr, act, R, PQ);
  else
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, grid, dim3(256), 0, st, (const char*)x, ldx, (const char*)dy, lddy, C, g, a, b, mr, act, R, PQ);
  int rc = mmd_check_launch("gn_bwd_reduce");
  if (rc) return rc;
  hipLaunchKernelGGL(gn_bwd_params_kernel, dim3(4, S), dim3(256), 0, st, PQ, C, Tn, gamma, beta, film, film_ld, dgamma, dbeta, dfilm,
                     dfilm_ld, m12, ws0 ? 1 : 0);
  rc = mmd_check_launch("gn_bwd_params");
  if (rc) return rc;
  int R3 = 4 * rpp;
  while ((int64_t)S * cdiv(Tn, R3) > 4096 && R3 < 1024) R3 *= 2;
  dim3 grid3(cdiv(Tn, R3), S);
  (void)rows;
  if (dtype == MMD_BF16)
    hipLaunchKernelGGL(gn_bwd_apply_kernel<__bf16>, grid3, dim3(256), 0, st, (const char*)x, ldx, (const char*)dy, lddy, (char*)dx, lddx, C, g,
                       a, b, mr, (const float*)m12, gamma, film, film_ld, act, R3);
  else
    hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, grid3, dim3(256), 0, st, (const char*)x, ldx, (const char*)dy, lddy, (char*)dx, lddx, C, g,
                       a, b, mr, (const float*)m12, gamma, film, film_ld, act, R3);
  return mmd_check_launch("gn_bwd_apply");
}

extern "C" int mmd_gn_bwd(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx, int64_t rows, int C,
                          int S, int Tn, int inner, int64_t outer_stride, int64_t inner_stride, int64_t tstride, const float* a,
                          const float* b, const float* mr, const float* gamma, const float* beta, const float* film, int64_t film_ld,
                          int act, float* dgamma, float* dbeta, float* dfilm, int64_t dfilm_ld, float* workspace, void* stream) {
  return gn_bwd_impl(dtype, x, ldx, dy, lddy, dx, lddx, rows, C, S, Tn, inner, outer_stride, inner_stride, tstride, a, b, mr, gamma, beta, film,
                     film_ld, act, dgamma, dbeta, dfilm, dfilm_ld, workspace, stream, false);
}
// The same with a workspace the CALLER keeps: its first S * C * 2 floats are zero on entry and zero again on exit (the parameter stage
// clears what it has read), so a training step's 243 norms do not pay a fill launch each.  One workspace per stream.
extern "C" int mmd_gn_bwd_ws0(int dtype, const void* x, int64_t ldx, const void* dy, int64_t lddy, void* dx, int64_t lddx, int64_t rows, int C,
                              int S, int Tn, int inner, int64_t outer_stride, int64_t inner_stride, int64_t tstride, const float* a,
                              const float* b, const float* mr, const float* gamma, const float* beta, const float* film, int64_t film_ld,
                              int act, float* dgamma, float* dbeta, float* dfilm, int64_t dfilm_ld, float* workspace, void* stream) {
  return gn_bwd_impl(dtype, x, ldx, dy, lddy, dx, lddx, rows, C, S, Tn, inner, outer_stride, inner_stride, tstride, a, b, mr, gamma, beta, film,
                     film_ld, act, dgamma, dbeta, dfilm, dfilm_ld, workspace, stream, true);
}

// out = silu(x) (dy == NULL) or dy * silu'(x); contiguous buffers of n elements (n % (16/elsize) == 0)
extern "C" int mmd_silu(int dtype, const void* x, const void* dy, void* out, int64_t n, void* stream) {
  const int epv = dtype == MMD_BF16 ? 8 : 4;
  MMD_REQUIRE((dtype == MMD_BF16 || dtype == MMD_F32) && x && out && n > 0 && n % epv == 0, "silu: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MMD_BF16) hipLaunchKernelGGL(silu_kernel<__bf16>, dim3(ew_grid_b(n / epv)), dim3(256), 0, st, (const char*)x, (const char*)dy, (char*)out, n / epv);
  else hipLaunchKernelGGL(silu_kernel<float>, dim3(ew_grid_b(n / epv)), dim3(256), 0, st, (const char*)x, (const char*)dy, (char*)out, n / epv);
  return mmd_check_launch("silu");
}

extern "C" int mmd_dropout(int dtype, const void* x, const uint8_t* mask, float scale, void* out, int64_t n, void* stream) {
  const int epv = dtype == MMD_BF16 ? 8 : 4;
  MMD_REQUIRE((dtype == MMD_BF16 || dtype == MMD_F32) && x && mask && out && n > 0 && n % epv == 0, "dropout: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MMD_BF16) hipLaunchKernelGGL(dropout_kernel<__bf16>, dim3(ew_grid_b(n / epv)), dim3(256), 0, st, (const char*)x, mask, scale, (char*)out, n / epv);
  else hipLaunchKernelGGL(dropout_kernel<float>, dim3(ew_grid_b(n / epv)), dim3(256), 0, st, (const char*)x, mask, scale, (char*)out, n / epv);
  return mmd_check_launch("dropout");
}

extern "C" int mmd_mse_grad(const float* out, const float* target, const float* w, float* g, int N, int64_t per_sample, void* stream) {
  MMD_REQUIRE(out && target && w && g && N > 0 && per_sample > 0, "mse_grad: bad argument");
  const int64_t total = per_sample * N;
  hipLaunchKernelGGL(mse_grad_kernel, dim3(ew_grid_b(total)), dim3(256), 0, (hipStream_t)stream, out, target, w, g, per_sample, total);
  return mmd_check_launch("mse_grad");
}

extern "C" int mmd_adamw_step(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, float ema_rate, void* stream) {
  MMD_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw_step: bad argument");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid_b(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, n, lr, beta1, beta2, eps, weight_decay,
                     bc1, bc2, ema_rate);
  return mmd_check_launch("adamw_step");
}

// ----------------------------------------------------------------------------- weight packing for the training step
// One launch re-packs EVERY conv weight after an optimizer step: src fp32 [Cout][Cin][nt] (torch conv layout, a view of the
// flat parameter buffer) -> fwd [Cout][nt*Cin] (GEMM operand of the forward conv) and bwd [Cin][nt*Cout] (operand of the
// data-gradient conv) in the activation dtype.  The per-weight torch version was ~6 tiny kernels per conv per step.
struct PackDesc {
  const float* src; void* fwd; void* bwd;
  int Cout, Cin, nt;
  int block_start;                 // first block of this weight; a block covers a tile of 32 output x 16 input channels x all taps
};
// Round 6: tiles through LDS.  The first version walked 2048 consecutive source elements per block and stored each one twice with 2-byte
// stores Cin / Cout elements apart, behind a nine-deep dependent binary search of the descriptor table per block (2.07 ms per step for
// the 133 M parameters = 0.39 TB/s of traffic that should take 0.2 ms).  A tile of 32 co x (216 / nt) ci x nt taps is read as 32 contiguous
// runs, then written as runs of consecutive ci per (co, tap) into the forward operand and runs of 32 consecutive co per (ci, tap) into
// the data-gradient operand; no integer division in the loops; the descriptor of a block is found by ONE parallel pass over the table.
#define PK_CO 32
#define PK_ROW 216
#define PK_NT_MAX 27
__host__ __device__ __forceinline__ int pack_ci_tile(int nt) { const int c = PK_ROW / nt; return c >= 16 ? (c & ~15) : c; }
__device__ __forceinline__ PackDesc pack_find(const PackDesc* __restrict__ descs, int n, int* s_cnt) {
  if (threadIdx.x == 0) *s_cnt = 0;
  __syncthreads();
  int local = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) local += descs[i].block_start <= (int)blockIdx.x ? 1 : 0;   // block_start ascends
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(s_cnt, local);
  __syncthreads();
  return descs[*s_cnt - 1];
}
struct PackTile { int co0, ci0, nco, nci, nt, run; };
__device__ __forceinline__ PackTile pack_tile(const PackDesc& d) {
  const int ct = pack_ci_tile(d.nt);
  const int tiles_ci = (d.Cin + ct - 1) / ct;
  const int tb = (int)blockIdx.x - d.block_start;
  PackTile t;
  t.co0 = (tb / tiles_ci) * PK_CO;
  t.ci0 = (tb % tiles_ci) * ct;
  t.nco = min(PK_CO, d.Cout - t.co0);
  t.nci = min(ct, d.Cin - t.ci0);
  t.nt = d.nt;
  t.run = t.nci * d.nt;
  return t;
}
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const PackDesc* __restrict__ descs, int n) {
  __shared__ float sT[PK_CO][PK_ROW + 1];
  __shared__ int s_cnt;
  const PackDesc d = pack_find(descs, n, &s_cnt);
  const PackTile t = pack_tile(d);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c = wave; c < t.nco; c += 4) {
    const float* sp = d.src + ((int64_t)(t.co0 + c) * d.Cin + t.ci0) * t.nt;
    for (int r = lane; r < t.run; r += 64) sT[c][r] = sp[r];
  }
  __syncthreads();
  // forward operand [Cout][nt * Cin]: lanes along ci; a tile narrower than a wave (16 ci at nine taps) puts 64 / nci taps side by side
  const int tstep = t.nci >= 64 ? 1 : 64 / t.nci, lt = lane / t.nci, lci = lane - lt * t.nci, cstep = tstep > 1 ? t.nci : 64;
  if (lt < tstep)
    for (int c = wave; c < t.nco; c += 4)
      for (int tap = lt; tap < t.nt; tap += tstep) {
        const int64_t base = (int64_t)(t.co0 + c) * t.nt * d.Cin + (int64_t)tap * d.Cin + t.ci0;
        for (int ci = lci; ci < t.nci; ci += cstep) Elt<T>::st(d.fwd, base + ci, sT[c][ci * t.nt + tap]);
      }
  const int c = lane & 31, h = lane >> 5;                      // data-gradient operand [Cin][nt * Cout]: lanes along co, two taps per wave pass
  if (c < t.nco)
    for (int ci = wave; ci < t.nci; ci += 4)
      for (int tap = h; tap < t.nt; tap += 2)
        Elt<T>::st(d.bwd, (int64_t)(t.ci0 + ci) * t.nt * d.Cout + (int64_t)tap * d.Cout + t.co0 + c, sT[c][ci * t.nt + tap]);
}
// Reverse direction for the gradients: wgrad accumulates with COALESCED atomics in the packed layout [Cout][nt*Cin] (fp32, `fwd` of
// the record); once per step this adds every packed gradient into the parameter's .grad ([Cout][Cin][nt], `src` of the record,
// written here) and clears the packed buffer for the next step.  (Atomics straight into the torch layout are strided by nt
// floats: 18 64-byte segments per wave instruction instead of 2.)  Same tiles as the weight pack: packed runs of consecutive ci per
// (co, tap) in, .grad runs of nci x nt floats per co out.
__global__ __launch_bounds__(256) void unpack_grads_kernel(const PackDesc* __restrict__ descs, int n) {
  __shared__ float sT[PK_CO][PK_ROW + 1];
  __shared__ int s_cnt;
  const PackDesc d = pack_find(descs, n, &s_cnt);
  const PackTile t = pack_tile(d);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* grad = const_cast<float*>(d.src);
  float* packed = (float*)d.fwd;
  const int tstep = t.nci >= 64 ? 1 : 64 / t.nci, lt = lane / t.nci, lci = lane - lt * t.nci, cstep = tstep > 1 ? t.nci : 64;
  if (lt < tstep)
    for (int c = wave; c < t.nco; c += 4)
      for (int tap = lt; tap < t.nt; tap += tstep) {
        float* pp = packed + (int64_t)(t.co0 + c) * t.nt * d.Cin + (int64_t)tap * d.Cin + t.ci0;
        for (int ci = lci; ci < t.nci; ci += cstep) {
          sT[c][ci * t.nt + tap] = pp[ci];
          pp[ci] = 0.f;
        }
      }
  __syncthreads();
  for (int c = wave; c < t.nco; c += 4) {
    float* gp = grad + ((int64_t)(t.co0 + c) * d.Cin + t.ci0) * t.nt;
    for (int r = lane; r < t.run; r += 64) gp[r] += sT[c][r];
  }
}

extern "C" int mmd_pack_blocks(int Cout, int Cin, int nt) {
  if (Cout <= 0 || Cin <= 0 || nt <= 0 || nt > PK_NT_MAX) return -1;
  const int ct = pack_ci_tile(nt);
  return ((Cout + PK_CO - 1) / PK_CO) * ((Cin + ct - 1) / ct);
}

extern "C" int mmd_unpack_conv_grads(const void* descs_dev, int n, int total_blocks, void* stream) {
  MMD_REQUIRE(descs_dev && n > 0 && total_blocks > 0, "unpack_conv_grads: bad argument");      // (descriptors: at most 27 taps, mmd_pack_blocks blocks each)
  hipLaunchKernelGGL(unpack_grads_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)descs_dev, n);
  return mmd_check_launch("unpack_conv_grads");
}

extern "C" int mmd_pack_conv_weights(int dtype, const void* descs_dev, int n, int total_blocks, void* stream) {
  MMD_REQUIRE((dtype == MMD_BF16 || dtype == MMD_F32) && descs_dev && n > 0 && total_blocks > 0, "pack_conv_weights: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MMD_BF16) hipLaunchKernelGGL(pack_weights_kernel<__bf16>, dim3(total_blocks), dim3(256), 0, st, (const PackDesc*)descs_dev, n);
  else hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(total_blocks), dim3(256), 0, st, (const PackDesc*)descs_dev, n);
  return mmd_check_launch("pack_conv_weights");
}
