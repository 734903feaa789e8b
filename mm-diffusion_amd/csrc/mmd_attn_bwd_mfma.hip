// bf16 MFMA attention backward (training step) for contiguous-row attention: spatial / audio self-attention and the
// random-shift windowed cross-modal attention (temporal attention, whose rows are strided, uses attn_small_bwd).
//
// Same register choreography as the forward kernel (mmd_attn.hip): every lane owns ONE query (dQ kernel) or ONE key
// (dK/dV kernel), so the softmax statistics, the dS = P (dP - D) product and the window mask are lane-local, and the
// 32x32 C layout of the score tile is exactly the k-slot order of the following MFMA's B operand.
//   dQ kernel : S^T = K Q^T, dP^T = V dO^T (A = row-major K / V tiles), dQ^T += K^T dS^T (A = K^T tile, key-permuted)
//   dKV kernel: S = Q K^T, dP = dO V^T (A = row-major Q / dO tiles), dV^T += dO^T P, dK^T += Q^T dS (A = transposed tiles)
// P is recomputed from the forward's log2-domain log-sum-exp; D_i = dO_i . O_i is produced by the dQ kernel.
#include "mmd_common.h"

struct AttnBwdMParams {
  const char* Q; int64_t ldq; int q_off;
  const char* KV; int64_t ldkv; int k_off, v_off;
  const char* O; int64_t ldo;
  const char* dO; int64_t lddo;
  char* dQ; int64_t lddq; int dq_off;
  char* dKV; int64_t lddkv; int dk_off, dv_off;
  const float* lse2; float* dsum;
  int heads, nb, G;
  int64_t q_rows_per_batch; int q_per_group;
  int64_t k_rows_per_batch; int k_per_group, win;
  const int* shift_ptr;
  float scale;
};

#define TSTRIDE 136     // bytes per row of a transposed [D][64] tile ((stride/8) odd: conflict-free ds_read_b64)

__device__ __forceinline__ int bm_qcount(const AttnBwdMParams& p, int g) {
  return g == p.G - 1 ? (int)(p.q_rows_per_batch - (int64_t)g * p.q_per_group) : p.q_per_group;
}
__device__ __forceinline__ int bm_kstart(const AttnBwdMParams& p, int g) {
  const int shift = p.shift_ptr ? *p.shift_ptr : 0;
  return (int)(((int64_t)(g + shift) * p.k_per_group) % p.k_rows_per_batch);
}

// Stage a [64 rows][D] bf16 tile: row-major into `rm` (row stride SK) and/or transposed into `tr` ([D][64], row-permuted
// exactly like the forward's V^T so the MFMA k-slots line up with the score-tile registers).  row_ptr(j) gives the
// global address of element 0 of row j or nullptr (zero fill).
template <int D, bool RM, bool TR, typename F>
__device__ __forceinline__ void stage_tile(char* rm, char* tr, int tid, F row_ptr) {
  constexpr int DV = D / 8, SK = D * 2 + 16;
  constexpr int NK = (64 * DV + 255) / 256;
#pragma unroll
  for (int i = 0; i < NK; ++i) {
    const int id = tid + 256 * i;
    if (id < 64 * DV) {
      if (RM) {
        const int j = id / DV, v = id % DV;
        const char* src = row_ptr(j);
        u32x4 x = {0u, 0u, 0u, 0u};
        if (src) x = *(const u32x4*)(src + v * 16);
        *(u32x4*)(rm + j * SK + v * 16) = x;
      }
      if (TR) {
        const int j = (id & 31) + 32 * ((id >> 6) & 1);
        const int v = 2 * (id >> 7) + ((id >> 5) & 1);
        const char* src = row_ptr(j);
        u32x4 x = {0u, 0u, 0u, 0u};
        if (src) x = *(const u32x4*)(src + v * 16);
        uint16_t* dst = (uint16_t*)(tr + (8 * v) * TSTRIDE + 2 * j);
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e * (TSTRIDE / 2)] = (uint16_t)((x[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
      }
    }
  }
}

// A fragment (8 k-slots) of a transposed tile for MFMA step (kt, st): slots j<4 -> cols 32kt+16st+4half+j, j>=4 -> +8
__device__ __forceinline__ u32x4 tr_frag(const char* tr, int row, int kt, int st, int half) {
  const char* vb = tr + row * TSTRIDE + (32 * kt + 16 * st + 4 * half) * 2;
  const u32x2 v0 = *(const u32x2*)(vb);
  const u32x2 v1 = *(const u32x2*)(vb + 16);
  return u32x4{v0[0], v0[1], v1[0], v1[1]};
}

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

// XCD-aware block order (round 6; the forward kernels' attn_block_coords, mmd_attn.hip): workgroups are dealt round-robin to the eight XCDs
// by flat id and blockIdx.x (the query / key tile) is the fastest index, so the tiles of one (head, batch-group) land on eight different L2s
// and each fetches the same K / V (dQ) or Q / dO (dK, dV) rows again.  Remapped so that all tiles of a (head, batch-group) share flat id % 8
// and are adjacent in dispatch order.  Returns (tile, head, blockIdx.z-equivalent).
__device__ __forceinline__ void bm_block_coords(int& tile, int& h, int& z) {
  const int nx = gridDim.x, ny = gridDim.y, nz = gridDim.z;
  int hz;
  if (((ny * nz) & 7) == 0 && nx > 1) {
    const int f = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int k = f >> 3, r = f & 7;
    tile = k % nx;
    hz = r + 8 * (k / nx);
  } else {
    tile = blockIdx.x;
    hz = blockIdx.y + ny * blockIdx.z;
  }
  h = hz % ny;
  z = hz / ny;
}

// ============================================================================= dQ
template <int D>
__global__ __launch_bounds__(256, (D <= 64 ? 2 : 1)) void attn_bwd_dq_mfma_kernel(const AttnBwdMParams p) {
  constexpr int SK = D * 2 + 16, KST = D / 16, DT = (D + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                          // [64][SK]
  char* sV = smem + 64 * SK;                // [64][SK]
  char* sKt = smem + 128 * SK;              // [DT*32][TSTRIDE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  int qtile, h, bz;
  bm_block_coords(qtile, h, bz);
  const int n = bz / p.G, g = bz % p.G;
  const int qcount = bm_qcount(p, g);
  const int q0 = qtile * 128;
  if (q0 >= qcount) return;
  const int64_t q_row0 = (int64_t)n * p.q_rows_per_batch + (int64_t)g * p.q_per_group;
  const int64_t k_row0 = (int64_t)n * p.k_rows_per_batch;
  const int k_mod = (int)p.k_rows_per_batch, kcount = p.win * p.k_per_group, kstart = bm_kstart(p, g);
  if (D % 32 != 0)
    for (int i = tid; i < (DT * 32 - D) * TSTRIDE / 4; i += 256) ((uint32_t*)(sKt + D * TSTRIDE))[i] = 0u;

  const int qi = q0 + wave * 32 + l31;
  const bool qok = qi < qcount;
  const int64_t qrow = q_row0 + (qok ? qi : 0);
  u32x4 qf[KST], dof[KST];
  float Dq = 0.f;
  {
    const char* qp = p.Q + (qrow * p.ldq + p.q_off + h * D) * 2;
    const char* gp = p.dO + (qrow * p.lddo + h * D) * 2;
    const char* op = p.O + (qrow * p.ldo + h * D) * 2;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u}, c = {0u, 0u, 0u, 0u};
      if (qok) {
        a = *(const u32x4*)(qp + (s * 16 + half * 8) * 2);
        b = *(const u32x4*)(gp + (s * 16 + half * 8) * 2);
        c = *(const u32x4*)(op + (s * 16 + half * 8) * 2);
      }
      qf[s] = a;
      dof[s] = b;
      float fb[8], fc[8];
      Elt<__bf16>::unpack(b, fb);
      Elt<__bf16>::unpack(c, fc);
#pragma unroll
      for (int e = 0; e < 8; ++e) Dq += fb[e] * fc[e];
    }
  }
  Dq += __shfl_xor(Dq, 32, 64);
  const float lse = qok ? p.lse2[qrow * p.heads + h] : 0.f;
  if (qok && half == 0) p.dsum[qrow * p.heads + h] = Dq;

  f32x16 dq[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[t][r] = 0.f;
  const float sc = p.scale * 1.4426950408889634f;
  const int ntiles = (kcount + 63) >> 6;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    auto kptr = [&](int j) -> const char* {
      const int kk = t * 64 + j;
      if (kk >= kcount) return nullptr;
      int r = kstart + kk;
      if (r >= k_mod) r -= k_mod;
      return p.KV + ((k_row0 + r) * p.ldkv + p.k_off + h * D) * 2;
    };
    auto vptr = [&](int j) -> const char* {
      const int kk = t * 64 + j;
      if (kk >= kcount) return nullptr;
      int r = kstart + kk;
      if (r >= k_mod) r -= k_mod;
      return p.KV + ((k_row0 + r) * p.ldkv + p.v_off + h * D) * 2;
    };
    stage_tile<D, true, true>(sK, sKt, tid, kptr);
    stage_tile<D, true, false>(sV, nullptr, tid, vptr);
    __syncthreads();
    f32x16 s[2], dp[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kt][r] = 0.f; dp[kt][r] = 0.f; }
#pragma unroll
    for (int st = 0; st < KST; ++st)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        const u32x4 kf = *(const u32x4*)(sK + (kt * 32 + l31) * SK + half * 16 + st * 32);
        const u32x4 vf = *(const u32x4*)(sV + (kt * 32 + l31) * SK + half * 16 + st * 32);
        s[kt] = MFMA_BF16(kf, qf[st], s[kt]);
        dp[kt] = MFMA_BF16(vf, dof[st], dp[kt]);
      }
    const int kbase = t * 64 + 4 * half;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = kbase + 32 * kt + (r & 3) + 8 * (r >> 2);
        const float pr = kk < kcount ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], sc, -lse)) : 0.f;
        s[kt][r] = pr * (dp[kt][r] - Dq);                 // dS^T for (key, this lane's query)
      }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        bf16x8 df;
#pragma unroll
        for (int j = 0; j < 8; ++j) df[j] = (__bf16)s[kt][8 * st + j];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const u32x4 kf = tr_frag(sKt, dt * 32 + l31, kt, st, half);
          dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), df, dq[dt], 0, 0, 0);
        }
      }
  }
  if (qok) {
    char* op = p.dQ + (qrow * p.lddq + p.dq_off + h * D) * 2;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = dt * 32 + 8 * q4 + 4 * half;
        if (d < D) {
          bf16x4 w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = (__bf16)(dq[dt][4 * q4 + j] * p.scale);
          *(bf16x4*)(op + d * 2) = w;
        }
      }
  }
}

// ============================================================================= dK / dV
template <int D>
__global__ __launch_bounds__(256, (D <= 64 ? 2 : 1)) void attn_bwd_dkv_mfma_kernel(const AttnBwdMParams p) {
  constexpr int SK = D * 2 + 16, KST = D / 16, DT = (D + 31) / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQ = smem;                                   // [64][SK]
  char* sdO = smem + 64 * SK;                        // [64][SK]
  char* sQt = smem + 128 * SK;                       // [DT*32][TSTRIDE]
  char* sdOt = sQt + DT * 32 * TSTRIDE;
  float* sLse = (float*)(sdOt + DT * 32 * TSTRIDE);  // [64]
  float* sD = sLse + 64;                             // [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  int ktile, h, n;
  bm_block_coords(ktile, h, n);
  const int k_mod = (int)p.k_rows_per_batch, kcount = p.win * p.k_per_group;
  const int k0 = ktile * 128;
  if (k0 >= k_mod) return;
  const int kidx = k0 + wave * 32 + l31;
  const bool kok = kidx < k_mod;
  const int64_t krow = (int64_t)n * p.k_rows_per_batch + (kok ? kidx : 0);
  if (D % 32 != 0)
    for (int i = tid; i < (DT * 32 - D) * TSTRIDE / 4; i += 256) {
      ((uint32_t*)(sQt + D * TSTRIDE))[i] = 0u;
      ((uint32_t*)(sdOt + D * TSTRIDE))[i] = 0u;
    }
  u32x4 kfr[KST], vfr[KST];
  {
    const char* kp = p.KV + (krow * p.ldkv + p.k_off + h * D) * 2;
    const char* vp = p.KV + (krow * p.ldkv + p.v_off + h * D) * 2;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
      if (kok) { a = *(const u32x4*)(kp + (s * 16 + half * 8) * 2); b = *(const u32x4*)(vp + (s * 16 + half * 8) * 2); }
      kfr[s] = a;
      vfr[s] = b;
    }
  }
  f32x16 dk[DT], dv[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[t][r] = 0.f; dv[t][r] = 0.f; }
  const float sc = p.scale * 1.4426950408889634f;

  for (int g = 0; g < p.G; ++g) {
    const int kstart = bm_kstart(p, g);
    int rel = kidx - kstart;
    if (rel < 0) rel += k_mod;
    const bool inwin = kok && rel < kcount;
    if (!__syncthreads_or(inwin ? 1 : 0)) continue;          // no key of this 128-key tile is in group g's window
    const int qcount = bm_qcount(p, g);
    const int64_t q_row0 = (int64_t)n * p.q_rows_per_batch + (int64_t)g * p.q_per_group;
    for (int q0 = 0; q0 < qcount; q0 += 64) {
      __syncthreads();
      auto qptr = [&](int j) -> const char* {
        return q0 + j < qcount ? p.Q + ((q_row0 + q0 + j) * p.ldq + p.q_off + h * D) * 2 : nullptr;
      };
      auto gptr = [&](int j) -> const char* {
        return q0 + j < qcount ? p.dO + ((q_row0 + q0 + j) * p.lddo + h * D) * 2 : nullptr;
      };
      stage_tile<D, true, true>(sQ, sQt, tid, qptr);
      stage_tile<D, true, true>(sdO, sdOt, tid, gptr);
      if (tid < 64) {
        const bool ok = q0 + tid < qcount;
        const int64_t r = q_row0 + q0 + (ok ? tid : 0);
        sLse[tid] = ok ? p.lse2[r * p.heads + h] : 0.f;
        sD[tid] = ok ? p.dsum[r * p.heads + h] : 0.f;
      }
      __syncthreads();
      f32x16 s[2], dp[2];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[qt][r] = 0.f; dp[qt][r] = 0.f; }
#pragma unroll
      for (int st = 0; st < KST; ++st)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          const u32x4 qfr = *(const u32x4*)(sQ + (qt * 32 + l31) * SK + half * 16 + st * 32);
          const u32x4 gfr = *(const u32x4*)(sdO + (qt * 32 + l31) * SK + half * 16 + st * 32);
          s[qt] = MFMA_BF16(qfr, kfr[st], s[qt]);             // rows = queries, cols = keys (this lane's key)
          dp[qt] = MFMA_BF16(gfr, vfr[st], dp[qt]);
        }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ql = 32 * qt + (r & 3) + 8 * (r >> 2) + 4 * half;
          const bool ok = inwin && (q0 + ql < qcount);
          const float pr = ok ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][r], sc, -sLse[ql])) : 0.f;
          s[qt][r] = pr;                                      // P
          dp[qt][r] = pr * (dp[qt][r] - sD[ql]);              // dS
        }
#pragma unroll
      for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          bf16x8 pf, df;
#pragma unroll
          for (int j = 0; j < 8; ++j) { pf[j] = (__bf16)s[qt][8 * st + j]; df[j] = (__bf16)dp[qt][8 * st + j]; }
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const u32x4 gt = tr_frag(sdOt, dt * 32 + l31, qt, st, half);
            const u32x4 qtf = tr_frag(sQt, dt * 32 + l31, qt, st, half);
            dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, gt), pf, dv[dt], 0, 0, 0);
            dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, qtf), df, dk[dt], 0, 0, 0);
          }
        }
    }
  }
  if (kok) {
    char* kp = p.dKV + (krow * p.lddkv + p.dk_off + h * D) * 2;
    char* vp = p.dKV + (krow * p.lddkv + p.dv_off + h * D) * 2;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = dt * 32 + 8 * q4 + 4 * half;
        if (d < D) {
          bf16x4 wk, wv;
#pragma unroll
          for (int j = 0; j < 4; ++j) { wk[j] = (__bf16)(dk[dt][4 * q4 + j] * p.scale); wv[j] = (__bf16)dv[dt][4 * q4 + j]; }
          *(bf16x4*)(kp + d * 2) = wk;
          *(bf16x4*)(vp + d * 2) = wv;
        }
      }
  }
}

template <int D>
static int launch_bwd_mfma(const AttnBwdMParams& p, hipStream_t st) {
  constexpr int SK = D * 2 + 16, DT = (D + 31) / 32;
  const size_t lds_q = 128 * SK + DT * 32 * TSTRIDE;
  const size_t lds_kv = 128 * SK + 2 * DT * 32 * TSTRIDE + 128 * sizeof(float);
  const int qmax = (int)(p.q_rows_per_batch - (int64_t)(p.G - 1) * p.q_per_group);
  if (lds_kv > 64 * 1024) {
    static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)attn_bwd_dkv_mfma_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv) != hipSuccess)
        return mmd_set_error(MMD_ERR_LAUNCH, "attn_bwd_mfma: set LDS attr failed");
      attr_set = true;
    }
  }
  hipLaunchKernelGGL(attn_bwd_dq_mfma_kernel<D>, dim3(cdiv(qmax, 128), p.heads, p.nb * p.G), dim3(256), lds_q, st, p);
  int rc = mmd_check_launch("attn_bwd_dq_mfma");
  if (rc) return rc;
  hipLaunchKernelGGL(attn_bwd_dkv_mfma_kernel<D>, dim3(cdiv(p.k_rows_per_batch, 128), p.heads, p.nb), dim3(256), lds_kv, st, p);
  return mmd_check_launch("attn_bwd_dkv_mfma");
}

// bf16 MFMA backward of mmd_attn_fwd_lse (same row/window arguments).  lse2 from the forward; dsum_ws fp32 [q rows, heads].
// dQ is written for every query row, dK/dV for every key row of every batch (column ranges dq_off / dk_off / dv_off).
extern "C" int mmd_attn_bwd_mfma(const void* Q, int64_t ldq, int q_off, const void* KV, int64_t ldkv, int k_off, int v_off,
                                 const void* O, int64_t ldo, const void* dO, int64_t lddo, void* dQ, int64_t lddq, int dq_off,
                                 void* dKV, int64_t lddkv, int dk_off, int dv_off, const float* lse2, float* dsum_ws, int heads,
                                 int ch, int nb, int G, int64_t q_rows_per_batch, int q_per_group, int64_t k_rows_per_batch,
                                 int k_per_group, int win, const int* shift_dev, void* stream) {
  MMD_REQUIRE(Q && KV && O && dO && dQ && dKV && lse2 && dsum_ws, "attn_bwd_mfma: null pointer");
  MMD_REQUIRE(heads > 0 && nb > 0 && G > 0 && (int64_t)win * k_per_group <= k_rows_per_batch, "attn_bwd_mfma: bad geometry");
  MMD_REQUIRE(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && lddq % 4 == 0 && lddkv % 4 == 0 && q_off % 8 == 0 &&
                  k_off % 8 == 0 && v_off % 8 == 0 && dq_off % 4 == 0 && dk_off % 4 == 0 && dv_off % 4 == 0,
              "attn_bwd_mfma: rows / column offsets must be 16-byte (inputs) and 8-byte (outputs) aligned");
  AttnBwdMParams p;
  p.Q = (const char*)Q; p.ldq = ldq; p.q_off = q_off; p.KV = (const char*)KV; p.ldkv = ldkv; p.k_off = k_off; p.v_off = v_off;
  p.O = (const char*)O; p.ldo = ldo; p.dO = (const char*)dO; p.lddo = lddo; p.dQ = (char*)dQ; p.lddq = lddq; p.dq_off = dq_off;
  p.dKV = (char*)dKV; p.lddkv = lddkv; p.dk_off = dk_off; p.dv_off = dv_off; p.lse2 = lse2; p.dsum = dsum_ws;
  p.heads = heads; p.nb = nb; p.G = G; p.q_rows_per_batch = q_rows_per_batch; p.q_per_group = q_per_group;
  p.k_rows_per_batch = k_rows_per_batch; p.k_per_group = k_per_group; p.win = win; p.shift_ptr = shift_dev;
  p.scale = 1.0f / sqrtf((float)ch);
  hipStream_t st = (hipStream_t)stream;
  switch (ch) {
    case 16: return launch_bwd_mfma<16>(p, st);
    case 32: return launch_bwd_mfma<32>(p, st);
    case 48: return launch_bwd_mfma<48>(p, st);
    case 64: return launch_bwd_mfma<64>(p, st);
    case 96: return launch_bwd_mfma<96>(p, st);
    case 128: return launch_bwd_mfma<128>(p, st);
    default: return mmd_set_error(MMD_ERR_UNSUPPORTED, "attn_bwd_mfma: head width %d not in {16,32,48,64,96,128}", ch);
  }
}
