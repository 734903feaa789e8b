// Attention backward (training step), one generic pair of kernels for every attention of the model:
//   spatial / audio / temporal self-attention (reference multimodal_unet.py:221-240) and the random-shift windowed
//   cross-modal attention in both directions (unet:507-564), through one strided + windowed row descriptor.
//
//   P = softmax(scale * Q K^T) (recomputed, never stored),  D_i = dO_i . O_i
//   dV_j = sum_i P_ij dO_i,   dS_ij = P_ij (dO_i . V_j - D_i),   dQ_i = scale sum_j dS_ij K_j,   dK_j = scale sum_i dS_ij Q_i
//
// attn_bwd_dq : parallel over query tiles; pass 1 recomputes the log-sum-exp of every query row (and D), pass 2
//               accumulates dQ; LSE and D go to a [rows, heads] workspace for the second kernel.
// attn_bwd_dkv: parallel over key tiles; walks the query groups whose window contains the tile (deterministic, no atomics).
// fp32 math on LDS tiles (correctness-first; the MFMA forward kernels are unaffected).
#include "mmd_common.h"

struct AttnBwdParams {
  const char* Q; int64_t ldq; int q_off;       // q at column q_off + h*ch
  const char* KV; int64_t ldkv; int k_off, v_off;
  const char* O; int64_t ldo;                  // forward output [q rows, heads*ch]
  const char* dO; int64_t lddo;
  char* dQ; int64_t lddq; int dq_off;          // gradient buffers (same row spaces as Q / KV)
  char* dKV; int64_t lddkv; int dk_off, dv_off;
  float* lse; float* dsum;                     // [q row space, heads]
  int heads, ch;
  int nb, G;
  // query rows of unit b: qbase(b) + idx * q_tstride, idx = g*q_per_group + i  (idx < q_count_total)
  int q_inner; int64_t q_outer, q_istride, q_tstride; int q_total, q_per_group;
  // key rows of unit b:  kbase(b) + ((k_start(g) + j) % k_mod) * k_tstride, j < win*k_per_group
  int k_inner; int64_t k_outer, k_istride, k_tstride; int k_mod, k_per_group, win;
  const int* shift_ptr;
  float scale;
};

__device__ __forceinline__ int64_t ab_qbase(const AttnBwdParams& p, int b) {
  return (int64_t)(b / p.q_inner) * p.q_outer + (int64_t)(b % p.q_inner) * p.q_istride;
}
__device__ __forceinline__ int64_t ab_kbase(const AttnBwdParams& p, int b) {
  return (int64_t)(b / p.k_inner) * p.k_outer + (int64_t)(b % p.k_inner) * p.k_istride;
}
__device__ __forceinline__ int ab_qcount(const AttnBwdParams& p, int g) {
  return g == p.G - 1 ? p.q_total - g * p.q_per_group : p.q_per_group;
}
__device__ __forceinline__ int ab_kstart(const AttnBwdParams& p, int g) {
  const int shift = p.shift_ptr ? *p.shift_ptr : 0;
  return (int)(((int64_t)(g + shift) * p.k_per_group) % p.k_mod);
}

// S block helper: thread (ty, tx) computes rows ty*4.., cols tx*4.. of A B^T for LDS tiles with leading dim LQ
__device__ __forceinline__ void tile_dot44(const float* sA, const float* sB, int LQ, int ch, int ty, int tx, float (&acc)[4][4]) {
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (int d = 0; d < ch; ++d) {
    float qa[4], kb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) qa[a] = sA[(ty * 4 + a) * LQ + d];
#pragma unroll
    for (int b = 0; b < 4; ++b) kb[b] = sB[(tx * 4 + b) * LQ + d];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += qa[a] * kb[b];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ch = p.ch, LQ = ch + 1;
  float* sQ = (float*)smem;            // [64][LQ] scaled q
  float* sdO = sQ + 64 * LQ;           // [64][LQ]
  float* sK = sdO + 64 * LQ;           // [64][LQ]
  float* sV = sK + 64 * LQ;            // [64][LQ]
  float* sS = sV + 64 * LQ;            // [64][65]
  float* sM = sS + 64 * 65;            // [64]
  float* sL = sM + 64;                 // [64]
  float* sD = sL + 64;                 // [64]
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int h = blockIdx.y;
  const int b = blockIdx.z / p.G, g = blockIdx.z % p.G;
  const int qcount = ab_qcount(p, g);
  const int q0 = blockIdx.x * 64;
  if (q0 >= qcount) return;
  const int64_t qb = ab_qbase(p, b), kb = ab_kbase(p, b);
  const int kstart = ab_kstart(p, g), kcount = p.win * p.k_per_group;
  auto qrow = [&](int i) { return qb + (int64_t)(g * p.q_per_group + q0 + i) * p.q_tstride; };
  auto krow = [&](int j) { int r = kstart + j; if (r >= p.k_mod) r -= p.k_mod; return kb + (int64_t)r * p.k_tstride; };

  for (int i = tid; i < 64 * ch; i += 256) {
    const int r = i / ch, d = i % ch;
    float q = 0.f, go = 0.f;
    if (q0 + r < qcount) {
      q = Elt<T>::ld(p.Q, qrow(r) * p.ldq + p.q_off + h * ch + d) * p.scale;
      go = Elt<T>::ld(p.dO, qrow(r) * p.lddo + h * ch + d);
    }
    sQ[r * LQ + d] = q;
    sdO[r * LQ + d] = go;
  }
  if (tid < 64) { sM[tid] = -1e30f; sL[tid] = 0.f; }
  __syncthreads();
  {   // D_i = dO_i . O_i   (4 threads per row)
    const int r = tid >> 2, part = tid & 3;
    float acc = 0.f;
    if (q0 + r < qcount)
      for (int d = part; d < ch; d += 4) acc += sdO[r * LQ + d] * Elt<T>::ld(p.O, qrow(r) * p.ldo + h * ch + d);
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) sD[r] = acc;
  }
  const int ntiles = (kcount + 63) >> 6;
  // ---- pass 1: running max / sum per query row
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    for (int i = tid; i < 64 * ch; i += 256) {
      const int r = i / ch, d = i % ch;
      sK[r * LQ + d] = (t * 64 + r < kcount) ? Elt<T>::ld(p.KV, krow(t * 64 + r) * p.ldkv + p.k_off + h * ch + d) : 0.f;
    }
    __syncthreads();
    float s[4][4];
    tile_dot44(sQ, sK, LQ, ch, ty, tx, s);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) sS[(ty * 4 + a) * 65 + tx * 4 + c] = (t * 64 + tx * 4 + c < kcount) ? s[a][c] : -1e30f;
    __syncthreads();
    const int r = tid >> 2, part = tid & 3;
    float mx = -1e30f;
    for (int k = part * 16; k < part * 16 + 16; ++k) mx = fmaxf(mx, sS[r * 65 + k]);
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    const float m_old = sM[r], m_new = fmaxf(m_old, mx);
    float sum = 0.f;
    for (int k = part * 16; k < part * 16 + 16; ++k) sum += __expf(sS[r * 65 + k] - m_new);
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    __syncthreads();
    if (part == 0) { sL[r] = sL[r] * __expf(m_old - m_new) + sum; sM[r] = m_new; }
  }
  __syncthreads();
  if (tid < 64) {
    const float l = sM[tid] + logf(sL[tid]);
    sM[tid] = l;                                        // sM now holds the log-sum-exp
    if (q0 + tid < qcount) {
      const int64_t qr = qrow(tid);
      p.lse[qr * p.heads + h] = l;
      p.dsum[qr * p.heads + h] = sD[tid];
    }
  }
  // ---- pass 2: dQ
  float dq[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 8; ++c) dq[a][c] = 0.f;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    for (int i = tid; i < 64 * ch; i += 256) {
      const int r = i / ch, d = i % ch;
      float kv = 0.f, vv = 0.f;
      if (t * 64 + r < kcount) {
        const int64_t row = krow(t * 64 + r);
        kv = Elt<T>::ld(p.KV, row * p.ldkv + p.k_off + h * ch + d);
        vv = Elt<T>::ld(p.KV, row * p.ldkv + p.v_off + h * ch + d);
      }
      sK[r * LQ + d] = kv;
      sV[r * LQ + d] = vv;
    }
    __syncthreads();
    float s[4][4], dp[4][4];
    tile_dot44(sQ, sK, LQ, ch, ty, tx, s);
    tile_dot44(sdO, sV, LQ, ch, ty, tx, dp);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = ty * 4 + a;
        const bool ok = t * 64 + tx * 4 + c < kcount;
        const float pr = ok ? __expf(s[a][c] - sM[r]) : 0.f;
        sS[r * 65 + tx * 4 + c] = pr * (dp[a][c] - sD[r]);
      }
    __syncthreads();
    for (int k = 0; k < 64; ++k) {
      float ds[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) ds[a] = sS[(ty * 4 + a) * 65 + k];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int d = tx + 16 * c;
        if (d < ch) {
          const float kv = sK[k * LQ + d];
#pragma unroll
          for (int a = 0; a < 4; ++a) dq[a][c] += ds[a] * kv;
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = ty * 4 + a;
    if (q0 + r < qcount) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int d = tx + 16 * c;
        if (d < ch) Elt<T>::st(p.dQ, qrow(r) * p.lddq + p.dq_off + h * ch + d, dq[a][c] * p.scale);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ch = p.ch, LQ = ch + 1;
  float* sQ = (float*)smem;            // [64][LQ] scaled q
  float* sdO = sQ + 64 * LQ;
  float* sK = sdO + 64 * LQ;
  float* sV = sK + 64 * LQ;
  float* sS = sV + 64 * LQ;            // [64 q][65]
  float* sLse = sS + 64 * 65;          // [64]
  float* sD = sLse + 64;               // [64]
  int* sFlag = (int*)(sD + 64);
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  const int k0 = blockIdx.x * 64;      // key index (within [0, k_mod)) of this tile
  if (k0 >= p.k_mod) return;
  const int64_t qb = ab_qbase(p, b), kb = ab_kbase(p, b);
  const int kcount = p.win * p.k_per_group;

  for (int i = tid; i < 64 * ch; i += 256) {
    const int r = i / ch, d = i % ch;
    float kv = 0.f, vv = 0.f;
    if (k0 + r < p.k_mod) {
      const int64_t row = kb + (int64_t)(k0 + r) * p.k_tstride;
      kv = Elt<T>::ld(p.KV, row * p.ldkv + p.k_off + h * ch + d);
      vv = Elt<T>::ld(p.KV, row * p.ldkv + p.v_off + h * ch + d);
    }
    sK[r * LQ + d] = kv;
    sV[r * LQ + d] = vv;
  }
  float dk[4][8], dv[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 8; ++c) dk[a][c] = dv[a][c] = 0.f;

  for (int g = 0; g < p.G; ++g) {
    const int kstart = ab_kstart(p, g);
    // does any key of this tile fall into the window [kstart, kstart + kcount) on the circle?
    __syncthreads();
    if (tid == 0) *sFlag = 0;
    __syncthreads();
    if (tid < 64 && k0 + tid < p.k_mod) {
      int rel = k0 + tid - kstart;
      if (rel < 0) rel += p.k_mod;
      if (rel < kcount) *sFlag = 1;
    }
    __syncthreads();
    if (*sFlag == 0) continue;
    const int qcount = ab_qcount(p, g);
    for (int q0 = 0; q0 < qcount; q0 += 64) {
      __syncthreads();
      for (int i = tid; i < 64 * ch; i += 256) {
        const int r = i / ch, d = i % ch;
        float q = 0.f, go = 0.f;
        if (q0 + r < qcount) {
          const int64_t row = qb + (int64_t)(g * p.q_per_group + q0 + r) * p.q_tstride;
          q = Elt<T>::ld(p.Q, row * p.ldq + p.q_off + h * ch + d) * p.scale;
          go = Elt<T>::ld(p.dO, row * p.lddo + h * ch + d);
        }
        sQ[r * LQ + d] = q;
        sdO[r * LQ + d] = go;
      }
      if (tid < 64) {
        float l = 0.f, dd = 0.f;
        if (q0 + tid < qcount) {
          const int64_t row = qb + (int64_t)(g * p.q_per_group + q0 + tid) * p.q_tstride;
          l = p.lse[row * p.heads + h];
          dd = p.dsum[row * p.heads + h];
        }
        sLse[tid] = l;
        sD[tid] = dd;
      }
      __syncthreads();
      float s[4][4], dp[4][4];
      tile_dot44(sQ, sK, LQ, ch, ty, tx, s);        // rows = queries, cols = keys of this tile
      tile_dot44(sdO, sV, LQ, ch, ty, tx, dp);
      float pr[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int r = ty * 4 + a, kk = k0 + tx * 4 + c;
          int rel = kk - kstart;
          if (rel < 0) rel += p.k_mod;
          const bool ok = (q0 + r < qcount) && kk < p.k_mod && rel < kcount;
          pr[a][c] = ok ? __expf(s[a][c] - sLse[r]) : 0.f;
          sS[r * 65 + tx * 4 + c] = pr[a][c];
        }
      __syncthreads();
      // dV[k][d] += sum_q P[q][k] dO[q][d]   (thread: keys ty*4+a, d = tx + 16c)
      for (int q = 0; q < 64; ++q) {
        float pv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) pv[a] = sS[q * 65 + ty * 4 + a];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int d = tx + 16 * c;
          if (d < ch) {
            const float go = sdO[q * LQ + d];
#pragma unroll
            for (int a = 0; a < 4; ++a) dv[a][c] += pv[a] * go;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int r = ty * 4 + a;
          sS[r * 65 + tx * 4 + c] = pr[a][c] * (dp[a][c] - sD[r]);
        }
      __syncthreads();
      // dK[k][d] += sum_q dS[q][k] (scale q)[q][d]
      for (int q = 0; q < 64; ++q) {
        float ds[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) ds[a] = sS[q * 65 + ty * 4 + a];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int d = tx + 16 * c;
          if (d < ch) {
            const float qv = sQ[q * LQ + d];
#pragma unroll
            for (int a = 0; a < 4; ++a) dk[a][c] += ds[a] * qv;
          }
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = ty * 4 + a;
    if (k0 + r < p.k_mod) {
      const int64_t row = kb + (int64_t)(k0 + r) * p.k_tstride;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int d = tx + 16 * c;
        if (d < ch) {
          Elt<T>::st(p.dKV, row * p.lddkv + p.dk_off + h * ch + d, dk[a][c]);
          Elt<T>::st(p.dKV, row * p.lddkv + p.dv_off + h * ch + d, dv[a][c]);
        }
      }
    }
  }
}

// Backward of mmd_attn_fwd / mmd_attn_small_fwd.  Row addressing: unit b (< nb) has query rows
//   qbase(b) + idx*q_tstride, qbase(b) = (b / q_inner)*q_outer + (b % q_inner)*q_istride, idx = g*q_per_group + i (< q_total)
// and key rows kbase(b) + ((k_start(g) + j) mod k_mod)*k_tstride, j < win*k_per_group (k_start as in mmd_attn_fwd).
// dQ is written for every query row; dK/dV are written (not accumulated) for every key row of every unit, so when Q and
// KV are the same buffer (self-attention) use separate column ranges: dq_off, dk_off, dv_off of one [rows, 3C] buffer.
// lse_ws / dsum_ws: fp32 [query row space * heads].
extern "C" int mmd_attn_bwd(int dtype, const void* Q, int64_t ldq, int q_off, const void* KV, int64_t ldkv, int k_off, int v_off,
                            const void* O, int64_t ldo, const void* dO, int64_t lddo, void* dQ, int64_t lddq, int dq_off, void* dKV,
                            int64_t lddkv, int dk_off, int dv_off, float* lse_ws, float* dsum_ws, int heads, int ch, int nb, int G,
                            int q_inner, int64_t q_outer, int64_t q_istride, int64_t q_tstride, int q_total, int q_per_group,
                            int k_inner, int64_t k_outer, int64_t k_istride, int64_t k_tstride, int k_mod, int k_per_group, int win,
                            const int* shift_dev, void* stream) {
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "attn_bwd: bad dtype");
  MMD_REQUIRE(Q && KV && O && dO && dQ && dKV && lse_ws && dsum_ws, "attn_bwd: null pointer");
  MMD_REQUIRE(heads > 0 && ch > 0 && ch <= 128 && nb > 0 && G > 0 && q_inner > 0 && k_inner > 0, "attn_bwd: bad geometry");
  MMD_REQUIRE((int64_t)win * k_per_group <= k_mod && (G - 1) * q_per_group < q_total, "attn_bwd: bad window / grouping");
  AttnBwdParams p;
  p.Q = (const char*)Q; p.ldq = ldq; p.q_off = q_off; p.KV = (const char*)KV; p.ldkv = ldkv; p.k_off = k_off; p.v_off = v_off;
  p.O = (const char*)O; p.ldo = ldo; p.dO = (const char*)dO; p.lddo = lddo;
  p.dQ = (char*)dQ; p.lddq = lddq; p.dq_off = dq_off; p.dKV = (char*)dKV; p.lddkv = lddkv; p.dk_off = dk_off; p.dv_off = dv_off;
  p.lse = lse_ws; p.dsum = dsum_ws; p.heads = heads; p.ch = ch; p.nb = nb; p.G = G;
  p.q_inner = q_inner; p.q_outer = q_outer; p.q_istride = q_istride; p.q_tstride = q_tstride; p.q_total = q_total; p.q_per_group = q_per_group;
  p.k_inner = k_inner; p.k_outer = k_outer; p.k_istride = k_istride; p.k_tstride = k_tstride; p.k_mod = k_mod; p.k_per_group = k_per_group;
  p.win = win; p.shift_ptr = shift_dev; p.scale = 1.0f / sqrtf((float)ch);
  const size_t lds = (size_t)(4 * 64 * (ch + 1) + 64 * 65 + 64 * 3 + 16) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const void* f1 = dtype == MMD_BF16 ? (const void*)attn_bwd_dq_kernel<__bf16> : (const void*)attn_bwd_dq_kernel<float>;
  const void* f2 = dtype == MMD_BF16 ? (const void*)attn_bwd_dkv_kernel<__bf16> : (const void*)attn_bwd_dkv_kernel<float>;
  if (hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess ||
      hipFuncSetAttribute(f2, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess)
    return mmd_set_error(MMD_ERR_LAUNCH, "attn_bwd: set LDS attr failed");
  const int qmax = q_total - (G - 1) * q_per_group;
  dim3 g1(cdiv(qmax, 64), heads, nb * G);
  if (dtype == MMD_BF16) hipLaunchKernelGGL(attn_bwd_dq_kernel<__bf16>, g1, dim3(256), lds, st, p);
  else hipLaunchKernelGGL(attn_bwd_dq_kernel<float>, g1, dim3(256), lds, st, p);
  int rc = mmd_check_launch("attn_bwd_dq");
  if (rc) return rc;
  dim3 g2(cdiv(k_mod, 64), heads, nb);
  if (dtype == MMD_BF16) hipLaunchKernelGGL(attn_bwd_dkv_kernel<__bf16>, g2, dim3(256), lds, st, p);
  else hipLaunchKernelGGL(attn_bwd_dkv_kernel<float>, g2, dim3(256), lds, st, p);
  return mmd_check_launch("attn_bwd_dkv");
}

// ============================================================================= short-sequence (temporal) backward
// Backward of attn_small_kernel (mmd_attn.hip): one wave per (slice, head), Tn <= 32 rows strided by tstride.
// Phase 1, lane = (query, channel quarter): s_j, dp_j = dO.v_j, P = softmax(s), D = sum_j P_j dp_j,
// dS_j = scale P_j (dp_j - D), dQ = dS K (written), P / dS parked in LDS.
// Phase 2, lane = (key, channel quarter): dK_j = sum_i dS_ij Q_i, dV_j = sum_i P_ij dO_i.
struct SmallAttnBwdParams {
  const char* QKV; int64_t ld;
  const char* dO; int64_t lddo;
  char* dQKV; int64_t ldd;
  int C, heads, ch;
  int S, Tn, inner;
  int64_t outer_stride, inner_stride, tstride;
  float scale;
};

template <typename T, int CHQ>
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const SmallAttnBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int ch = CHQ * 4;
  constexpr int EPV = Elt<T>::EPV, ES = 16 / EPV;
  const int Tn = p.Tn, TP = Tn + 1;
  const int per_wave = 4 * Tn * ch + 2 * Tn * TP;
  float* sQ = (float*)smem + (size_t)wave * per_wave;   // [Tn][ch] each
  float* sK = sQ + Tn * ch;
  float* sV = sK + Tn * ch;
  float* sG = sV + Tn * ch;                              // dO
  float* sP = sG + Tn * ch;                              // [Tn][Tn+1]
  float* sS = sP + Tn * TP;                              // dS (scale folded in)
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  const bool active = item < (int64_t)p.S * p.heads;
  const int s = active ? (int)(item / p.heads) : 0, h = active ? (int)(item % p.heads) : 0;
  const int64_t base = (int64_t)(s / p.inner) * p.outer_stride + (int64_t)(s % p.inner) * p.inner_stride;
  const bool vec_ok = (p.ld % EPV == 0) && (p.lddo % EPV == 0) && (p.ldd % EPV == 0) && (p.C % EPV == 0) && (ch % EPV == 0) &&
                      (((uintptr_t)p.QKV | (uintptr_t)p.dO | (uintptr_t)p.dQKV) % 16 == 0);
  if (active) {
    if (vec_ok) {
      constexpr int cvn = ch / EPV > 0 ? ch / EPV : 1;
      for (int i = lane; i < Tn * cvn; i += 64) {
        const int j = i / cvn, v = i % cvn;
        const int64_t row = base + (int64_t)j * p.tstride;
        const u32x4 xq = *(const u32x4*)(p.QKV + (row * p.ld + h * ch + v * EPV) * ES);
        const u32x4 xk = *(const u32x4*)(p.QKV + (row * p.ld + p.C + h * ch + v * EPV) * ES);
        const u32x4 xv = *(const u32x4*)(p.QKV + (row * p.ld + 2 * p.C + h * ch + v * EPV) * ES);
        const u32x4 xg = *(const u32x4*)(p.dO + (row * p.lddo + h * ch + v * EPV) * ES);
        float fq[EPV], fk[EPV], fv[EPV], fg[EPV];
        Elt<T>::unpack(xq, fq); Elt<T>::unpack(xk, fk); Elt<T>::unpack(xv, fv); Elt<T>::unpack(xg, fg);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          const int o = j * ch + v * EPV + e;
          sQ[o] = fq[e]; sK[o] = fk[e]; sV[o] = fv[e]; sG[o] = fg[e];
        }
      }
    } else {
      for (int i = lane; i < Tn * ch; i += 64) {
        const int j = i / ch, d = i % ch;
        const int64_t row = base + (int64_t)j * p.tstride;
        sQ[i] = Elt<T>::ld(p.QKV, row * p.ld + h * ch + d);
        sK[i] = Elt<T>::ld(p.QKV, row * p.ld + p.C + h * ch + d);
        sV[i] = Elt<T>::ld(p.QKV, row * p.ld + 2 * p.C + h * ch + d);
        sG[i] = Elt<T>::ld(p.dO, row * p.lddo + h * ch + d);
      }
    }
  }
  __syncthreads();
  if (!active) return;
  const int dq = lane & 3, c0 = dq * CHQ;
  auto store_row = [&](int64_t row, int col0, const float* o) {
    if (vec_ok && CHQ % EPV == 0) {
#pragma unroll
      for (int d = 0; d < CHQ; d += EPV) {
        float f[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) f[e] = o[(d + e) < CHQ ? (d + e) : 0];
        *(u32x4*)(p.dQKV + (row * p.ldd + col0 + c0 + d) * ES) = Elt<T>::pack(f);
      }
    } else {
#pragma unroll
      for (int d = 0; d < CHQ; ++d) Elt<T>::st(p.dQKV, row * p.ldd + col0 + c0 + d, o[d]);
    }
  };
  // ---- phase 1: per query
  for (int qb = 0; qb < Tn; qb += 16) {
    const int qi = qb + (lane >> 2);
    const bool ok = qi < Tn;
    const int qr = ok ? qi : 0;
    float q[CHQ], g[CHQ];
#pragma unroll
    for (int d = 0; d < CHQ; ++d) { q[d] = sQ[qr * ch + c0 + d] * p.scale; g[d] = sG[qr * ch + c0 + d]; }
    float sc[32], dp[32];
    float mx = -1e30f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      float a = 0.f, b = 0.f;
      if (j < Tn) {
#pragma unroll
        for (int d = 0; d < CHQ; ++d) { a += q[d] * sK[j * ch + c0 + d]; b += g[d] * sV[j * ch + c0 + d]; }
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64);
        b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64);
        mx = fmaxf(mx, a);
      }
      sc[j] = a; dp[j] = b;
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float e = j < Tn ? __expf(sc[j] - mx) : 0.f;
      sc[j] = e; sum += e;
    }
    const float inv = 1.f / sum;
    float Dq = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) { sc[j] *= inv; Dq += sc[j] * dp[j]; }
    float o[CHQ];
#pragma unroll
    for (int d = 0; d < CHQ; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (j < Tn) {
        const float ds = sc[j] * (dp[j] - Dq) * p.scale;
        if (ok && dq == (j & 3)) { sP[qi * TP + j] = sc[j]; sS[qi * TP + j] = ds; }
#pragma unroll
        for (int d = 0; d < CHQ; ++d) o[d] += ds * sK[j * ch + c0 + d];
      }
    }
    if (ok) store_row(base + (int64_t)qi * p.tstride, h * ch, o);
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): P / dS visible to the whole wave
  // ---- phase 2: per key
  for (int kb = 0; kb < Tn; kb += 16) {
    const int kj = kb + (lane >> 2);
    if (kj >= Tn) continue;
    float dk[CHQ], dv[CHQ];
#pragma unroll
    for (int d = 0; d < CHQ; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    for (int i = 0; i < Tn; ++i) {
      const float ds = sS[i * TP + kj], pr = sP[i * TP + kj];
#pragma unroll
      for (int d = 0; d < CHQ; ++d) { dk[d] += ds * sQ[i * ch + c0 + d]; dv[d] += pr * sG[i * ch + c0 + d]; }
    }
    const int64_t row = base + (int64_t)kj * p.tstride;
    store_row(row, p.C + h * ch, dk);
    store_row(row, 2 * p.C + h * ch, dv);
  }
}

template <typename T, int CHQ>
static int launch_small_bwd(const SmallAttnBwdParams& p, hipStream_t st) {
  const size_t lds = (size_t)4 * (4 * p.Tn * (CHQ * 4) + 2 * p.Tn * (p.Tn + 1)) * sizeof(float);
  if (lds > 64 * 1024) {
    static size_t attr = 0;
    if (lds > attr) {
      hipError_t e = hipFuncSetAttribute((const void*)attn_small_bwd_kernel<T, CHQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "attn_small_bwd: set LDS attr: %s", hipGetErrorString(e));
      attr = lds;
    }
  }
  const int64_t items = (int64_t)p.S * p.heads;
  hipLaunchKernelGGL((attn_small_bwd_kernel<T, CHQ>), dim3((unsigned)((items + 3) / 4)), dim3(256), lds, st, p);
  return mmd_check_launch("attn_small_bwd");
}

template <typename T>
static int dispatch_small_bwd(const SmallAttnBwdParams& p, hipStream_t st) {
  switch (p.ch) {
    case 16: return launch_small_bwd<T, 4>(p, st);
    case 32: return launch_small_bwd<T, 8>(p, st);
    case 48: return launch_small_bwd<T, 12>(p, st);
    case 64: return launch_small_bwd<T, 16>(p, st);
    case 96: return launch_small_bwd<T, 24>(p, st);
    case 128: return launch_small_bwd<T, 32>(p, st);
    default: return mmd_set_error(MMD_ERR_UNSUPPORTED, "attn_small_bwd: head width %d not in {16,32,48,64,96,128}", p.ch);
  }
}

// Backward of mmd_attn_small_fwd: dQKV rows get [dq | dk | dv] for the same slice geometry.
extern "C" int mmd_attn_small_bwd(int dtype, const void* QKV, int64_t ld, const void* dO, int64_t lddo, void* dQKV, int64_t ldd,
                                  int C, int heads, int S, int Tn, int inner, int64_t outer_stride, int64_t inner_stride,
                                  int64_t tstride, void* stream) {
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "attn_small_bwd: bad dtype %d", dtype);
  MMD_REQUIRE(QKV && dO && dQKV && C > 0 && heads > 0 && C % heads == 0, "attn_small_bwd: bad argument");
  MMD_REQUIRE(Tn >= 1 && Tn <= 32, "attn_small_bwd: sequence length %d not in [1,32]", Tn);
  SmallAttnBwdParams p;
  p.QKV = (const char*)QKV; p.ld = ld; p.dO = (const char*)dO; p.lddo = lddo; p.dQKV = (char*)dQKV; p.ldd = ldd;
  p.C = C; p.heads = heads; p.ch = C / heads; p.S = S; p.Tn = Tn; p.inner = inner;
  p.outer_stride = outer_stride; p.inner_stride = inner_stride; p.tstride = tstride;
  p.scale = 1.0f / sqrtf((float)p.ch);
  hipStream_t st = (hipStream_t)stream;
  return dtype == MMD_BF16 ? dispatch_small_bwd<__bf16>(p, st) : dispatch_small_bwd<float>(p, st);
}
