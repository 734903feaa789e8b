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
