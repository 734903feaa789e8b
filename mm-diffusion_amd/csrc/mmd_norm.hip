// GroupNorm32 (+FiLM) (+SiLU) on channels-last activations, native layout - no permute copies.
//
// Replaces reference nn.py:16-33 (GroupNorm32: 32 groups, eps 1e-5, fp32 statistics, the
// 'b t c h w -> b c t h w' round trip), nn.SiLU and the FiLM modulation norm(h)*(1+scale)+shift of
// ResBlock._forward (multimodal_unet.py:457-470).
//
// A "slice" is the set of rows one GroupNorm instance normalises over:
//   rows(s) = base(s) + j*tstride, j < Tn,  base(s) = (s / inner)*outer_stride + (s % inner)*inner_stride
//   per-sample video/audio GN : S=N,     inner=1,  outer_stride=rows/sample, tstride=1, Tn=rows/sample
//   spatial self-attn GN      : S=N*F,   inner=1,  outer_stride=HW,          tstride=1, Tn=HW
//   temporal self-attn GN     : S=N*HW,  inner=HW, outer_stride=F*HW, inner_stride=1, tstride=HW, Tn=F
// Stage 1 (gn_partial): per (chunk of R rows, slice) per-thread fp32 partial sums of pivot-shifted data, combined per
//   group in fp64 in a fixed order (deterministic - no atomics) -> per-group (sum, sumsq) doubles.
// Stage 2 (gn_finalize): mean / rstd per (slice, group) and the fused affine
//   a[s,c] = rstd*gamma[c]*(1+scale[s,c]),  b[s,c] = (beta[c]-mean*rstd*gamma[c])*(1+scale[s,c]) + shift[s,c]
// Stage 3 (gn_apply): y = act(x*a + b), 16-byte vector loads/stores (HBM-bound, 2 bytes moved per byte read).
#include "mmd_common.h"

#define GN_GROUPS 32
typedef __attribute__((ext_vector_type(2))) double f64x2;

struct SliceGeom {
  int S, Tn, inner;
  int64_t outer_stride, inner_stride, tstride;
};

__device__ __forceinline__ int64_t slice_base(const SliceGeom& g, int s) {
  return (int64_t)(s / g.inner) * g.outer_stride + (int64_t)(s % g.inner) * g.inner_stride;
}
__device__ __forceinline__ int slice_of_row(const SliceGeom& g, int64_t m) {
  const int64_t o = m / g.outer_stride, rem = m % g.outer_stride;
  return (int)(o * g.inner + (rem / g.inner_stride) % g.inner);
}

// Rows per block R is chosen on the host (gn_rows_per_block) so that even a 4-slice, 400-row GroupNorm fans
// out over ~100 blocks and every thread keeps 4 independent 16-byte loads in flight (the first version ran
// 256-row blocks with one load in flight: 56-800 GB/s).  ONE = true: one block owns the whole slice and
// writes the fused affine directly (no partials, no second launch).
template <typename T, bool ONE>
__global__ __launch_bounds__(256) void gn_partial_kernel(const char* __restrict__ x, int64_t ld, int C, SliceGeom g, int R,
                                                         double* __restrict__ part, int nchunks,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ film, int64_t film_ld, float eps,
                                                         float* __restrict__ a_out, float* __restrict__ b_out, float* __restrict__ mr_out) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  __shared__ float s_sum[2048];           // [rows per pass][C], C <= 2048
  __shared__ float s_sq[2048];
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS];
  const int s = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int CV = C / EPV;                 // vecs per row; more than 256 (fp32 rows wider than 1024 channels - the SR U-Net's 1536-channel
  const int CVB = min(CV, 256);           // skip concatenations in fp32 mode) are walked in column passes of 256 vectors, one row lane
  const int RPP = 256 / CVB;              // rows per pass
  const int col0 = tid % CVB, rl = tid / CVB;
  const int j0 = chunk * R;
  const int j1 = min(j0 + R, g.Tn);
  const int64_t base = slice_base(g, s);
  // shifted-data sums: pivot = first element of the group in the slice's first row (kills the
  // E[x^2]-E[x]^2 cancellation when a group carries a large common offset)
  const int cpg = C / GN_GROUPS;
  for (int cp = 0; cp < CV; cp += 256) {
  const int col = col0 + cp;
  if (col >= CV) break;
  float sum[EPV], sq[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) sum[j] = sq[j] = 0.f;
  float piv[EPV];
  {
    const int c0 = col * EPV;
    const int g0 = c0 / cpg;
    int gc = g0, rem = c0 - g0 * cpg;
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      piv[e] = Elt<T>::ld(x, base * ld + (int64_t)gc * cpg);
      if (++rem == cpg) { rem = 0; ++gc; }
    }
  }
  if (rl < RPP) {
    const char* xp = x + (base * ld + (int64_t)col * EPV) * ES;
    const int64_t rstride = g.tstride * ld * ES;
    for (int jb = j0 + rl; jb < j1; jb += 4 * RPP) {
      u32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = jb + u * RPP;
        if (j < j1) v[u] = *(const u32x4*)(xp + (int64_t)j * rstride);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (jb + u * RPP < j1) {
          float f[EPV];
          Elt<T>::unpack(v[u], f);
#pragma unroll
          for (int e = 0; e < EPV; ++e) { const float d = f[e] - piv[e]; sum[e] += d; sq[e] += d * d; }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      s_sum[rl * C + col * EPV + e] = sum[e];
      s_sq[rl * C + col * EPV + e] = sq[e];
    }
  }
  }
  __syncthreads();
  // Group combine straight from the per-thread fp32 sums: a group owns RPP*cpg <= 64 of them (RPP*C/32 <= 8*EPV), eight
  // threads per group sum <= 8 each in double and fold with a fixed xor tree (deterministic; no per-channel staging in LDS:
  // the earlier version held 32 KB of doubles for it, which capped the kernel at 3 blocks per CU).
  {
    const int gi = tid >> 3, k = tid & 7;
    const int n = RPP * cpg;
    double a = 0.0, b = 0.0;
    for (int e = k; e < n; e += 8) {
      const int r = e / cpg, c = gi * cpg + (e - r * cpg);
      a += (double)s_sum[r * C + c];
      b += (double)s_sq[r * C + c];
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      a += __shfl_xor(a, m);
      b += __shfl_xor(b, m);
    }
    if (k == 0) {
      if (!ONE) {
        double* o = part + (((int64_t)s * nchunks + chunk) * GN_GROUPS + gi) * 2;
        o[0] = a;
        o[1] = b;
      } else {
        const double cnt = (double)g.Tn * (double)cpg;
        const double piv0 = (double)Elt<T>::ld(x, base * ld + (int64_t)gi * cpg);
        const double dm = a / cnt;
        double var = b / cnt - dm * dm;
        if (var < 0.0) var = 0.0;
        s_mean[gi] = (float)(piv0 + dm);
        s_rstd[gi] = (float)(1.0 / sqrt(var + (double)eps));
        if (mr_out) {
          mr_out[((int64_t)s * GN_GROUPS + gi) * 2] = s_mean[gi];
          mr_out[((int64_t)s * GN_GROUPS + gi) * 2 + 1] = s_rstd[gi];
        }
      }
    }
  }
  if (ONE) {
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      const int gi = c / cpg;
      float a = s_rstd[gi] * gamma[c];
      float b = beta[c] - s_mean[gi] * a;
      if (film) {
        const float sc = 1.f + film[(int64_t)s * film_ld + c];
        const float sh = film[(int64_t)s * film_ld + C + c];
        a *= sc;
        b = b * sc + sh;
      }
      a_out[(int64_t)s * C + c] = a;
      b_out[(int64_t)s * C + c] = b;
    }
  }
}

__global__ __launch_bounds__(1024) void gn_finalize_kernel(const char* __restrict__ x, int dtype, int64_t ld, SliceGeom g,
                                                           const double* __restrict__ part, int nchunks, int C, int Tn,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ film, int64_t film_ld, float eps,
                                                           float* __restrict__ a_out, float* __restrict__ b_out, float* __restrict__ mr_out) {
  // 1024 threads = 32 chunk lanes x 32 groups: the kernel runs on S (= batch) blocks and is a chain of memory round trips, so
  // the partials of a slice (up to 320 chunks) are fetched in one or two rounds of independent 16-byte loads per thread
  // (the 256-thread version walked 40 chunks per thread: 6.8 us per call, 165 calls per denoising step).
  __shared__ double s_pa[32][GN_GROUPS + 1], s_pb[32][GN_GROUPS + 1];
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int cpg = C / GN_GROUPS;
  // Everything that does not depend on the partials is requested first (pivot, gamma / beta / FiLM of this thread's channels)
  constexpr int CPT = 2048 / 1024;        // channels per thread, C <= 2048
  float gm[CPT], bt[CPT], fsc[CPT], fsh[CPT];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int c = tid + k * 1024;
    if (c < C) {
      gm[k] = gamma[c];
      bt[k] = beta[c];
      fsc[k] = film ? 1.f + film[(int64_t)s * film_ld + c] : 1.f;
      fsh[k] = film ? film[(int64_t)s * film_ld + C + c] : 0.f;
    }
  }
  double piv = 0.0;
  if (tid < GN_GROUPS) {
    const int64_t pidx = slice_base(g, s) * ld + (int64_t)tid * cpg;
    piv = dtype == MMD_BF16 ? (double)Elt<__bf16>::ld(x, pidx) : (double)Elt<float>::ld(x, pidx);
  }
  {
    const int gi = tid & 31, cl = tid >> 5;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    const double* p0 = part + ((int64_t)s * nchunks * GN_GROUPS + gi) * 2;
    for (int k = cl; k < nchunks; k += 128) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {        // branch-free: clamped address, masked value -> the four loads issue together
        const int kk = k + 32 * u;
        const f64x2 v = *(const f64x2*)(p0 + (int64_t)min(kk, nchunks - 1) * GN_GROUPS * 2);
        const double m = kk < nchunks ? 1.0 : 0.0;
        a[u] += m * v[0];
        b[u] += m * v[1];
      }
    }
    s_pa[cl][gi] = (a[0] + a[1]) + (a[2] + a[3]);
    s_pb[cl][gi] = (b[0] + b[1]) + (b[2] + b[3]);
  }
  __syncthreads();
  if (tid < GN_GROUPS) {
    double a = 0.0, b = 0.0;
    for (int k = 0; k < 32; ++k) { a += s_pa[k][tid]; b += s_pb[k][tid]; }
    const double cnt = (double)Tn * (double)cpg;
    const double dm = a / cnt;
    double var = b / cnt - dm * dm;
    if (var < 0.0) var = 0.0;
    s_mean[tid] = (float)(piv + dm);
    s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    if (mr_out) {
      mr_out[((int64_t)s * GN_GROUPS + tid) * 2] = s_mean[tid];
      mr_out[((int64_t)s * GN_GROUPS + tid) * 2 + 1] = s_rstd[tid];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int c = tid + k * 1024;
    if (c < C) {
      const int gi = c / cpg;
      const float a = s_rstd[gi] * gm[k];
      const float b = bt[k] - s_mean[gi] * a;
      a_out[(int64_t)s * C + c] = a * fsc[k];
      b_out[(int64_t)s * C + c] = b * fsc[k] + fsh[k];
    }
  }
}

// Block = (row chunk, slice); thread = (channel vector, row lane): the fused affine of the thread's 4/8 channels sits in
// registers for the whole chunk (the flat version re-read 64 B of a/b per 16 B of data through L1), four rows in flight.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const char* __restrict__ x, int64_t ldx, char* __restrict__ y, int64_t ldy,
                                                       int C, SliceGeom g, const float* __restrict__ a,
                                                       const float* __restrict__ b, int act, int R) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  const int s = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int CV = C / EPV, CVB = min(CV, 256), RPP = 256 / CVB;      // CV > 256: column passes (see gn_partial_kernel)
  const int col0 = tid % CVB, rl = tid / CVB;
  if (rl >= RPP) return;
  for (int cp = 0; cp < CV; cp += 256) {
  const int col = col0 + cp;
  if (col >= CV) break;
  float av[EPV], bv[EPV];
#pragma unroll
  for (int e = 0; e < EPV; e += 4) {
    const f32x4 a4 = *(const f32x4*)(a + (int64_t)s * C + col * EPV + e), b4 = *(const f32x4*)(b + (int64_t)s * C + col * EPV + e);
#pragma unroll
    for (int k = 0; k < 4; ++k) { av[e + k] = a4[k]; bv[e + k] = b4[k]; }
  }
  const int64_t base = slice_base(g, s);
  const char* xp = x + (base * ldx + (int64_t)col * EPV) * ES;
  char* yp = y + (base * ldy + (int64_t)col * EPV) * ES;
  const int64_t xs = g.tstride * ldx * ES, ys = g.tstride * ldy * ES;
  const int j0 = chunk * R, j1 = min(j0 + R, g.Tn);
  for (int jb = j0 + rl; jb < j1; jb += 4 * RPP) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = jb + u * RPP;
      if (j < j1) v[u] = *(const u32x4*)(xp + (int64_t)j * xs);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = jb + u * RPP;
      if (j < j1) {
        float f[EPV];
        Elt<T>::unpack(v[u], f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          const float w = f[e] * av[e] + bv[e];
          f[e] = act ? silu_f(w) : w;
        }
        *(u32x4*)(yp + (int64_t)j * ys) = Elt<T>::pack(f);
      }
    }
  }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// One-pass GroupNorm32 (+SiLU) for SHORT slices (Tn <= 16 rows: the temporal-attention norm over the 16 frames of a pixel,
// unet:489-490 -> nn.py:16-33): statistics pass + finalize + apply cost three launches and read the tensor twice; here a thread owns
// ONE 16-byte channel vector of ONE slice for all of its rows (16 vectors in registers, packed), so the tensor is read once and the
// statistics are exact two-pass ones (mean first, then sum (x - mean)^2 - the data never leaves the registers).  A group is cpg
// channels x Tn rows: the thread folds its two QUADS of channels (every group size of the model is a multiple of 4), the quads go
// through LDS, and each thread sums the quads of the (at most two) groups it needs - two barriers per block, no atomics.
// Block = SPB slices x C / EPV vectors (<= 256 threads).
template <typename T>
__global__ __launch_bounds__(256) void gn_small_kernel(const char* __restrict__ x, int64_t ldx, char* __restrict__ y, int64_t ldy, int C,
                                                       SliceGeom g, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int act, int SPB) {
  constexpr int EPV = Elt<T>::EPV, ES = 16 / EPV, MAXT = 16, QPV = EPV / 4;   // quads per vector: 2 (bf16) / 1 (fp32)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sQ = (float*)smem;                       // [SPB][C / 4] quad partials (pass 1: sums, pass 2: centred sums of squares)
  const int CV = C / EPV, NQ = C / 4, cpg = C / GN_GROUPS, qpg = cpg / 4;
  const int tid = threadIdx.x, pl = tid / CV, cv = tid - pl * CV;
  const int s = blockIdx.x * SPB + pl;
  const bool live = pl < SPB && s < g.S;
  u32x4 v[MAXT];
  const int64_t base = live ? slice_base(g, s) : 0;
  const char* xp = x + (base * ldx + (int64_t)cv * EPV) * ES;
  const int64_t xs = g.tstride * ldx * ES;
#pragma unroll
  for (int j = 0; j < MAXT; ++j)
    if (live && j < g.Tn) v[j] = *(const u32x4*)(xp + (int64_t)j * xs);
  float q[2] = {0.f, 0.f};
  if (live) {
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
      if (j < g.Tn) {
        float f[EPV];
        Elt<T>::unpack(v[j], f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) q[e / 4] += f[e];
      }
#pragma unroll
    for (int h = 0; h < QPV; ++h) sQ[pl * NQ + cv * QPV + h] = q[h];
  }
  __syncthreads();
  // the groups of this thread's quads: quad k belongs to group k / qpg; a vector touches at most two groups
  const float inv_cnt = 1.f / ((float)cpg * (float)g.Tn);
  float mean[2] = {0.f, 0.f};
  if (live) {
#pragma unroll
    for (int h = 0; h < QPV; ++h) {
      const int gi = (cv * QPV + h) / qpg;
      float a = 0.f;
      for (int k = 0; k < qpg; ++k) a += sQ[pl * NQ + gi * qpg + k];
      mean[h] = a * inv_cnt;
    }
  }
  __syncthreads();
  if (live) {
    q[0] = q[1] = 0.f;
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
      if (j < g.Tn) {
        float f[EPV];
        Elt<T>::unpack(v[j], f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) { const float d = f[e] - mean[e / 4]; q[e / 4] += d * d; }
      }
#pragma unroll
    for (int h = 0; h < QPV; ++h) sQ[pl * NQ + cv * QPV + h] = q[h];
  }
  __syncthreads();
  if (live) {
    float av[EPV], bv[EPV];
#pragma unroll
    for (int h = 0; h < QPV; ++h) {
      const int gi = (cv * QPV + h) / qpg;
      float a = 0.f;
      for (int k = 0; k < qpg; ++k) a += sQ[pl * NQ + gi * qpg + k];
      const float rstd = rsqrtf(a * inv_cnt + eps);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = cv * EPV + h * 4 + e;
        av[h * 4 + e] = rstd * gamma[c];
        bv[h * 4 + e] = beta[c] - mean[h] * av[h * 4 + e];
      }
    }
    char* yp = y + (base * ldy + (int64_t)cv * EPV) * ES;
    const int64_t ys = g.tstride * ldy * ES;
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
      if (j < g.Tn) {
        float f[EPV];
        Elt<T>::unpack(v[j], f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          const float w = f[e] * av[e] + bv[e];
          f[e] = act ? silu_f(w) : w;
        }
        *(u32x4*)(yp + (int64_t)j * ys) = Elt<T>::pack(f);
      }
  }
}

// x[m, c] += e[n(m), c]   (non-FiLM ResBlock: h + emb_out, multimodal_unet.py:473-477)
template <typename T>
__global__ __launch_bounds__(256) void add_rowbias_kernel(char* __restrict__ x, int64_t ld, int64_t rows, int C,
                                                          int64_t rows_per_sample, const float* __restrict__ e, int64_t e_ld) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  const int CV = C / EPV;
  const int64_t total = rows * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / CV;
    const int cvi = (int)(i % CV);
    const int64_t n = m / rows_per_sample;
    char* p = x + (m * ld + (int64_t)cvi * EPV) * ES;
    float f[EPV];
    Elt<T>::unpack(*(const u32x4*)p, f);
#pragma unroll
    for (int k = 0; k < EPV; ++k) f[k] += e[n * e_ld + cvi * EPV + k];
    *(u32x4*)p = Elt<T>::pack(f);
  }
}

static int check_geom(const char* who, int dtype, int C, int S, int Tn, int inner) {
  const int epv = dtype == MMD_BF16 ? 8 : 4;
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "%s: bad dtype %d", who, dtype);
  MMD_REQUIRE(C % GN_GROUPS == 0 && C % epv == 0 && C <= 2048, "%s: unsupported channel count %d", who, C);
  MMD_REQUIRE(S > 0 && Tn > 0 && inner > 0, "%s: empty slice geometry", who);
  return MMD_OK;
}

// rows per stage-1 block: >= 4 rows per thread (4 loads in flight), grown until the grid is <= ~4096 blocks
static int gn_rows_per_block(int dtype, int C, int S, int Tn) {
  const int cv = C / (dtype == MMD_BF16 ? 8 : 4);
  const int rpp = 256 / cv > 0 ? 256 / cv : 1;
  static const int cap = [] { const char* e = getenv("MMD_GN_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1280; }();
  // R depends on the slice (Tn, C) only, never on the number of slices: the per-thread fp32 partial sums - and with them the last
  // bit of the statistics - must not change with the batch size (a batch-4 run equals four batch-1 runs bitwise).  320 chunks per
  // slice = one resident wave of blocks (5/CU x 256 CUs) at the 4 slices of the headline batch; MMD_GN_BLOCKS: tuning (x4).
  (void)S;
  int R = 4 * rpp;
  while (cdiv(Tn, R) > cap / 4 && R < 1024) R *= 2;
  return R;
}

extern "C" int64_t mmd_gn_workspace_bytes(int dtype, int C, int S, int Tn) {
  if (C <= 0 || S <= 0 || Tn <= 0) return 0;
  return (int64_t)S * cdiv(Tn, gn_rows_per_block(dtype, C, S, Tn)) * GN_GROUPS * 2 * sizeof(double);
}

extern "C" int mmd_gn_stats(int dtype, const void* x, int64_t ld, int C, int S, int Tn, int inner, int64_t outer_stride,
                            int64_t inner_stride, int64_t tstride, const float* gamma, const float* beta,
                            const float* film, int64_t film_ld, float eps, float* a_out, float* b_out, float* mr_out,
                            void* workspace, void* stream) {
  int rc = check_geom("gn_stats", dtype, C, S, Tn, inner);
  if (rc) return rc;
  MMD_REQUIRE(x && gamma && beta && a_out && b_out, "gn_stats: null pointer");
  MMD_REQUIRE(((uintptr_t)x) % 16 == 0 && ld % (dtype == MMD_BF16 ? 8 : 4) == 0, "gn_stats: x must be 16-byte aligned rows");
  SliceGeom g{S, Tn, inner, outer_stride, inner_stride, tstride};
  hipStream_t st = (hipStream_t)stream;
  const int R = gn_rows_per_block(dtype, C, S, Tn);
  const int nchunks = cdiv(Tn, R);
  if (nchunks == 1) {
    if (dtype == MMD_BF16)
      hipLaunchKernelGGL((gn_partial_kernel<__bf16, true>), dim3(1, S), dim3(256), 0, st, (const char*)x, ld, C, g, R, (double*)nullptr, 1,
                         gamma, beta, film, film_ld, eps, a_out, b_out, mr_out);
    else
      hipLaunchKernelGGL((gn_partial_kernel<float, true>), dim3(1, S), dim3(256), 0, st, (const char*)x, ld, C, g, R, (double*)nullptr, 1,
                         gamma, beta, film, film_ld, eps, a_out, b_out, mr_out);
    return mmd_check_launch("gn_stats_one");
  }
  MMD_REQUIRE(workspace && ((uintptr_t)workspace) % 16 == 0, "gn_stats: 16-byte aligned workspace required for multi-block slices");
  dim3 grid(nchunks, S);
  if (dtype == MMD_BF16)
    hipLaunchKernelGGL((gn_partial_kernel<__bf16, false>), grid, dim3(256), 0, st, (const char*)x, ld, C, g, R, (double*)workspace, nchunks,
                       gamma, beta, film, film_ld, eps, a_out, b_out, (float*)nullptr);
  else
    hipLaunchKernelGGL((gn_partial_kernel<float, false>), grid, dim3(256), 0, st, (const char*)x, ld, C, g, R, (double*)workspace, nchunks,
                       gamma, beta, film, film_ld, eps, a_out, b_out, (float*)nullptr);
  rc = mmd_check_launch("gn_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(S), dim3(1024), 0, st, (const char*)x, dtype, ld, g, (const double*)workspace, nchunks, C, Tn, gamma, beta,
                     film, film_ld, eps, a_out, b_out, mr_out);
  return mmd_check_launch("gn_finalize");
}

// ---- GroupNorm finalize from PRODUCER-side statistics (mmd_conv_gemm_stats): rec[(row / 64) * rec_ld + q] = (sum, sum of squares) of
// the stored values of 64 consecutive rows of the channel QUAD q (4 channels; per column until round 3).  One block per (group, slice): the slice's Tn / 64 records x cpg channels
// are summed in double in a fixed order (thread-strided, then a fixed tree), var = E[x^2] - mean^2 in double (no pivot: the records
// carry plain sums; the sums themselves are exact to fp32 rounding of <= 64-term partials, so the cancellation costs
// (1 + mean^2 / var) x 1e-7 relative - fine for conv outputs; the engine uses this path in bf16 mode only).
__global__ __launch_bounds__(256) void gn_finalize_rec_kernel(const float* __restrict__ rec, int64_t rec_ld, int C, int S, int Tn,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ film, int64_t film_ld, float eps,
                                                              float* __restrict__ a_out, float* __restrict__ b_out, float* __restrict__ mr_out) {
  __shared__ double s_a[4], s_b[4];
  const int gi = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / GN_GROUPS, qpg = cpg / 4;           // records are per QUAD of 4 channels (round 3)
  const int nrec = Tn / 64;
  const float* base = rec + ((int64_t)s * nrec * rec_ld + (int64_t)gi * qpg) * 2;
  const int total = nrec * qpg;
  // round 5: the affine / FiLM operands are requested BEFORE the records (they do not depend on the reduction): the launch is a chain of
  // memory round trips (kernel arguments -> records -> gamma / beta / film -> store) and this takes one of them out; the (mean, rstd) of
  // the group is computed by every writer thread itself instead of one thread + a second barrier (same operations, same bits)
  const int c = gi * cpg + min(tid, cpg - 1);
  const float gv = gamma[c], bv0 = beta[c];
  float fsc = 0.f, fsh = 0.f;
  if (film) {
    fsc = film[(int64_t)s * film_ld + c];
    fsh = film[(int64_t)s * film_ld + C + c];
  }
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = tid; i < total; i += 1024) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {          // branch-free: clamped address, masked value -> the four loads issue together
      const int k = i + 256 * u;
      const int kk = min(k, total - 1);
      const int r = kk / qpg, cq = kk - r * qpg;
      const float2 v = *(const float2*)(base + ((int64_t)r * rec_ld + cq) * 2);
      const double m = k < total ? 1.0 : 0.0;
      a[u] += m * (double)v.x;
      b[u] += m * (double)v.y;
    }
  }
  double ta = (a[0] + a[1]) + (a[2] + a[3]), tb = (b[0] + b[1]) + (b[2] + b[3]);
  ta = wave_sum_d(ta);
  tb = wave_sum_d(tb);
  if ((tid & 63) == 0) { s_a[tid >> 6] = ta; s_b[tid >> 6] = tb; }
  __syncthreads();
  if (tid < cpg) {
    const double sa = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]), sb = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
    const double cnt = (double)Tn * (double)cpg;
    const double mean = sa / cnt;
    double var = sb / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float mean_f = (float)mean, rstd_f = (float)(1.0 / sqrt(var + (double)eps));
    if (mr_out && tid == 0) {
      mr_out[((int64_t)s * GN_GROUPS + gi) * 2] = mean_f;
      mr_out[((int64_t)s * GN_GROUPS + gi) * 2 + 1] = rstd_f;
    }
    float av = rstd_f * gv;
    float bv = bv0 - mean_f * av;
    if (film) {
      const float sc = 1.f + fsc;
      av *= sc;
      bv = bv * sc + fsh;
    }
    a_out[(int64_t)s * C + c] = av;
    b_out[(int64_t)s * C + c] = bv;
  }
}

// GroupNorm32 fused affine from producer-side statistics: S contiguous slices of Tn rows (Tn % 64 == 0; slice s = records
// [s Tn / 64, (s + 1) Tn / 64)), rec / rec_ld as written by mmd_conv_gemm_stats.  Outputs as mmd_gn_stats.
extern "C" int mmd_gn_finalize_stats(const float* rec, int64_t rec_ld, int C, int S, int Tn, const float* gamma, const float* beta,
                                     const float* film, int64_t film_ld, float eps, float* a_out, float* b_out, float* mr_out,
                                     void* stream) {
  MMD_REQUIRE(rec && gamma && beta && a_out && b_out, "gn_finalize_stats: null pointer");
  MMD_REQUIRE(C > 0 && C % (4 * GN_GROUPS) == 0 && C / GN_GROUPS <= 256 && rec_ld >= C / 4,
              "gn_finalize_stats: channel count %d (groups of whole quads: a multiple of 128; ld %ld quads)", C, (long)rec_ld);
  MMD_REQUIRE(S > 0 && Tn > 0 && Tn % 64 == 0, "gn_finalize_stats: slices must be multiples of 64 rows (S=%d Tn=%d)", S, Tn);
  hipLaunchKernelGGL(gn_finalize_rec_kernel, dim3(GN_GROUPS, S), dim3(256), 0, (hipStream_t)stream, rec, rec_ld, C, S, Tn, gamma, beta,
                     film, film_ld, eps, a_out, b_out, mr_out);
  return mmd_check_launch("gn_finalize_stats");
}

// ---- GroupNorm32(+FiLM)(+SiLU) of slices of a few hundred rows in ONE launch: block = one (group, slice), its Tn x cpg elements stay in
// registers (at most 16 quads of 4 channels per thread), exact two-pass statistics (mean, then centred squares) in fp32 with fixed-order
// block reductions, then either only the fused affine (a, b) - the consumer is a GEMM that normalises in its loader - or the
// normalised tensor too.  For slices whose rows are not a multiple of the 64-row producer records (the 400-row audio samples at ds8:
// mmd_gn_stats there is a partial launch + a finalize launch, then mmd_gn_apply).  The apply step is gn_apply's arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void gn_group_kernel(const char* __restrict__ x, int64_t ldx, char* __restrict__ y, int64_t ldy, int C,
                                                       SliceGeom g, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ film, int64_t film_ld, float eps, int act,
                                                       float* __restrict__ a_out, float* __restrict__ b_out, float* __restrict__ mr_out) {
  constexpr int ES = sizeof(T), MAXI = 16;
  __shared__ float s_red[2][4];
  __shared__ float s_ab[2][64];
  const int gi = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / GN_GROUPS, qpg = cpg / 4, items = g.Tn * qpg;
  const int64_t base = slice_base(g, s);
  const char* xg = x + (int64_t)gi * cpg * ES;
  float f[MAXI][4];
#pragma unroll
  for (int j = 0; j < MAXI; ++j) {
    const int i = tid + 256 * j;
    if (i < items) {
      const int r = i / qpg, k = i - r * qpg;
      const char* q = xg + ((base + (int64_t)r * g.tstride) * ldx + 4 * k) * ES;
      if constexpr (ES == 2) {
        const u32x2 v = *(const u32x2*)q;
        f[j][0] = __uint_as_float(v[0] << 16); f[j][1] = __uint_as_float(v[0] & 0xffff0000u);
        f[j][2] = __uint_as_float(v[1] << 16); f[j][3] = __uint_as_float(v[1] & 0xffff0000u);
      } else {
        const f32x4 v = *(const f32x4*)q;
        f[j][0] = v[0]; f[j][1] = v[1]; f[j][2] = v[2]; f[j][3] = v[3];
      }
    } else {
      f[j][0] = f[j][1] = f[j][2] = f[j][3] = 0.f;
    }
  }
  auto block_sum = [&](float v, int slot) {
    v = wave_sum(v);
    if ((tid & 63) == 0) s_red[slot][tid >> 6] = v;
    __syncthreads();
    return (s_red[slot][0] + s_red[slot][1]) + (s_red[slot][2] + s_red[slot][3]);
  };
  float t = 0.f;
#pragma unroll
  for (int j = 0; j < MAXI; ++j) t += (f[j][0] + f[j][1]) + (f[j][2] + f[j][3]);
  const float inv_cnt = 1.f / ((float)g.Tn * (float)cpg);
  const float mean = block_sum(t, 0) * inv_cnt;
  t = 0.f;
#pragma unroll
  for (int j = 0; j < MAXI; ++j)
    if (tid + 256 * j < items) {
      const float d0 = f[j][0] - mean, d1 = f[j][1] - mean, d2 = f[j][2] - mean, d3 = f[j][3] - mean;
      t += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
  const float var = block_sum(t, 1) * inv_cnt;
  const float rstd = (float)(1.0 / sqrt((double)var + (double)eps));
  if (tid < cpg) {
    const int c = gi * cpg + tid;
    float av = rstd * gamma[c];
    float bv = beta[c] - mean * av;
    if (film) {
      const float sc = 1.f + film[(int64_t)s * film_ld + c];
      av *= sc;
      bv = bv * sc + film[(int64_t)s * film_ld + C + c];
    }
    if (a_out) {
      a_out[(int64_t)s * C + c] = av;
      b_out[(int64_t)s * C + c] = bv;
    }
    s_ab[0][tid] = av;
    s_ab[1][tid] = bv;
    if (mr_out && tid == 0) {
      mr_out[((int64_t)s * GN_GROUPS + gi) * 2] = mean;
      mr_out[((int64_t)s * GN_GROUPS + gi) * 2 + 1] = rstd;
    }
  }
  if (!y) return;                                            // block-uniform
  __syncthreads();
  char* yg = y + (int64_t)gi * cpg * ES;
#pragma unroll
  for (int j = 0; j < MAXI; ++j) {
    const int i = tid + 256 * j;
    if (i < items) {
      const int r = i / qpg, k = i - r * qpg;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float w = f[j][e] * s_ab[0][4 * k + e] + s_ab[1][4 * k + e];
        o[e] = act ? silu_f(w) : w;
      }
      char* q = yg + ((base + (int64_t)r * g.tstride) * ldy + 4 * k) * ES;
      if constexpr (ES == 2) {
        bf16x4 pk = {(__bf16)o[0], (__bf16)o[1], (__bf16)o[2], (__bf16)o[3]};
        *(u32x2*)q = __builtin_bit_cast(u32x2, pk);
      } else {
        *(f32x4*)q = f32x4{o[0], o[1], o[2], o[3]};
      }
    }
  }
}

extern "C" int mmd_gn_group(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int C, int S, int Tn, int inner, int64_t outer_stride,
                            int64_t inner_stride, int64_t tstride, const float* gamma, const float* beta, const float* film, int64_t film_ld,
                            float eps, int act, float* a_out, float* b_out, float* mr_out, void* stream) {
  int rc = check_geom("gn_group", dtype, C, S, Tn, inner);
  if (rc) return rc;
  MMD_REQUIRE(x && gamma && beta && (y || a_out), "gn_group: null pointer (needs y or a_out / b_out)");
  MMD_REQUIRE((a_out == nullptr) == (b_out == nullptr), "gn_group: a_out and b_out come together");
  MMD_REQUIRE(C % 128 == 0 && C <= 2048, "gn_group: groups of whole channel quads, at most 2048 channels (C = %d)", C);
  MMD_REQUIRE((int64_t)Tn * (C / 128) <= 4096, "gn_group: a (group, slice) of %d rows x %d channels does not fit one block's registers", Tn, C / 32);
  const int eb = dtype == MMD_BF16 ? 2 : 4;
  MMD_REQUIRE(((uintptr_t)x) % 16 == 0 && ldx % 4 == 0 && (!y || (((uintptr_t)y) % 16 == 0 && ldy % 4 == 0)), "gn_group: rows must be aligned to channel quads");
  (void)eb;
  SliceGeom g{S, Tn, inner, outer_stride, inner_stride, tstride};
  if (dtype == MMD_BF16)
    hipLaunchKernelGGL(gn_group_kernel<__bf16>, dim3(GN_GROUPS, S), dim3(256), 0, (hipStream_t)stream, (const char*)x, ldx, (char*)y, ldy, C, g, gamma,
                       beta, film, film_ld, eps, act, a_out, b_out, mr_out);
  else
    hipLaunchKernelGGL(gn_group_kernel<float>, dim3(GN_GROUPS, S), dim3(256), 0, (hipStream_t)stream, (const char*)x, ldx, (char*)y, ldy, C, g, gamma,
                       beta, film, film_ld, eps, act, a_out, b_out, mr_out);
  return mmd_check_launch("gn_group");
}

extern "C" int mmd_gn_apply(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int64_t rows, int C, int S, int Tn,
                            int inner, int64_t outer_stride, int64_t inner_stride, int64_t tstride, const float* a,
                            const float* b, int act, void* stream) {
  int rc = check_geom("gn_apply", dtype, C, S, Tn, inner);
  if (rc) return rc;
  MMD_REQUIRE(x && y && a && b && rows > 0, "gn_apply: null pointer / empty");
  SliceGeom g{S, Tn, inner, outer_stride, inner_stride, tstride};
  MMD_REQUIRE(rows == (int64_t)S * Tn, "gn_apply: rows %ld != S*Tn (%d x %d): the slices must tile the rows", (long)rows, S, Tn);
  const int rpp = max(1, 256 / (C / (dtype == MMD_BF16 ? 8 : 4)));
  int R = 4 * rpp;
  while ((int64_t)S * cdiv(Tn, R) > 4096 && R < 4096) R *= 2;
  dim3 grid(cdiv(Tn, R), S);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MMD_BF16)
    hipLaunchKernelGGL(gn_apply_kernel<__bf16>, grid, dim3(256), 0, st, (const char*)x, ldx, (char*)y, ldy, C, g, a, b, act, R);
  else
    hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, st, (const char*)x, ldx, (char*)y, ldy, C, g, a, b, act, R);
  return mmd_check_launch("gn_apply");
}

extern "C" int mmd_add_rowbias(int dtype, void* x, int64_t ld, int64_t rows, int C, int64_t rows_per_sample, const float* e,
                               int64_t e_ld, void* stream) {
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "add_rowbias: bad dtype");
  MMD_REQUIRE(x && e && rows > 0 && C % 8 == 0, "add_rowbias: bad argument");
  const int64_t total = rows * (C / (dtype == MMD_BF16 ? 8 : 4));
  const int grid = (int)min((int64_t)4096, (total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MMD_BF16)
    hipLaunchKernelGGL(add_rowbias_kernel<__bf16>, dim3(grid), dim3(256), 0, st, (char*)x, ld, rows, C, rows_per_sample, e, e_ld);
  else
    hipLaunchKernelGGL(add_rowbias_kernel<float>, dim3(grid), dim3(256), 0, st, (char*)x, ld, rows, C, rows_per_sample, e, e_ld);
  return mmd_check_launch("add_rowbias");
}

// GroupNorm32(+SiLU) of S short slices (Tn <= 16 rows each; geometry as mmd_gn_stats) in ONE launch: y = act(GN(x) gamma + beta).
extern "C" int mmd_gn_small(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int C, int S, int Tn, int inner,
                            int64_t outer_stride, int64_t inner_stride, int64_t tstride, const float* gamma, const float* beta,
                            float eps, int act, void* stream) {
  int rc = check_geom("gn_small", dtype, C, S, Tn, inner);
  if (rc) return rc;
  MMD_REQUIRE(C / (dtype == MMD_BF16 ? 8 : 4) <= 256, "gn_small: rows of at most 256 16-byte vectors (C = %d)", C);
  MMD_REQUIRE(x && y && gamma && beta, "gn_small: null pointer");
  MMD_REQUIRE(Tn <= 16 && (C / GN_GROUPS) % 4 == 0, "gn_small: slices of at most 16 rows, groups of whole channel quads (got Tn=%d C=%d)", Tn, C);
  SliceGeom g{S, Tn, inner, outer_stride, inner_stride, tstride};
  const int cv = C / (dtype == MMD_BF16 ? 8 : 4);
  MMD_REQUIRE(cv <= 256, "gn_small: %d channels is too wide", C);
  const int spb = 256 / cv;
  const size_t lds = (size_t)spb * (C / 4) * sizeof(float);
  if (dtype == MMD_BF16)
    hipLaunchKernelGGL((gn_small_kernel<__bf16>), dim3(cdiv(S, spb)), dim3(256), lds, (hipStream_t)stream, (const char*)x, ldx, (char*)y, ldy, C, g,
                       gamma, beta, eps, act, spb);
  else
    hipLaunchKernelGGL((gn_small_kernel<float>), dim3(cdiv(S, spb)), dim3(256), lds, (hipStream_t)stream, (const char*)x, ldx, (char*)y, ldy, C, g,
                       gamma, beta, eps, act, spb);
  return mmd_check_launch("gn_small");
}

