// Small / bandwidth-bound kernels of the denoising step:
//   timestep embedding + MLP (nn.py:192-210, multimodal_unet.py:791-795,1075), batched emb_layers Linear
//   (unet:366-372), avg-pool / nearest resampling (unet:133-208), the stem (InitialBlock, unet:680-694) and
//   head (unet:1003-1012) convolutions at the API layout edge, skip-concat copies (unet:1093-1094) and the
//   fused DDPM ancestral update (multimodal_gaussian_diffusion.py:231-343 get_variance + 453-470 p_sample).
#include "mmd_common.h"

// ----------------------------------------------------------------------------- timestep embedding + MLP
// one block per sample; out_silu = SiLU(W2 SiLU(W0 e + b0) + b2), out_raw (optional) = without the last SiLU
__global__ __launch_bounds__(256) void temb_kernel(const void* __restrict__ t, int t_kind, int dim, const float* __restrict__ W0,
                                                   const float* __restrict__ b0, const float* __restrict__ W2,
                                                   const float* __restrict__ b2, float* __restrict__ out_silu,
                                                   float* __restrict__ out_raw) {
  extern __shared__ float sm[];
  float* e = sm;          // [dim]
  float* h = sm + dim;    // [dim]
  const int n = blockIdx.x, tid = threadIdx.x;
  float tv;
  if (t_kind == 0) tv = (float)((const int64_t*)t)[n];
  else if (t_kind == 1) tv = (float)((const int32_t*)t)[n];
  else tv = ((const float*)t)[n];
  const int half = dim / 2;
  for (int i = tid; i < dim; i += 256) {
    float v = 0.f;
    if (i < 2 * half) {
      const int k = i < half ? i : i - half;
      const float f = expf(-logf(10000.f) * (float)k / (float)half);
      const float a = tv * f;
      v = i < half ? cosf(a) : sinf(a);
    }
    e[i] = v;
  }
  __syncthreads();
  for (int j = tid; j < dim; j += 256) {
    float a = b0[j];
    for (int k = 0; k < dim; ++k) a += W0[j * dim + k] * e[k];
    h[j] = silu_f(a);
  }
  __syncthreads();
  for (int j = tid; j < dim; j += 256) {
    float a = b2[j];
    for (int k = 0; k < dim; ++k) a += W2[j * dim + k] * h[k];
    if (out_raw) out_raw[n * dim + j] = a;
    out_silu[n * dim + j] = silu_f(a);
  }
}

// y[n, j] = b[j] + sum_k x[n,k] W[j,k] ; one wave per output column j, all n (N <= 16)
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ b, float* __restrict__ y, int N, int K, int J) {
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= J) return;
  for (int n0 = 0; n0 < N; n0 += 8) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int k = lane; k < K; k += 64) {
      const float w = W[(int64_t)j * K + k];
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (n0 + i < N) acc[i] += w * x[(int64_t)(n0 + i) * K + k];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float s = wave_sum(acc[i]);
      if (lane == 0 && n0 + i < N) y[(int64_t)(n0 + i) * J + j] = s + (b ? b[j] : 0.f);
    }
  }
}

// ----------------------------------------------------------------------------- resampling (channels-last rows)
// mode 0: average pool, mode 1: nearest upsample.  Rows are (n, f, h, w) with factors (1, fh, fw) - audio uses
// F=1, H=1, W=L, fw=4.  Geometry given for the INPUT; output dims = in / factor (pool) or in * factor (up).
template <typename T>
__global__ __launch_bounds__(256) void resample_kernel(const char* __restrict__ x, int64_t ldx, char* __restrict__ y, int64_t ldy,
                                                       int C, int NF, int H, int W, int fh, int fw, int mode, float scale) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  const int CV = C / EPV;
  const int Ho = mode == 0 ? H / fh : H * fh, Wo = mode == 0 ? W / fw : W * fw;
  const int64_t total = (int64_t)NF * Ho * Wo * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const int64_t orow = i / CV;
    const int wo = (int)(orow % Wo), ho = (int)((orow / Wo) % Ho);
    const int64_t nf = orow / ((int64_t)Wo * Ho);
    float acc[EPV];
    if (mode == 0) {
#pragma unroll
      for (int e = 0; e < EPV; ++e) acc[e] = 0.f;
      for (int a = 0; a < fh; ++a)
        for (int b = 0; b < fw; ++b) {
          const int64_t irow = (nf * H + ho * fh + a) * W + wo * fw + b;
          float f[EPV];
          Elt<T>::unpack(*(const u32x4*)(x + (irow * ldx + (int64_t)cv * EPV) * ES), f);
#pragma unroll
          for (int e = 0; e < EPV; ++e) acc[e] += f[e];
        }
      const float inv = scale / (float)(fh * fw);
#pragma unroll
      for (int e = 0; e < EPV; ++e) acc[e] *= inv;
      *(u32x4*)(y + (orow * ldy + (int64_t)cv * EPV) * ES) = Elt<T>::pack(acc);
    } else {
      const int64_t irow = (nf * H + ho / fh) * W + wo / fw;
      u32x4 v = *(const u32x4*)(x + (irow * ldx + (int64_t)cv * EPV) * ES);
      if (scale != 1.f) {
        Elt<T>::unpack(v, acc);
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[e] *= scale;
        v = Elt<T>::pack(acc);
      }
      *(u32x4*)(y + (orow * ldy + (int64_t)cv * EPV) * ES) = v;
    }
  }
}

// The same resampling (bf16) with the GroupNorm statistics of the OUTPUT in the epilogue (round 5): one block owns one 64-row RECORD of the
// output, writes its rows and leaves (sum, sum of squares) of the values AS STORED per quad of channels - the record format of the GEMM
// epilogues (mmd_conv_gemm_stats) - so a norm that follows a resample (the out_layers norm of the down ResBlocks, every norm that reads
// an up ResBlock's output, unet:441-448) finalizes from records instead of paying a statistics pass over the tensor.
// Thread (rg, cv) owns channel vector cv (8 channels = two quads) of the rows rg, rg + RGN, ...: per-thread sums in row order, then the
// row groups are folded in ascending order by the rg == 0 threads - a fixed order, bitwise repeatable and independent of the grid.
__global__ __launch_bounds__(256) void resample_stats_kernel(const char* __restrict__ x, int64_t ldx, char* __restrict__ y, int64_t ldy, int C,
                                                             int NF, int H, int W, int fh, int fw, int mode, float* __restrict__ stats,
                                                             int64_t stats_ld, int nrec, int lcvp) {
  __shared__ float sPart[256 * 4];
  const int tid = threadIdx.x, CV = C >> 3;
  const int cv = tid & ((1 << lcvp) - 1), rg = tid >> lcvp, rgn = 256 >> lcvp;
  const int Ho = mode == 0 ? H / fh : H * fh, Wo = mode == 0 ? W / fw : W * fw;
  const float inv = 1.f / (float)(fh * fw);
  for (int rec = blockIdx.x; rec < nrec; rec += gridDim.x) {
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    if (cv < CV) {
#pragma unroll 4
      for (int r = rg; r < 64; r += rgn) {
        const int orow = rec * 64 + r;                     // < 2^31 (checked by the launcher)
        const int wo = orow % Wo, t = orow / Wo, ho = t % Ho, nf = t / Ho;
        u32x4 v;
        if (mode == 0) {
          float acc[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] = 0.f;
          for (int a = 0; a < fh; ++a)
            for (int b = 0; b < fw; ++b) {
              const int64_t irow = ((int64_t)nf * H + ho * fh + a) * W + wo * fw + b;
              float f[8];
              Elt<__bf16>::unpack(*(const u32x4*)(x + (irow * ldx + (int64_t)cv * 8) * 2), f);
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[e] += f[e];
            }
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] *= inv;
          v = Elt<__bf16>::pack(acc);
        } else {
          const int64_t irow = ((int64_t)nf * H + ho / fh) * W + wo / fw;
          v = *(const u32x4*)(x + (irow * ldx + (int64_t)cv * 8) * 2);
        }
        *(u32x4*)(y + ((int64_t)orow * ldy + (int64_t)cv * 8) * 2) = v;
        float f[8];
        Elt<__bf16>::unpack(v, f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s0 += f[j];
          s1 += f[4 + j];
          q0 += f[j] * f[j];
          q1 += f[4 + j] * f[4 + j];
        }
      }
    }
    *(f32x4*)(sPart + tid * 4) = f32x4{s0, s1, q0, q1};
    __syncthreads();
    if (rg == 0 && cv < CV) {
      for (int g = 1; g < rgn; ++g) {
        const f32x4 o = *(const f32x4*)(sPart + ((g << lcvp) + cv) * 4);
        s0 += o[0];
        s1 += o[1];
        q0 += o[2];
        q1 += o[3];
      }
      *(f32x4*)(stats + ((int64_t)rec * stats_ld + 2 * cv) * 2) = f32x4{s0, q0, s1, q1};     // quads 2 cv, 2 cv + 1: (sum, sum of squares)
    }
    __syncthreads();
  }
}

// strided 2-D copy of 16-byte vecs (skip-connection concat: write a tensor into a column slice)
__global__ __launch_bounds__(256) void copy2d_kernel(const char* __restrict__ x, int64_t ldx_b, char* __restrict__ y, int64_t ldy_b,
                                                     int64_t rows, int vecs) {
  const int64_t total = rows * vecs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / vecs;
    const int v = (int)(i % vecs);
    *(u32x4*)(y + r * ldy_b + (int64_t)v * 16) = *(const u32x4*)(x + r * ldx_b + (int64_t)v * 16);
  }
}

// ----------------------------------------------------------------------------- stem conv (API layout -> channels-last)
// in : fp32 [N, F, Cin, H, W] (audio: F=1, H=1, W=L)   W packed fp32 [ntaps][Cin][Cout]   out: T [N*F*H*W, Cout]
struct EdgeConvParams {
  const float* x; const float* w; const float* bias;
  char* y; int64_t ldy;
  int N, F, Cin, H, W, Cout, ntaps;
  int taps[27 * 3];
};
template <typename T>
__global__ __launch_bounds__(256) void stem_conv_kernel(const EdgeConvParams p) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  extern __shared__ float sw[];    // [ntaps*Cin][Cout]
  const int KW = p.ntaps * p.Cin;
  for (int i = threadIdx.x; i < KW * p.Cout; i += 256) sw[i] = p.w[i];
  __syncthreads();
  const int CV = p.Cout / EPV;
  const int HW = p.H * p.W;
  const int64_t rows = (int64_t)p.N * p.F * HW;
  const int64_t total = rows * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    const int64_t m = i / CV;
    const int w0 = (int)(m % p.W), h0 = (int)((m / p.W) % p.H), f0 = (int)((m / HW) % p.F);
    const int64_t n = m / ((int64_t)HW * p.F);
    float acc[EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) acc[e] = p.bias ? p.bias[cv * EPV + e] : 0.f;
    for (int t = 0; t < p.ntaps; ++t) {
      const int f = f0 + p.taps[t * 3], h = h0 + p.taps[t * 3 + 1], w = w0 + p.taps[t * 3 + 2];
      if ((unsigned)f >= (unsigned)p.F || (unsigned)h >= (unsigned)p.H || (unsigned)w >= (unsigned)p.W) continue;
      for (int ci = 0; ci < p.Cin; ++ci) {
        const float xv = p.x[(((n * p.F + f) * p.Cin + ci) * p.H + h) * p.W + w];
        const float* wr = sw + (t * p.Cin + ci) * p.Cout + cv * EPV;
#pragma unroll
        for (int e = 0; e < EPV; ++e) acc[e] += xv * wr[e];
      }
    }
    *(u32x4*)(p.y + (m * p.ldy + (int64_t)cv * EPV) * ES) = Elt<T>::pack(acc);
  }
}

// Strip variant (W % 4 == 0, Cin in {1, 3}): thread = (one 16-byte chunk of output channels, FOUR consecutive pixels along w).
// The per-pixel kernel above walks its taps one dependent scalar load at a time (27 L2 round trips per thread: 210 us for the
// 16x64x64 stem, 8.6 TFLOP/s) and re-reads every weight from LDS per pixel; here three taps x Cin x 4 pixels of loads are in
// flight before the first FMA, borders are handled branch-free (clamped address, zeroed value), and each weight read from LDS
// feeds four pixels.
template <typename T, int CIN>
__global__ __launch_bounds__(256) void stem_conv_strip_kernel(const EdgeConvParams p) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  constexpr int PX = 4;
  extern __shared__ float sw[];    // [ntaps*CIN][Cout]
  for (int i = threadIdx.x; i < p.ntaps * CIN * p.Cout; i += 256) sw[i] = p.w[i];
  __syncthreads();
  const int CV = p.Cout / EPV;
  const int WS = p.W / PX;
  const int64_t total = (int64_t)p.N * p.F * p.H * WS * CV;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    int64_t sidx = i / CV;
    const int w0 = (int)(sidx % WS) * PX;
    sidx /= WS;
    const int h0 = (int)(sidx % p.H);
    sidx /= p.H;
    const int f0 = (int)(sidx % p.F);
    const int64_t n = sidx / p.F;
    float acc[PX][EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const float b = p.bias ? p.bias[cv * EPV + e] : 0.f;
#pragma unroll
      for (int px = 0; px < PX; ++px) acc[px][e] = b;
    }
    for (int tg = 0; tg < p.ntaps; tg += 3) {
      float xv[3][CIN][PX];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int t = min(tg + u, p.ntaps - 1);
        const int f = f0 + p.taps[t * 3], h = h0 + p.taps[t * 3 + 1], wb = w0 + p.taps[t * 3 + 2];
        const bool okfh = (tg + u < p.ntaps) && (unsigned)f < (unsigned)p.F && (unsigned)h < (unsigned)p.H;
        const int fc = okfh ? f : f0, hc = okfh ? h : h0;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const float* xr = p.x + (((n * p.F + fc) * CIN + ci) * p.H + hc) * (int64_t)p.W;
#pragma unroll
          for (int px = 0; px < PX; ++px) {
            const int w = wb + px;
            const float v = xr[min(max(w, 0), p.W - 1)];
            xv[u][ci][px] = (okfh && (unsigned)w < (unsigned)p.W) ? v : 0.f;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (tg + u < p.ntaps) {
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) {
            const float* wr = sw + ((tg + u) * CIN + ci) * p.Cout + cv * EPV;
            float wv[EPV];
#pragma unroll
            for (int e = 0; e < EPV; e += 4) {
              const f32x4 w4 = *(const f32x4*)(wr + e);
#pragma unroll
              for (int k = 0; k < 4; ++k) wv[e + k] = w4[k];
            }
#pragma unroll
            for (int px = 0; px < PX; ++px)
#pragma unroll
              for (int e = 0; e < EPV; ++e) acc[px][e] += xv[u][ci][px] * wv[e];
          }
        }
      }
    }
    const int64_t m0 = ((n * p.F + f0) * p.H + h0) * (int64_t)p.W + w0;
#pragma unroll
    for (int px = 0; px < PX; ++px) *(u32x4*)(p.y + ((m0 + px) * p.ldy + (int64_t)cv * EPV) * ES) = Elt<T>::pack(acc[px]);
  }
}

// ----------------------------------------------------------------------------- head conv (channels-last -> API layout)
// in: T rows [N*F*H*W, Cin] (already GN+SiLU'd)   W packed fp32 [ntaps][Cin][Co] (Co <= 8)   out fp32 [N,F,Co,H,W]
struct HeadConvParams {
  const char* x; int64_t ldx; const float* w; const float* bias;
  float* y;
  int N, F, Cin, H, W, Co, ntaps;
  int taps[27 * 3];
};
template <typename T, int CO>
__global__ __launch_bounds__(256) void head_conv_kernel(const HeadConvParams p) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  extern __shared__ float sw[];    // [ntaps*Cin][CO]
  const int KW = p.ntaps * p.Cin;
  for (int i = threadIdx.x; i < KW * CO; i += 256) {
    const int co = i % CO;
    sw[i] = co < p.Co ? p.w[(i / CO) * p.Co + co] : 0.f;
  }
  __syncthreads();
  const int HW = p.H * p.W;
  const int64_t rows = (int64_t)p.N * p.F * HW;
  const int CinV = p.Cin / EPV;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < rows; m += (int64_t)gridDim.x * 256) {
    const int w0 = (int)(m % p.W), h0 = (int)((m / p.W) % p.H), f0 = (int)((m / HW) % p.F);
    const int64_t n = m / ((int64_t)HW * p.F);
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = (p.bias && c < p.Co) ? p.bias[c] : 0.f;
    for (int t = 0; t < p.ntaps; ++t) {
      const int df = p.taps[t * 3], dh = p.taps[t * 3 + 1], dw = p.taps[t * 3 + 2];
      if ((unsigned)(f0 + df) >= (unsigned)p.F || (unsigned)(h0 + dh) >= (unsigned)p.H || (unsigned)(w0 + dw) >= (unsigned)p.W) continue;
      const int64_t src = m + (int64_t)df * HW + dh * p.W + dw;
      const char* xr = p.x + src * p.ldx * ES;
      const float* wt = sw + (int64_t)t * p.Cin * CO;
      for (int v = 0; v < CinV; ++v) {
        float f[EPV];
        Elt<T>::unpack(*(const u32x4*)(xr + v * 16), f);
#pragma unroll
        for (int e = 0; e < EPV; ++e)
#pragma unroll
          for (int c = 0; c < CO; ++c) acc[c] += f[e] * wt[(v * EPV + e) * CO + c];
      }
    }
    const int hw = h0 * p.W + w0;
    for (int c = 0; c < p.Co; ++c) p.y[((n * p.F + f0) * p.Co + c) * HW + hw] = acc[c];
  }
}

// ----------------------------------------------------------------------------- fused DDPM ancestral update
// tables: fp32 [7][T] rows = sqrt_recip_ac, sqrt_recipm1_ac, post_c1, post_c2, logvar_fixed, min_log, max_log
// x, noise, out: fp32 [N, F, C, HW] ; model_out fp32 [N, F, Cm, HW] with Cm = C (fixed var) or 2C (learned range)
// flags bit0: clip x0 to [-1,1], bit1: model predicts x0, bit2: learned-range variance
struct DdpmParams {
  const float* x; const float* mo; const float* noise; float* out; float* x0_out; float* mean_out; float* logvar_out;
  const float* tables; const int64_t* t;
  int T, N, F, C, HW, flags;
};
__global__ __launch_bounds__(256) void ddpm_update_kernel(const DdpmParams p) {
  const int64_t per = (int64_t)p.F * p.C * p.HW;
  const int64_t total = per * p.N;
  const int Cm = (p.flags & 4) ? 2 * p.C : p.C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / per, r = i % per;
    const int hw = (int)(r % p.HW), c = (int)((r / p.HW) % p.C);
    const int64_t f = r / ((int64_t)p.HW * p.C);
    const int ti = (int)p.t[n];
    const float cr = p.tables[ti], crm1 = p.tables[p.T + ti], c1 = p.tables[2 * p.T + ti], c2 = p.tables[3 * p.T + ti];
    const int64_t mbase = ((n * p.F + f) * Cm) * (int64_t)p.HW + hw;
    const float o = p.mo[mbase + (int64_t)c * p.HW];
    float logvar;
    if (p.flags & 4) {
      const float vv = p.mo[mbase + (int64_t)(c + p.C) * p.HW];
      const float frac = (vv + 1.f) / 2.f;
      logvar = frac * p.tables[6 * p.T + ti] + (1.f - frac) * p.tables[5 * p.T + ti];
    } else {
      logvar = p.tables[4 * p.T + ti];
    }
    const float xv = p.x[i];
    float x0 = (p.flags & 2) ? o : cr * xv - crm1 * o;
    if (p.flags & 1) x0 = fminf(fmaxf(x0, -1.f), 1.f);
    const float mean = c1 * x0 + c2 * xv;
    const float nz = ti != 0 ? 1.f : 0.f;
    if (p.out) p.out[i] = mean + nz * expf(0.5f * logvar) * p.noise[i];
    if (p.x0_out) p.x0_out[i] = x0;
    if (p.mean_out) p.mean_out[i] = mean;
    if (p.logvar_out) p.logvar_out[i] = logvar;
  }
}

// x_t = sqrt_ac[t] x0 + sqrt_1mac[t] eps   (q_sample, multimodal_gaussian_diffusion.py:187-205); tab2 = [2][T]
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ eps, float* __restrict__ out,
                                                       const float* __restrict__ tab2, const int64_t* __restrict__ t, int T, int64_t per,
                                                       int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ti = (int)t[i / per];
    out[i] = tab2[ti] * x0[i] + tab2[T + ti] * eps[i];
  }
}


// ----------------------------------------------------------------------------- training-loss reductions (forward values)
// Per-sample terms of multimodal_training_losses (gd:1114-1203) for one stream, API layout [N, F, Cm, HW]:
//   mse[n]  = mean((target - eps_hat)^2)                      (target = noise, or x0 when the model predicts x0)
//   vb[n]   = mean(KL(q(x_{t-1}|x_t,x_0) || p) ) / ln2  for t > 0, decoder NLL / ln2 at t == 0   (learned-range variance;
//             _vb_terms_bpd gd:1048-1092 with the frozen mean, normal_kl / discretized_gaussian_log_likelihood losses.py:12-77)
// One block per (sample, chunk); fixed-order tree reduction -> deterministic.  partial [N, nchunk, 2] doubles.
struct LossParams {
  const float* x0; const float* xt; const float* mo; const float* target;
  const float* tables; const int64_t* t;
  double* partial;
  int T, N, F, C, HW, flags, nchunk;
};
__device__ __forceinline__ float approx_std_normal_cdf(float x) {
  return 0.5f * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__global__ __launch_bounds__(256) void loss_terms_kernel(const LossParams p) {
  __shared__ double s_a[256], s_b[256];
  const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int64_t per = (int64_t)p.F * p.C * p.HW;
  const int Cm = (p.flags & 4) ? 2 * p.C : p.C;
  const int ti = (int)p.t[n];
  const float cr = p.tables[ti], crm1 = p.tables[p.T + ti], c1 = p.tables[2 * p.T + ti], c2 = p.tables[3 * p.T + ti];
  const float min_log = p.tables[5 * p.T + ti], max_log = p.tables[6 * p.T + ti];
  double mse = 0.0, vb = 0.0;
  const int64_t lo = per * chunk / p.nchunk, hi = per * (chunk + 1) / p.nchunk;
  for (int64_t r = lo + tid; r < hi; r += 256) {
    const int hw = (int)(r % p.HW), c = (int)((r / p.HW) % p.C);
    const int64_t f = r / ((int64_t)p.HW * p.C);
    const int64_t i = n * per + r;
    const int64_t mbase = ((n * (int64_t)p.F + f) * Cm) * (int64_t)p.HW + hw;
    const float o = p.mo[mbase + (int64_t)c * p.HW];
    const float d = p.target[i] - o;
    mse += (double)(d * d);
    if (p.flags & 4) {
      const float vv = p.mo[mbase + (int64_t)(c + p.C) * p.HW];
      const float frac = (vv + 1.f) / 2.f;
      const float logvar = frac * max_log + (1.f - frac) * min_log;
      const float xv = p.xt[i], x0 = p.x0[i];
      const float px0 = (p.flags & 2) ? o : cr * xv - crm1 * o;          // clip_denoised=False in the vb term
      const float mean = c1 * px0 + c2 * xv;
      const float tmean = c1 * x0 + c2 * xv;
      float term;
      if (ti == 0) {        // decoder NLL
        const float cx = x0 - mean, inv = expf(-0.5f * logvar);
        const float cdf_p = approx_std_normal_cdf(inv * (cx + 1.f / 255.f));
        const float cdf_m = approx_std_normal_cdf(inv * (cx - 1.f / 255.f));
        const float lp = logf(fmaxf(cdf_p, 1e-12f)), lm = logf(fmaxf(1.f - cdf_m, 1e-12f));
        const float ld = logf(fmaxf(cdf_p - cdf_m, 1e-12f));
        term = -(x0 < -0.999f ? lp : (x0 > 0.999f ? lm : ld));
      } else {              // KL(q || p), true posterior log-variance = min_log (posterior_log_variance_clipped)
        const float dm = tmean - mean;
        term = 0.5f * (-1.0f + logvar - min_log + expf(min_log - logvar) + dm * dm * expf(-logvar));
      }
      vb += (double)term;
    }
  }
  s_a[tid] = mse;
  s_b[tid] = vb;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { s_a[tid] += s_a[tid + o]; s_b[tid] += s_b[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    p.partial[((int64_t)n * p.nchunk + chunk) * 2] = s_a[0];
    p.partial[((int64_t)n * p.nchunk + chunk) * 2 + 1] = s_b[0];
  }
}
__global__ void loss_finalize_kernel(const double* __restrict__ partial, int nchunk, double inv_count, float vb_scale,
                                     float* __restrict__ mse_out, float* __restrict__ vb_out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (int)gridDim.x * (int)blockDim.x) return;
  double a = 0.0, b = 0.0;
  for (int k = 0; k < nchunk; ++k) { a += partial[((int64_t)n * nchunk + k) * 2]; b += partial[((int64_t)n * nchunk + k) * 2 + 1]; }
  mse_out[n] = (float)(a * inv_count);
  if (vb_out) vb_out[n] = (float)(b * inv_count / 0.6931471805599453) * vb_scale;
}

// ============================================================================= C-ABI
static inline int ew_grid(int64_t total) { return (int)min((int64_t)4096, (total + 255) / 256); }

extern "C" int mmd_temb_fwd(const void* t, int t_kind, int N, int dim, const float* W0, const float* b0, const float* W2,
                            const float* b2, float* out_silu, float* out_raw, void* stream) {
  MMD_REQUIRE(t && W0 && b0 && W2 && b2 && out_silu && N > 0 && dim > 0 && dim <= 4096, "temb_fwd: bad argument");
  MMD_REQUIRE(t_kind >= 0 && t_kind <= 2, "temb_fwd: t_kind must be 0 (int64), 1 (int32) or 2 (float32)");
  hipLaunchKernelGGL(temb_kernel, dim3(N), dim3(256), 2 * dim * sizeof(float), (hipStream_t)stream, t, t_kind, dim, W0, b0, W2,
                     b2, out_silu, out_raw);
  return mmd_check_launch("temb");
}

extern "C" int mmd_linear_fwd(const float* x, const float* W, const float* b, float* y, int N, int K, int J, void* stream) {
  MMD_REQUIRE(x && W && y && N > 0 && K > 0 && J > 0, "linear_fwd: bad argument");
  hipLaunchKernelGGL(linear_kernel, dim3(cdiv(J, 4)), dim3(256), 0, (hipStream_t)stream, x, W, b, y, N, K, J);
  return mmd_check_launch("linear");
}

extern "C" int mmd_resample(int dtype, const void* x, int64_t ldx, void* y, int64_t ldy, int C, int NF, int H, int W, int fh,
                            int fw, int mode, float scale, void* stream) {
  const int epv = dtype == MMD_BF16 ? 8 : 4;
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "resample: bad dtype");
  MMD_REQUIRE(x && y && C % epv == 0 && NF > 0 && H > 0 && W > 0 && fh > 0 && fw > 0, "resample: bad argument");
  MMD_REQUIRE(mode == 1 || (H % fh == 0 && W % fw == 0), "resample: pooled dims must divide (%d/%d, %d/%d)", H, fh, W, fw);
  const int64_t orows = mode == 0 ? (int64_t)NF * (H / fh) * (W / fw) : (int64_t)NF * H * fh * W * fw;
  const int grid = ew_grid(orows * (C / epv));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MMD_BF16)
    hipLaunchKernelGGL(resample_kernel<__bf16>, dim3(grid), dim3(256), 0, st, (const char*)x, ldx, (char*)y, ldy, C, NF, H, W, fh, fw, mode, scale);
  else
    hipLaunchKernelGGL(resample_kernel<float>, dim3(grid), dim3(256), 0, st, (const char*)x, ldx, (char*)y, ldy, C, NF, H, W, fh, fw, mode, scale);
  return mmd_check_launch("resample");
}

extern "C" int mmd_resample_stats(const void* x, int64_t ldx, void* y, int64_t ldy, int C, int NF, int H, int W, int fh, int fw, int mode,
                                  float* stats, int64_t stats_ld, void* stream) {
  MMD_REQUIRE(x && y && stats && C > 0 && C % 8 == 0 && C <= 2048 && NF > 0 && H > 0 && W > 0 && fh > 0 && fw > 0 && (mode == 0 || mode == 1),
              "resample_stats: bad argument (bf16 rows of 8 .. 2048 channels in whole 16-byte vectors)");
  MMD_REQUIRE(mode == 1 || (H % fh == 0 && W % fw == 0), "resample_stats: pooled dims must divide (%d/%d, %d/%d)", H, fh, W, fw);
  const int64_t orows = mode == 0 ? (int64_t)NF * (H / fh) * (W / fw) : (int64_t)NF * H * fh * W * fw;
  MMD_REQUIRE(orows % 64 == 0 && orows < ((int64_t)1 << 31), "resample_stats: %ld output rows (records are 64 rows; < 2^31)", (long)orows);
  MMD_REQUIRE((uintptr_t)stats % 16 == 0 && stats_ld % 2 == 0 && stats_ld >= C / 4, "resample_stats: the record view must start on an even quad "
              "of a 16-byte aligned buffer (ld %ld quads)", (long)stats_ld);
  MMD_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "resample_stats: unaligned rows");
  int lcvp = 0;
  while ((1 << lcvp) < C / 8) ++lcvp;
  const int nrec = (int)(orows / 64);
  hipLaunchKernelGGL(resample_stats_kernel, dim3(min(nrec, 8192)), dim3(256), 0, (hipStream_t)stream, (const char*)x, ldx, (char*)y, ldy, C, NF, H,
                     W, fh, fw, mode, stats, stats_ld, nrec, lcvp);
  return mmd_check_launch("resample_stats");
}

extern "C" int mmd_copy2d(const void* x, int64_t ldx_bytes, void* y, int64_t ldy_bytes, int64_t rows, int64_t row_bytes, void* stream) {
  MMD_REQUIRE(x && y && rows > 0 && row_bytes > 0 && row_bytes % 16 == 0 && ldx_bytes % 16 == 0 && ldy_bytes % 16 == 0,
              "copy2d: rows must be 16-byte multiples");
  MMD_REQUIRE(((uintptr_t)x | (uintptr_t)y) % 16 == 0, "copy2d: unaligned pointer");
  const int vecs = (int)(row_bytes / 16);
  hipLaunchKernelGGL(copy2d_kernel, dim3(ew_grid(rows * vecs)), dim3(256), 0, (hipStream_t)stream, (const char*)x, ldx_bytes,
                     (char*)y, ldy_bytes, rows, vecs);
  return mmd_check_launch("copy2d");
}


// MFMA stem conv (bf16 rows out; ntaps * Cin <= 28, Cout = 32 NB <= 128, W % 32 == 0): the 27-term dot products of the video stem on
// the fp32 matrix pipe (v_mfma_f32_32x32x2f32: exact fp32 products, fp32 accumulation) instead of 0.9 G scalar FMAs - the strip kernel
// above runs at 17 TFLOP/s of VALU (107 us for the 16 x 64 x 64 stem against an 8 us output write).  D[cout][pixel] = W[cout][k] X[k][pixel]:
// a wave owns 32 consecutive pixels of one image row; lane (n = lane % 32, kk = lane / 32) gathers x for k = 2 s + kk straight from the
// API-layout input (coalesced along w; padding reads as zero), the weights of the lane's output channel sit in registers for the whole
// kernel, and the epilogue is the row-strip GEMM's: half-wave swap -> 8 consecutive channels per lane -> bias -> one 16-byte store.
template <int NB>
__global__ __launch_bounds__(256, 2) void stem_conv_mfma_kernel(const EdgeConvParams p) {
  constexpr int MAXS = 14;
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int K = p.ntaps * p.Cin, KS = (K + 1) >> 1;
  float wreg[MAXS][NB];
  int kd[MAXS];                                          // per step: this lane's (df, dh, dw, ci), or -1 past K
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const int k = 2 * s + half;
    const bool kv = s < KS && k < K;
    const int kc = kv ? k : 0, t = kc / p.Cin, ci = kc - t * p.Cin;
    kd[s] = kv ? ((p.taps[t * 3] + 1) | ((p.taps[t * 3 + 1] + 1) << 2) | ((p.taps[t * 3 + 2] + 1) << 4) | (ci << 6)) : -1;
#pragma unroll
    for (int b = 0; b < NB; ++b) wreg[s][b] = kv ? p.w[(int64_t)kc * p.Cout + b * 32 + l31] : 0.f;
  }
  const int HW = p.H * p.W, WG = p.W >> 5;
  const int64_t groups = (int64_t)p.N * p.F * p.H * WG;
  const int64_t nwave = (int64_t)gridDim.x * 4, wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  auto gather = [&](int64_t g, float (&xv)[MAXS]) {
    const int w = (int)(g % WG) * 32 + l31;
    int64_t r = g / WG;
    const int h = (int)(r % p.H);
    r /= p.H;
    const int f = (int)(r % p.F);
    const int64_t n = r / p.F;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      if (s < KS) {                                        // uniform: the audio stem (3 taps x 1 channel) has two steps, not fourteen
        const int d = kd[s];
        const int df = (d & 3) - 1, dh = ((d >> 2) & 3) - 1, dw = ((d >> 4) & 3) - 1, ci = (d >> 6) & 3;
        const bool ok = d >= 0 && (unsigned)(f + df) < (unsigned)p.F && (unsigned)(h + dh) < (unsigned)p.H && (unsigned)(w + dw) < (unsigned)p.W;
        const int64_t src = (((n * p.F + (f + df)) * p.Cin + ci) * p.H + (h + dh)) * (int64_t)p.W + (w + dw);
        xv[s] = ok ? p.x[src] : 0.f;
      }
    }
  };
  float xcur[MAXS] = {}, xnext[MAXS] = {};
  if (wave_id < groups) gather(wave_id, xcur);
  for (int64_t g = wave_id; g < groups; g += nwave) {
    if (g + nwave < groups) gather(g + nwave, xnext);    // the next group's gather flies under this group's MFMAs
    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      if (s < KS) {
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[s][b], xcur[s], acc[b], 0, 0, 0);
      }
    }
    // acc[b][4 q + j] = channel 32 b + 8 q + 4 half + j of pixel l31: pair q = 2 j2 with q = 2 j2 + 1 across the half-waves
    const int64_t m = g * 32 + l31;                      // groups walk the rows in order: 32 consecutive pixels of one image row
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[b][8 * j2 + j]), __float_as_uint(acc[b][8 * j2 + 4 + j]), false, false);
          v[j] = __uint_as_float(sw[0]);
          v[4 + j] = __uint_as_float(sw[1]);
        }
        const int col = b * 32 + 16 * j2 + 8 * half;
        if (p.bias) {
          const f32x4 b0 = *(const f32x4*)(p.bias + col), b1 = *(const f32x4*)(p.bias + col + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
        }
        *(u32x4*)(p.y + (m * p.ldy + col) * 2) = Elt<__bf16>::pack(v);
      }
#pragma unroll
    for (int s = 0; s < MAXS; ++s) xcur[s] = s < KS ? xnext[s] : 0.f;
  }
}

static bool stem_mfma_ok(int dtype, const EdgeConvParams& p) {
  static const bool on = [] { const char* e = getenv("MMD_STEM_MFMA"); return !(e && e[0] == '0'); }();
  for (int i = 0; i < p.ntaps * 3; ++i)
    if (p.taps[i] < -1 || p.taps[i] > 1) return false;          // the kernel packs a tap offset + 1 into two bits
  return on && dtype == MMD_BF16 && p.W % 32 == 0 && p.ntaps * p.Cin <= 28 && p.Cin <= 3 && p.Cout % 32 == 0 && p.Cout <= 128 && p.ldy % 8 == 0 &&
         ((uintptr_t)p.y) % 16 == 0 && (!p.bias || ((uintptr_t)p.bias) % 16 == 0);
}

static int launch_stem_mfma(const EdgeConvParams& p, hipStream_t st) {
  const int64_t groups = (int64_t)p.N * p.F * p.H * (p.W / 32);
  const int grid = (int)min((int64_t)2048, (groups + 3) / 4);
  switch (p.Cout / 32) {
    case 1: hipLaunchKernelGGL(stem_conv_mfma_kernel<1>, dim3(grid), dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL(stem_conv_mfma_kernel<2>, dim3(grid), dim3(256), 0, st, p); break;
    case 3: hipLaunchKernelGGL(stem_conv_mfma_kernel<3>, dim3(grid), dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL(stem_conv_mfma_kernel<4>, dim3(grid), dim3(256), 0, st, p); break;
  }
  return mmd_check_launch("stem_conv_mfma");
}

extern "C" int mmd_stem_conv(int dtype, const float* x, const float* w, const float* bias, void* y, int64_t ldy, int N, int F,
                             int Cin, int H, int W, int Cout, int ntaps, const int* taps, void* stream) {
  const int epv = dtype == MMD_BF16 ? 8 : 4;
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "stem_conv: bad dtype");
  MMD_REQUIRE(x && w && y && taps && ntaps >= 1 && ntaps <= 27 && Cout % epv == 0, "stem_conv: bad argument");
  const size_t lds = (size_t)ntaps * Cin * Cout * sizeof(float);
  MMD_REQUIRE(lds <= 64 * 1024, "stem_conv: weights (%zu B) exceed the 64 KiB LDS stage", lds);
  EdgeConvParams p;
  p.x = x; p.w = w; p.bias = bias; p.y = (char*)y; p.ldy = ldy;
  p.N = N; p.F = F; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.ntaps = ntaps;
  for (int i = 0; i < ntaps * 3; ++i) p.taps[i] = taps[i];
  const int64_t total = (int64_t)N * F * H * W * (Cout / epv);
  hipStream_t st = (hipStream_t)stream;
  if (stem_mfma_ok(dtype, p)) return launch_stem_mfma(p, st);
  if (W % 4 == 0 && (Cin == 1 || Cin == 3) && Cout % 4 == 0) {
    const dim3 grid(ew_grid(total / 4));
    if (dtype == MMD_BF16 && Cin == 3) hipLaunchKernelGGL((stem_conv_strip_kernel<__bf16, 3>), grid, dim3(256), lds, st, p);
    else if (dtype == MMD_BF16) hipLaunchKernelGGL((stem_conv_strip_kernel<__bf16, 1>), grid, dim3(256), lds, st, p);
    else if (Cin == 3) hipLaunchKernelGGL((stem_conv_strip_kernel<float, 3>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((stem_conv_strip_kernel<float, 1>), grid, dim3(256), lds, st, p);
    return mmd_check_launch("stem_conv_strip");
  }
  if (dtype == MMD_BF16) hipLaunchKernelGGL(stem_conv_kernel<__bf16>, dim3(ew_grid(total)), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(stem_conv_kernel<float>, dim3(ew_grid(total)), dim3(256), lds, st, p);
  return mmd_check_launch("stem_conv");
}

// Cooperative head conv: LPR = Cin/EPV lanes share one output row (each lane owns one 16-byte channel chunk, so every
// tap is ONE coalesced row read), partial dot products are reduced across the row's lanes with xor-shuffles.
// Weights sit in LDS as [tap][quad j][lane chunk][4 floats] so the 16 lanes of a row read 256 contiguous bytes
// (conflict-free) and the row groups of a wave broadcast.
__device__ __attribute__((aligned(16))) uint32_t g_zero_page_misc[8] = {0, 0, 0, 0, 0, 0, 0, 0};

template <typename T, int CO, int LPR>
__global__ __launch_bounds__(256) void head_conv_coop_kernel(const HeadConvParams p) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  constexpr int NQ = EPV * CO / 4;       // float4 quads of weights per (tap, lane)
  constexpr int RPW = 64 / LPR;          // rows per wave pass
  extern __shared__ __attribute__((aligned(16))) float sw[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < p.ntaps * NQ * LPR * 4; i += 256) {
    const int k = i & 3, cvi = (i >> 2) % LPR, j = ((i >> 2) / LPR) % NQ, t = (i >> 2) / (LPR * NQ);
    const int ec = j * 4 + k, e = ec / CO, c = ec % CO;
    sw[i] = c < p.Co ? p.w[((int64_t)t * p.Cin + cvi * EPV + e) * p.Co + c] : 0.f;
  }
  __syncthreads();
  const int HW = p.H * p.W;
  const int64_t rows = (int64_t)p.N * p.F * HW;
  const int cvi = lane % LPR, rsel = lane / LPR;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (tid >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t mb = wave_id * RPW; mb < rows; mb += nwaves * RPW) {
    const int64_t m = mb + rsel;
    const bool rok = m < rows;
    const int64_t mm = rok ? m : 0;
    const int w0 = (int)(mm % p.W), h0 = (int)((mm / p.W) % p.H), f0 = (int)((mm / HW) % p.F);
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    // taps in groups of 9: all 9 row reads are issued branch-free (padding -> zero page) before any is consumed
    for (int tg = 0; tg < p.ntaps; tg += 9) {
      u32x4 v[9];
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const int t = min(tg + u, p.ntaps - 1);
        const int df = p.taps[t * 3], dh = p.taps[t * 3 + 1], dw = p.taps[t * 3 + 2];
        const bool ok = rok && (tg + u < p.ntaps) && (unsigned)(f0 + df) < (unsigned)p.F && (unsigned)(h0 + dh) < (unsigned)p.H &&
                        (unsigned)(w0 + dw) < (unsigned)p.W;
        const int64_t src = mm + (int64_t)df * HW + dh * p.W + dw;
        const char* sp = ok ? p.x + (src * p.ldx + (int64_t)cvi * EPV) * ES : (const char*)g_zero_page_misc;
        v[u] = *(const u32x4*)sp;
      }
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        if (tg + u < p.ntaps) {
          float f[EPV];
          Elt<T>::unpack(v[u], f);
          const float* wq = sw + ((int64_t)(tg + u) * NQ * LPR + cvi) * 4;
#pragma unroll
          for (int j = 0; j < NQ; ++j) {
            const f32x4 w4 = *(const f32x4*)(wq + j * LPR * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int ec = j * 4 + k;
              acc[ec % CO] += f[ec / CO] * w4[k];
            }
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) {
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
    }
    if (rok && cvi == 0) {
      const int64_t n = m / ((int64_t)HW * p.F);
      const int hw = h0 * p.W + w0;
      for (int c = 0; c < p.Co; ++c) p.y[((n * p.F + f0) * p.Co + c) * HW + hw] = acc[c] + (p.bias ? p.bias[c] : 0.f);
    }
  }
}

// Strip variant of the cooperative kernel (W % 4 == 0): the LPR lanes of a group own FOUR consecutive output pixels, so
// every weight quad read from LDS feeds four pixels (the per-row version re-reads all ntaps*Cin*CO weights per output row:
// 14 GB of LDS returns for the 16x64x64 head, 355 us), CO is the exact output width (3, not 4), and twelve row reads are in
// flight per tap group.
template <typename T, int CO, int LPR>
__global__ __launch_bounds__(256) void head_conv_strip_kernel(const HeadConvParams p) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  constexpr int PX = 4;
  constexpr int NQ = EPV * CO / 4;       // float4 quads of weights per (tap, lane)
  constexpr int SPW = 64 / LPR;          // strips per wave pass
  extern __shared__ __attribute__((aligned(16))) float sw[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < p.ntaps * NQ * LPR * 4; i += 256) {
    const int k = i & 3, cvi = (i >> 2) % LPR, j = ((i >> 2) / LPR) % NQ, t = (i >> 2) / (LPR * NQ);
    const int ec = j * 4 + k, e = ec / CO, c = ec % CO;
    sw[i] = p.w[((int64_t)t * p.Cin + cvi * EPV + e) * CO + c];
  }
  __syncthreads();
  const int HW = p.H * p.W, WS = p.W / PX;
  const int64_t strips = (int64_t)p.N * p.F * p.H * WS;
  const int cvi = lane % LPR, rsel = lane / LPR;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (tid >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t sb = wave_id * SPW; sb < strips; sb += nwaves * SPW) {
    const bool rok = sb + rsel < strips;
    int64_t sidx = rok ? sb + rsel : 0;
    const int w0 = (int)(sidx % WS) * PX;
    sidx /= WS;
    const int h0 = (int)(sidx % p.H);
    sidx /= p.H;
    const int f0 = (int)(sidx % p.F);
    const int64_t n = sidx / p.F;
    const int64_t m0 = ((n * p.F + f0) * p.H + h0) * (int64_t)p.W + w0;
    float acc[PX][CO];
#pragma unroll
    for (int px = 0; px < PX; ++px)
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[px][c] = 0.f;
    for (int tg = 0; tg < p.ntaps; tg += 3) {
      u32x4 v[3][PX];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int t = min(tg + u, p.ntaps - 1);
        const int df = p.taps[t * 3], dh = p.taps[t * 3 + 1], dw = p.taps[t * 3 + 2];
        const bool okfh = rok && (tg + u < p.ntaps) && (unsigned)(f0 + df) < (unsigned)p.F && (unsigned)(h0 + dh) < (unsigned)p.H;
        const int64_t src = m0 + (int64_t)df * HW + dh * p.W + dw;
#pragma unroll
        for (int px = 0; px < PX; ++px) {
          const bool ok = okfh && (unsigned)(w0 + px + dw) < (unsigned)p.W;
          const char* sp = ok ? p.x + ((src + px) * p.ldx + (int64_t)cvi * EPV) * ES : (const char*)g_zero_page_misc;
          v[u][px] = *(const u32x4*)sp;
        }
      }
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (tg + u < p.ntaps) {
          const float* wq = sw + ((int64_t)(tg + u) * NQ * LPR + cvi) * 4;
          float wv[NQ * 4];
#pragma unroll
          for (int j = 0; j < NQ; ++j) {
            const f32x4 w4 = *(const f32x4*)(wq + j * LPR * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) wv[j * 4 + k] = w4[k];
          }
#pragma unroll
          for (int px = 0; px < PX; ++px) {
            float f[EPV];
            Elt<T>::unpack(v[u][px], f);
#pragma unroll
            for (int e = 0; e < EPV; ++e)
#pragma unroll
              for (int c = 0; c < CO; ++c) acc[px][c] += f[e] * wv[e * CO + c];
          }
        }
      }
    }
#pragma unroll
    for (int px = 0; px < PX; ++px)
#pragma unroll
      for (int c = 0; c < CO; ++c) {
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) acc[px][c] += __shfl_xor(acc[px][c], o, 64);
      }
    if (rok && cvi == 0) {
      const int hw = h0 * p.W + w0;
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        const float b = p.bias ? p.bias[c] : 0.f;
        const f32x4 o4 = {acc[0][c] + b, acc[1][c] + b, acc[2][c] + b, acc[3][c] + b};
        *(f32x4*)(p.y + ((n * p.F + f0) * CO + c) * HW + hw) = o4;
      }
    }
  }
}

template <typename T, int CO>
static int launch_head_strip(const HeadConvParams& p, int lpr, hipStream_t st) {
  constexpr int EPV = Elt<T>::EPV;
  const size_t lds = (size_t)p.ntaps * (EPV * CO / 4) * lpr * 4 * sizeof(float);
  const int64_t strips = (int64_t)p.N * p.F * p.H * (p.W / 4);
  const int spb = 4 * (64 / lpr);                       // strips per block pass
  const int grid = (int)min((int64_t)2048, (strips + spb - 1) / spb);
#define MMD_HEADS_LAUNCH(L)                                                                                         \
  do {                                                                                                              \
    if (lds > 64 * 1024) {                                                                                          \
      hipError_t e = hipFuncSetAttribute((const void*)head_conv_strip_kernel<T, CO, L>,                             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
      if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "head_conv: set LDS attr: %s", hipGetErrorString(e)); \
    }                                                                                                               \
    hipLaunchKernelGGL((head_conv_strip_kernel<T, CO, L>), dim3(grid), dim3(256), lds, st, p);                      \
  } while (0)
  switch (lpr) {
    case 4: MMD_HEADS_LAUNCH(4); break;
    case 8: MMD_HEADS_LAUNCH(8); break;
    case 16: MMD_HEADS_LAUNCH(16); break;
    default: MMD_HEADS_LAUNCH(32); break;
  }
#undef MMD_HEADS_LAUNCH
  return mmd_check_launch("head_conv_strip");
}

template <typename T, int CO>
static int launch_head_coop(const HeadConvParams& p, int lpr, hipStream_t st) {
  constexpr int EPV = Elt<T>::EPV;
  const size_t lds = (size_t)p.ntaps * (EPV * CO / 4) * lpr * 4 * sizeof(float);
  const int64_t rows = (int64_t)p.N * p.F * p.H * p.W;
  const int grid = (int)min((int64_t)2048, (rows + 63) / 64);
#define MMD_HEAD_LAUNCH(L)                                                                                         \
  do {                                                                                                             \
    if (lds > 64 * 1024) {                                                                                         \
      hipError_t e = hipFuncSetAttribute((const void*)head_conv_coop_kernel<T, CO, L>,                             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                    \
      if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "head_conv: set LDS attr: %s", hipGetErrorString(e)); \
    }                                                                                                              \
    hipLaunchKernelGGL((head_conv_coop_kernel<T, CO, L>), dim3(grid), dim3(256), lds, st, p);                      \
  } while (0)
  switch (lpr) {
    case 4: MMD_HEAD_LAUNCH(4); break;
    case 8: MMD_HEAD_LAUNCH(8); break;
    case 16: MMD_HEAD_LAUNCH(16); break;
    case 32: MMD_HEAD_LAUNCH(32); break;
    case 64: MMD_HEAD_LAUNCH(64); break;
    default: return mmd_set_error(MMD_ERR_UNSUPPORTED, "head_conv: lanes per row %d", lpr);
  }
#undef MMD_HEAD_LAUNCH
  return mmd_check_launch("head_conv_coop");
}


// (Round 3 tried the head as a GEMM on 32x32x16 MFMAs with the X fragments read straight from global memory - the output channels as
// the mostly empty M side, fp32 weights split into two bf16 parts.  Correct, and slower: 270 us against the strip kernel's 160 us.  All
// 27 taps re-read the tensor through the texture path, 1.8 GB per launch at the ~8 TB/s that 16-byte-per-lane row reads sustain; a
// version that pays would stage a three-frame halo in LDS like the 3x3 conv tiles do.  Not built: the head is 1.3 % of the step.)

template <typename T>
static int launch_head(const HeadConvParams& p, hipStream_t st) {
  const int64_t rows = (int64_t)p.N * p.F * p.H * p.W;
  const int grid = (int)min((int64_t)8192, (rows + 255) / 256);
  {   // strip kernel: four pixels per lane group, exact output width
    const int lpr = p.Cin / Elt<T>::EPV;
    const bool lpr_ok = p.Cin % Elt<T>::EPV == 0 && (lpr == 4 || lpr == 8 || lpr == 16 || lpr == 32);
    const size_t lds_s = (size_t)p.ntaps * Elt<T>::EPV * p.Co * lpr * sizeof(float);
    if (p.W % 4 == 0 && ((uintptr_t)p.y) % 16 == 0 && lpr_ok && lds_s <= 150 * 1024) {
      switch (p.Co) {
        case 1: return launch_head_strip<T, 1>(p, lpr, st);
        case 2: return launch_head_strip<T, 2>(p, lpr, st);
        case 3: return launch_head_strip<T, 3>(p, lpr, st);
        case 6: return launch_head_strip<T, 6>(p, lpr, st);
        default: break;
      }
    }
  }
  const int CO = p.Co <= 2 ? 2 : (p.Co <= 4 ? 4 : 8);
  {   // cooperative kernel whenever the channel chunks of a row map onto a power-of-two lane group
    const int lpr = p.Cin / Elt<T>::EPV;
    const size_t lds_c = (size_t)p.ntaps * (Elt<T>::EPV * CO / 4) * lpr * 4 * sizeof(float);
    if (p.Cin % Elt<T>::EPV == 0 && (lpr == 4 || lpr == 8 || lpr == 16 || lpr == 32 || lpr == 64) && lds_c <= 150 * 1024) {
      if (CO == 2) return launch_head_coop<T, 2>(p, lpr, st);
      if (CO == 4) return launch_head_coop<T, 4>(p, lpr, st);
      return launch_head_coop<T, 8>(p, lpr, st);
    }
  }
  const size_t lds = (size_t)p.ntaps * p.Cin * CO * sizeof(float);
  if (lds > 150 * 1024) return mmd_set_error(MMD_ERR_UNSUPPORTED, "head_conv: weights (%zu B) exceed LDS", lds);
  const void* fn = CO == 2 ? (const void*)head_conv_kernel<T, 2> : CO == 4 ? (const void*)head_conv_kernel<T, 4> : (const void*)head_conv_kernel<T, 8>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "head_conv: set LDS attr: %s", hipGetErrorString(e));
  }
  if (CO == 2) hipLaunchKernelGGL((head_conv_kernel<T, 2>), dim3(grid), dim3(256), lds, st, p);
  else if (CO == 4) hipLaunchKernelGGL((head_conv_kernel<T, 4>), dim3(grid), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((head_conv_kernel<T, 8>), dim3(grid), dim3(256), lds, st, p);
  return mmd_check_launch("head_conv");
}

// ----------------------------------------------------------------------------- head conv as GEMM + gather (round 5, bf16)
// The head (GroupNorm32 + SiLU + Conv3d 3x3x3, 128 -> 3 channels, unet:1003-1012) was the video stream's LAST launch pair and its
// slowest HBM-side kernel: gn_apply wrote the normalised tensor (134 MB of traffic) and head_conv_strip read it 27 times through L1 / L2
// (0.45 TB/s).  A convolution with few output channels factors the other way round: FIRST the per-row products
//     P[o, m] = sum_ci W[tap, ci, co] act(norm(x))[m, ci],   o = tap Co + co   (a GEMM with N = ntaps Co = 81 columns, K = Cin),
// with the norm applied in registers on the way into the MFMA operand (x is read ONCE, nothing normalised is written), THEN
//     y[n, f, co, h, w] = bias[co] + sum_tap P[tap Co + co, m + offset(tap)]    (zero outside the frame),
// a pure gather over fp32 planes P[o][m] that are contiguous in m (coalesced along w) and read exactly once.
// Weights enter the matrix pipe as a bf16 (hi, lo) pair, so the products keep the fp32 weights to 2^-17 (the direct kernel uses fp32
// weights); the activations are rounded to bf16 exactly where gn_apply used to store them.
struct HeadGemmParams {
  const char* x; int64_t ldx; int64_t M;
  const float* gn_a; const float* gn_b; int64_t gn_rows; int gn_S; int act;
  const char* wimg;          // [2 hi/lo][3 blocks of 32 outputs][KS k-steps][64 lanes][16 B]: lane (l31, half) = W[32 ob + l31][16 cg + 8 half .. + 8]
  float* P;                  // [NO][M] fp32 planes
  int NO;                    // ntaps * Co <= 96
  int per_block;             // consecutive 128-row groups per block
};
template <int KS>
__global__ __launch_bounds__(256, 2) void head_gemm_kernel(const HeadGemmParams p) {
  constexpr int C = 16 * KS, WIMG_B = 2 * 3 * KS * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;
  float* sGN = (float*)(smem + WIMG_B);                  // [a | b][C] of the current slice
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  for (int i = tid; i < WIMG_B / 16; i += 256) *(u32x4*)(sW + i * 16) = *(const u32x4*)(p.wimg + i * 16);
  const int64_t ngroups = (p.M + 127) / 128;
  const int64_t g0 = (int64_t)blockIdx.x * p.per_block, g1 = min(g0 + p.per_block, ngroups);
  int cur_slice = -1;
  for (int64_t g = g0; g < g1; ++g) {
    const int64_t m = g * 128 + wave * 32 + l31;
    const bool ok = m < p.M;
    const int64_t mc = ok ? m : p.M - 1;
    u32x4 xa[KS];
    const char* ap = p.x + (mc * p.ldx + half * 8) * 2;
#pragma unroll
    for (int cg = 0; cg < KS; ++cg) xa[cg] = *(const u32x4*)(ap + cg * 32);
    const int slice = (int)((g * 128) / p.gn_rows);       // gn_rows % 128 == 0: a 128-row group lies inside one slice (block-uniform)
    if (slice != cur_slice) {
      __syncthreads();                                     // every wave is past its reads of the previous table (and of nothing, first time)
      for (int i = tid; i < 2 * C; i += 256) sGN[i] = (i < C ? p.gn_a : p.gn_b)[(int64_t)slice * C + (i < C ? i : i - C)];
      cur_slice = slice;
      __syncthreads();                                     // (also covers the weight image on the first pass)
    }
#pragma unroll
    for (int cg = 0; cg < KS; ++cg) {
      float v[8];
      Elt<__bf16>::unpack(xa[cg], v);
      const float* a4 = sGN + cg * 16 + half * 8;
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        const f32x4 av = *(const f32x4*)(a4 + e), bv = *(const f32x4*)(a4 + C + e);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float y = v[e + k] * av[k] + bv[k];
          v[e + k] = p.act ? silu_f(y) : y;
        }
      }
      u32x4 y = Elt<__bf16>::pack(v);
      asm volatile("" : "+v"(y.x), "+v"(y.y), "+v"(y.z), "+v"(y.w));      // keep the normalisation here (see the strip GEMM)
      xa[cg] = y;
    }
#pragma unroll
    for (int ob = 0; ob < 3; ++ob) {
      if (ob * 32 >= p.NO) break;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int hl = 0; hl < 2; ++hl)
#pragma unroll
        for (int cg = 0; cg < KS; ++cg) {
          const u32x4 fw = *(const u32x4*)(sW + (((hl * 3 + ob) * KS + cg) * 64 + lane) * 16);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fw), __builtin_bit_cast(bf16x8, xa[cg]), acc, 0, 0, 0);
        }
      // acc[4 q + j] = output 32 ob + 8 q + 4 half + j of row m: lanes 0 - 31 of a register are 32 consecutive m of one plane (128 bytes)
      if (ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int o = ob * 32 + 8 * q + 4 * half + j;
            if (o < p.NO) p.P[(int64_t)o * p.M + m] = acc[4 * q + j];
          }
      }
    }
  }
}

struct HeadGatherParams {
  const float* P; int64_t M; const float* bias; float* y;
  int N, F, H, W, Co, ntaps;
  int taps[27 * 3];
};
template <int CO>
__global__ __launch_bounds__(256) void head_gather_kernel(const HeadGatherParams p) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= p.M) return;
  const int w = (int)(m % p.W), h = (int)((m / p.W) % p.H);
  const int64_t nf = m / ((int64_t)p.W * p.H);
  const int f = (int)(nf % p.F);
  const int64_t HW = (int64_t)p.H * p.W;
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = p.bias ? p.bias[c] : 0.f;
  // nine taps per trip: their 9 CO loads are independent and issue together (branch-free: a tap outside the frame reads the centre
  // element and is multiplied by zero); the sum runs in tap order
  for (int t0 = 0; t0 < p.ntaps; t0 += 9) {
    float v[9][CO], k[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int t = min(t0 + u, p.ntaps - 1);
      const int df = p.taps[3 * t], dh = p.taps[3 * t + 1], dw = p.taps[3 * t + 2];
      const bool ok = t0 + u < p.ntaps && (unsigned)(f + df) < (unsigned)p.F && (unsigned)(h + dh) < (unsigned)p.H && (unsigned)(w + dw) < (unsigned)p.W;
      const int64_t src = ok ? m + df * HW + dh * p.W + dw : m;
      k[u] = ok ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < CO; ++c) v[u][c] = p.P[(int64_t)(t * CO + c) * p.M + src];
    }
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
      for (int c = 0; c < CO; ++c) acc[c] += k[u] * v[u][c];
  }
  const int64_t n = nf / p.F;
#pragma unroll
  for (int c = 0; c < CO; ++c) p.y[(((n * p.F + f) * CO + c) * p.H + h) * (int64_t)p.W + w] = acc[c];
}

extern "C" int64_t mmd_head_gemm_weight_bytes(int Cin) { return (Cin == 128) ? 2 * 3 * (Cin / 16) * 1024 : 0; }
extern "C" int64_t mmd_head_gemm_workspace_bytes(int64_t M, int ntaps, int Co) { return (int64_t)ntaps * Co * M * 4; }

// P = W act(x a + b): x bf16 rows [M, Cin] (Cin = 128), a / b fp32 [S, Cin] = the fused GroupNorm affine over S slices of gn_rows rows
// (gn_rows % 128 == 0; a == NULL is not supported: the head always follows its norm), wimg = the packed (hi, lo) weight image
// (mmd_head_gemm_weight_bytes; packed by the host mirror), P fp32 [ntaps * Co][M].
extern "C" int mmd_head_gemm(const void* x, int64_t ldx, int64_t M, int Cin, const float* gn_a, const float* gn_b, int S, int64_t gn_rows,
                             int act, const void* wimg, float* P, int NO, void* stream) {
  MMD_REQUIRE(x && gn_a && gn_b && wimg && P && M > 0, "head_gemm: null pointer / empty");
  MMD_REQUIRE(Cin == 128, "head_gemm: built for 128 input channels (got %d)", Cin);
  MMD_REQUIRE(NO >= 1 && NO <= 96, "head_gemm: 1 .. 96 outputs (taps x channels), got %d", NO);
  MMD_REQUIRE(S > 0 && gn_rows > 0 && gn_rows % 128 == 0 && (int64_t)S * gn_rows == M, "head_gemm: S x gn_rows must tile the rows in multiples of 128");
  MMD_REQUIRE(((uintptr_t)x | (uintptr_t)wimg) % 16 == 0 && ldx % 8 == 0 && (uintptr_t)P % 4 == 0, "head_gemm: unaligned operand");
  HeadGemmParams p;
  p.x = (const char*)x; p.ldx = ldx; p.M = M; p.gn_a = gn_a; p.gn_b = gn_b; p.gn_rows = gn_rows; p.gn_S = S; p.act = act;
  p.wimg = (const char*)wimg; p.P = P; p.NO = NO;
  const int64_t ngroups = (M + 127) / 128;
  p.per_block = (int)max((int64_t)1, (ngroups + 1023) / 1024);          // <= 1024 blocks: two per CU, each a run of consecutive row groups
  const int grid = (int)((ngroups + p.per_block - 1) / p.per_block);
  const size_t lds = 2 * 3 * 8 * 1024 + 2 * 128 * sizeof(float);
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)head_gemm_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "head_gemm: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(head_gemm_kernel<8>, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
  return mmd_check_launch("head_gemm");
}

// y[n, f, co, h, w] = bias[co] + sum_tap P[tap Co + co][m + offset(tap)] (zero outside (F, H, W)); y fp32 API layout [N, F, Co, H, W].
extern "C" int mmd_head_gather(const float* P, const float* bias, float* y, int N, int F, int H, int W, int Co, int ntaps, const int* taps,
                               void* stream) {
  MMD_REQUIRE(P && y && taps && N > 0 && F > 0 && H > 0 && W > 0 && ntaps >= 1 && ntaps <= 27, "head_gather: bad argument");
  MMD_REQUIRE(Co == 1 || Co == 2 || Co == 3 || Co == 4 || Co == 6, "head_gather: Co in {1, 2, 3, 4, 6} (got %d)", Co);
  HeadGatherParams p;
  p.P = P; p.M = (int64_t)N * F * H * W; p.bias = bias; p.y = y; p.N = N; p.F = F; p.H = H; p.W = W; p.Co = Co; p.ntaps = ntaps;
  for (int i = 0; i < ntaps * 3; ++i) p.taps[i] = taps[i];
  const int grid = (int)((p.M + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  switch (Co) {
    case 1: hipLaunchKernelGGL(head_gather_kernel<1>, dim3(grid), dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL(head_gather_kernel<2>, dim3(grid), dim3(256), 0, st, p); break;
    case 3: hipLaunchKernelGGL(head_gather_kernel<3>, dim3(grid), dim3(256), 0, st, p); break;
    case 4: hipLaunchKernelGGL(head_gather_kernel<4>, dim3(grid), dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL(head_gather_kernel<6>, dim3(grid), dim3(256), 0, st, p); break;
  }
  return mmd_check_launch("head_gather");
}

extern "C" int mmd_head_conv(int dtype, const void* x, int64_t ldx, const float* w, const float* bias, float* y, int N, int F,
                             int Cin, int H, int W, int Co, int ntaps, const int* taps, void* stream) {
  const int epv = dtype == MMD_BF16 ? 8 : 4;
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "head_conv: bad dtype");
  MMD_REQUIRE(x && w && y && taps && ntaps >= 1 && ntaps <= 27 && Cin % epv == 0 && Co >= 1 && Co <= 8, "head_conv: bad argument");
  HeadConvParams p;
  p.x = (const char*)x; p.ldx = ldx; p.w = w; p.bias = bias; p.y = y;
  p.N = N; p.F = F; p.Cin = Cin; p.H = H; p.W = W; p.Co = Co; p.ntaps = ntaps;
  for (int i = 0; i < ntaps * 3; ++i) p.taps[i] = taps[i];
  hipStream_t st = (hipStream_t)stream;
  return dtype == MMD_BF16 ? launch_head<__bf16>(p, st) : launch_head<float>(p, st);
}

extern "C" int mmd_ddpm_update(const float* x, const float* model_out, const float* noise, float* out, float* x0_out,
                               float* mean_out, float* logvar_out, const float* tables, const int64_t* t, int T, int N, int F,
                               int C, int HW, int flags, void* stream) {
  MMD_REQUIRE(x && model_out && tables && t && T > 0 && N > 0 && F > 0 && C > 0 && HW > 0, "ddpm_update: bad argument");
  MMD_REQUIRE(!out || noise, "ddpm_update: sampling (out != NULL) needs noise");
  DdpmParams p;
  p.x = x; p.mo = model_out; p.noise = noise; p.out = out; p.x0_out = x0_out; p.mean_out = mean_out; p.logvar_out = logvar_out;
  p.tables = tables; p.t = t;
  p.T = T; p.N = N; p.F = F; p.C = C; p.HW = HW; p.flags = flags;
  hipLaunchKernelGGL(ddpm_update_kernel, dim3(ew_grid((int64_t)N * F * C * HW)), dim3(256), 0, (hipStream_t)stream, p);
  return mmd_check_launch("ddpm_update");
}

extern "C" int mmd_q_sample(const float* x0, const float* eps, float* out, const float* tab2, const int64_t* t, int T, int N,
                            int64_t per_sample, void* stream) {
  MMD_REQUIRE(x0 && eps && out && tab2 && t && T > 0 && N > 0 && per_sample > 0, "q_sample: bad argument");
  const int64_t total = per_sample * N;
  hipLaunchKernelGGL(q_sample_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x0, eps, out, tab2, t, T, per_sample, total);
  return mmd_check_launch("q_sample");
}

#define MMD_LOSS_CHUNKS 64
extern "C" int64_t mmd_loss_workspace_bytes(int N) { return (int64_t)N * MMD_LOSS_CHUNKS * 2 * sizeof(double); }

// Per-sample loss terms of one stream (see loss_terms_kernel).  x0/xt may be NULL without flag 4.  vb_scale = T/1000 for
// RESCALED_MSE else 1.  mse_out/vb_out fp32 [N].
extern "C" int mmd_loss_terms(const float* x0, const float* xt, const float* model_out, const float* target, const float* tables,
                              const int64_t* t, int T, int N, int F, int C, int HW, int flags, float vb_scale, float* mse_out,
                              float* vb_out, void* workspace, void* stream) {
  MMD_REQUIRE(model_out && target && tables && t && mse_out && workspace && T > 0 && N > 0 && F > 0 && C > 0 && HW > 0, "loss_terms: bad argument");
  MMD_REQUIRE(!(flags & 4) || (x0 && xt && vb_out), "loss_terms: the vb term needs x0, x_t and vb_out");
  LossParams p;
  p.x0 = x0; p.xt = xt; p.mo = model_out; p.target = target; p.tables = tables; p.t = t; p.partial = (double*)workspace;
  p.T = T; p.N = N; p.F = F; p.C = C; p.HW = HW; p.flags = flags; p.nchunk = MMD_LOSS_CHUNKS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_terms_kernel, dim3(MMD_LOSS_CHUNKS, N), dim3(256), 0, st, p);
  int rc = mmd_check_launch("loss_terms");
  if (rc) return rc;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(N), 0, st, (const double*)workspace, MMD_LOSS_CHUNKS,
                     1.0 / ((double)F * C * HW), vb_scale, mse_out, (flags & 4) ? vb_out : nullptr);
  return mmd_check_launch("loss_finalize");
}

// sinusoidal timestep embedding alone (nn.py:192-210): out[N, dim] fp32 (training path keeps the MLP as separate linears)
__global__ void timestep_embedding_kernel(const void* __restrict__ t, int t_kind, int dim, float* __restrict__ out) {
  const int n = blockIdx.x;
  float tv;
  if (t_kind == 0) tv = (float)((const int64_t*)t)[n];
  else if (t_kind == 1) tv = (float)((const int32_t*)t)[n];
  else tv = ((const float*)t)[n];
  const int half = dim / 2;
  for (int i = threadIdx.x; i < dim; i += blockDim.x) {
    float v = 0.f;
    if (i < 2 * half) {
      const int k = i < half ? i : i - half;
      const float a = tv * expf(-logf(10000.f) * (float)k / (float)half);
      v = i < half ? cosf(a) : sinf(a);
    }
    out[(int64_t)n * dim + i] = v;
  }
}
extern "C" int mmd_timestep_embedding(const void* t, int t_kind, int N, int dim, float* out, void* stream) {
  MMD_REQUIRE(t && out && N > 0 && dim > 0 && t_kind >= 0 && t_kind <= 2, "timestep_embedding: bad argument");
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(N), dim3(128), 0, (hipStream_t)stream, t, t_kind, dim, out);
  return mmd_check_launch("timestep_embedding");
}

// ----------------------------------------------------------------------------- DDIM step / helper combinations
// ddim_sample (gd:821-901) and ddim_reverse_sample (gd:903-953) for one stream, API layout [N, F, Cm, HW]:
//   x0 = eps-or-x0 prediction (clamped with flag 1), eps = (sqrt_recip_ac x - x0) / sqrt_recipm1_ac,
//   sigma = eta sqrt((1-ac_prev)/(1-ac)) sqrt(1 - ac/ac_prev),
//   out = x0 sqrt(ac_prev) + sqrt(1 - ac_prev - sigma^2) eps + [t != 0] sigma noise          (flag 8: reverse ODE with ac_next, no noise)
// tab3 = [3][T] fp32: alphas_cumprod, alphas_cumprod_prev, alphas_cumprod_next.
struct DdimParams {
  const float* x; const float* mo; const float* noise;
  float* out; float* x0_out;
  const float* tables; const float* tab3; const int64_t* t;
  int T, N, F, C, HW, flags;
  float eta;
};
__global__ __launch_bounds__(256) void ddim_update_kernel(const DdimParams p) {
  const int64_t per = (int64_t)p.F * p.C * p.HW;
  const int64_t total = per * p.N;
  const int Cm = (p.flags & 4) ? 2 * p.C : p.C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / per, r = i % per;
    const int hw = (int)(r % p.HW), c = (int)((r / p.HW) % p.C);
    const int64_t f = r / ((int64_t)p.HW * p.C);
    const int ti = (int)p.t[n];
    const float cr = p.tables[ti], crm1 = p.tables[p.T + ti];
    const float o = p.mo[((n * p.F + f) * Cm + c) * (int64_t)p.HW + hw];
    const float xv = p.x[i];
    float x0 = (p.flags & 2) ? o : cr * xv - crm1 * o;
    if (p.flags & 1) x0 = fminf(fmaxf(x0, -1.f), 1.f);
    const float eps = (cr * xv - x0) / crm1;
    float res;
    if (p.flags & 8) {
      const float an = p.tab3[2 * p.T + ti];
      res = x0 * sqrtf(an) + sqrtf(1.f - an) * eps;
    } else {
      const float ab = p.tab3[ti], ap = p.tab3[p.T + ti];
      const float sigma = p.eta * sqrtf((1.f - ap) / (1.f - ab)) * sqrtf(1.f - ab / ap);
      const float mean = x0 * sqrtf(ap) + sqrtf(1.f - ap - sigma * sigma) * eps;
      const float nz = ti != 0 ? 1.f : 0.f;
      res = mean + nz * sigma * (p.noise ? p.noise[i] : 0.f);
    }
    if (p.out) p.out[i] = res;
    if (p.x0_out) p.x0_out[i] = x0;
  }
}
extern "C" int mmd_ddim_update(const float* x, const float* model_out, const float* noise, float* out, float* x0_out,
                               const float* tables, const float* tab3, const int64_t* t, int T, int N, int F, int C, int HW,
                               int flags, float eta, void* stream) {
  MMD_REQUIRE(x && model_out && tables && tab3 && t && T > 0 && N > 0 && F > 0 && C > 0 && HW > 0, "ddim_update: bad argument");
  MMD_REQUIRE((flags & 8) || eta == 0.f || noise, "ddim_update: eta > 0 needs noise");
  DdimParams p;
  p.x = x; p.mo = model_out; p.noise = noise; p.out = out; p.x0_out = x0_out; p.tables = tables; p.tab3 = tab3; p.t = t;
  p.T = T; p.N = N; p.F = F; p.C = C; p.HW = HW; p.flags = flags; p.eta = eta;
  hipLaunchKernelGGL(ddim_update_kernel, dim3(ew_grid((int64_t)N * F * C * HW)), dim3(256), 0, (hipStream_t)stream, p);
  return mmd_check_launch("ddim_update");
}

// out[n, i] = (ca[t_n] a + cb[t_n] b) * cs[t_n]    per-sample coefficients looked up from fp32 tables of length T
// (ca / cb / cs may be NULL = 1; b may be NULL).  Covers _predict_xstart_from_eps, _predict_eps_from_xstart,
// _predict_xstart_from_xprev, q_posterior mean, q_mean (gd:170-229,345-366).
__global__ __launch_bounds__(256) void lincomb_t_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                        const float* __restrict__ ca, const float* __restrict__ cb, const float* __restrict__ cs,
                                                        const int64_t* __restrict__ t, int64_t per, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ti = (int)t[i / per];
    float v = (ca ? ca[ti] : 1.f) * a[i];
    if (b) v += (cb ? cb[ti] : 1.f) * b[i];
    out[i] = cs ? v * cs[ti] : v;
  }
}
extern "C" int mmd_lincomb_t(const float* a, const float* b, float* out, const float* ca, const float* cb, const float* cs,
                             const int64_t* t, int N, int64_t per_sample, void* stream) {
  MMD_REQUIRE(a && out && t && N > 0 && per_sample > 0, "lincomb_t: bad argument");
  const int64_t total = per_sample * N;
  hipLaunchKernelGGL(lincomb_t_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, a, b, out, ca, cb, cs, t, per_sample, total);
  return mmd_check_launch("lincomb_t");
}

// out = ca a + cb b + cc c with host scalars (b, c nullable): the DPM-Solver update combinations (dpm:520-1100).
__global__ __launch_bounds__(256) void lincomb_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                      float* __restrict__ out, float ca, float cb, float cc, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float v = ca * a[i];
    if (b) v += cb * b[i];
    if (c) v += cc * c[i];
    out[i] = v;
  }
}
extern "C" int mmd_lincomb(const float* a, float ca, const float* b, float cb, const float* c, float cc, float* out, int64_t n,
                           void* stream) {
  MMD_REQUIRE(a && out && n > 0, "lincomb: bad argument");
  hipLaunchKernelGGL(lincomb_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, a, b, c, out, ca, cb, cc, n);
  return mmd_check_launch("lincomb");
}

// Gradient payload conversion of the data-parallel all-reduce (optim.FlatAdamW, grad_payload = "bf16"): y = (T_out)(x * scale).
__global__ __launch_bounds__(256) void cast_kernel(const void* __restrict__ x, void* __restrict__ y, int src_bf16, int dst_bf16, float scale,
                                                   int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = (src_bf16 ? Elt<__bf16>::ld(x, i) : ((const float*)x)[i]) * scale;
    if (dst_bf16) Elt<__bf16>::st(y, i, v);
    else ((float*)y)[i] = v;
  }
}
extern "C" int mmd_cast(const void* x, int src_dtype, void* y, int dst_dtype, float scale, int64_t n, void* stream) {
  MMD_REQUIRE(x && y && n > 0, "cast: bad argument");
  MMD_REQUIRE((src_dtype == MMD_F32 || src_dtype == MMD_BF16) && (dst_dtype == MMD_F32 || dst_dtype == MMD_BF16), "cast: bad dtype");
  hipLaunchKernelGGL(cast_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, src_dtype == MMD_BF16, dst_dtype == MMD_BF16,
                     scale, n);
  return mmd_check_launch("cast");
}

// Backward of the sampling update through the posterior mean (gradient-guided conditional sampling, gd:722-817):
//   sample = c1 clamp(x0) + c2 x + noise term,  x0 = cr x - crm1 eps  (or x0 = model output with flag 2)
//   dx = dsample (c1 cr [|x0| <= 1] + c2),  dmo = dsample (-c1 crm1 [|x0| <= 1])   (fixed variance only)
__global__ __launch_bounds__(256) void ddpm_update_bwd_kernel(const DdpmParams p, const float* __restrict__ ds, float* __restrict__ dx,
                                                              float* __restrict__ dmo) {
  const int64_t per = (int64_t)p.F * p.C * p.HW;
  const int64_t total = per * p.N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ti = (int)p.t[i / per];
    const float cr = p.tables[ti], crm1 = p.tables[p.T + ti], c1 = p.tables[2 * p.T + ti], c2 = p.tables[3 * p.T + ti];
    const float o = p.mo[i], xv = p.x[i], g = ds[i];
    const float x0 = (p.flags & 2) ? o : cr * xv - crm1 * o;
    const float pass = (!(p.flags & 1) || (x0 >= -1.f && x0 <= 1.f)) ? 1.f : 0.f;
    if (dx) dx[i] = g * (((p.flags & 2) ? 0.f : c1 * cr * pass) + c2);
    if (dmo) dmo[i] = g * ((p.flags & 2) ? c1 * pass : -c1 * crm1 * pass);
  }
}
extern "C" int mmd_ddpm_update_bwd(const float* x, const float* model_out, const float* dsample, float* dx, float* dmodel_out,
                                   const float* tables, const int64_t* t, int T, int N, int64_t per_sample, int flags, void* stream) {
  MMD_REQUIRE(x && model_out && dsample && tables && t && T > 0 && N > 0 && per_sample > 0, "ddpm_update_bwd: bad argument");
  MMD_REQUIRE(!(flags & 4), "ddpm_update_bwd: learned variance is not differentiable here");
  DdpmParams p;
  p.x = x; p.mo = model_out; p.noise = nullptr; p.out = nullptr; p.x0_out = nullptr; p.mean_out = nullptr; p.logvar_out = nullptr;
  p.tables = tables; p.t = t; p.T = T; p.N = N; p.F = 1; p.C = 1; p.HW = (int)per_sample; p.flags = flags;
  hipLaunchKernelGGL(ddpm_update_bwd_kernel, dim3(ew_grid((int64_t)N * per_sample)), dim3(256), 0, (hipStream_t)stream, p, dsample, dx, dmodel_out);
  return mmd_check_launch("ddpm_update_bwd");
}

// ----------------------------------------------------------------------------- DPM-Solver helpers
// Dynamic thresholding of the x0 prediction (multimodal_dpm_solver_plus.py:419-440): per sample, s = the p-quantile of
// |x0| (torch.quantile 'linear' interpolation), s = max(s, 1), x0 = clamp(x0, -s, s) / (s / max_val).
// Exact selection: |x| as IEEE bits is order-preserving for non-negative floats -> 4 passes of an 8-bit radix select per
// wanted rank; one 1024-thread block per sample.
__device__ __forceinline__ uint32_t absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

__device__ uint32_t radix_select_block(const float* __restrict__ x, int64_t n, int64_t rank, uint32_t* hist, int tid, int nth) {
  uint32_t prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 256; i += nth) hist[i] = 0;
    __syncthreads();
    for (int64_t i = tid; i < n; i += nth) {
      const uint32_t b = absbits(x[i]);
      if ((b & mask) == prefix) atomicAdd(&hist[(b >> shift) & 255u], 1u);
    }
    __syncthreads();
    // every thread walks the 256 bins (uniform result, no extra broadcast)
    int64_t r = rank;
    uint32_t digit = 0;
    for (int d = 0; d < 256; ++d) {
      const uint32_t c = hist[d];
      if (r < (int64_t)c) { digit = (uint32_t)d; break; }
      r -= c;
    }
    rank = r;
    prefix |= digit << shift;
    mask |= 255u << shift;
    __syncthreads();
  }
  return prefix;
}

__global__ __launch_bounds__(1024) void abs_quantile_kernel(const float* __restrict__ x, int64_t per, float q, float* __restrict__ out) {
  __shared__ uint32_t hist[256];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* xs = x + (int64_t)n * per;
  const double pos = (double)q * (double)(per - 1);
  const int64_t lo = (int64_t)floor(pos);
  const int64_t hi = lo + 1 < per ? lo + 1 : lo;
  const float frac = (float)(pos - (double)lo);
  const float vlo = __uint_as_float(radix_select_block(xs, per, lo, hist, tid, blockDim.x));
  const float vhi = hi == lo ? vlo : __uint_as_float(radix_select_block(xs, per, hi, hist, tid, blockDim.x));
  if (tid == 0) out[n] = vlo + (vhi - vlo) * frac;          // torch.lerp(lo, hi, frac) for frac < 0.5 and its mirror agree to 1 ulp
}
extern "C" int mmd_abs_quantile(const float* x, int N, int64_t per_sample, float q, float* out, void* stream) {
  MMD_REQUIRE(x && out && N > 0 && per_sample > 0 && q >= 0.f && q <= 1.f, "abs_quantile: bad argument");
  hipLaunchKernelGGL(abs_quantile_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, x, per_sample, q, out);
  return mmd_check_launch("abs_quantile");
}

// x[n, :] = clamp(x, -s_n, s_n) / (s_n / max_val),  s_n = max(s[n], 1)        (in place)
__global__ __launch_bounds__(256) void clamp_scale_kernel(float* __restrict__ x, const float* __restrict__ s, float max_val, int64_t per,
                                                          int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float sn = fmaxf(s[i / per], 1.f);
    x[i] = fminf(fmaxf(x[i], -sn), sn) / (sn / max_val);
  }
}
extern "C" int mmd_clamp_scale(float* x, const float* s, float max_val, int N, int64_t per_sample, void* stream) {
  MMD_REQUIRE(x && s && N > 0 && per_sample > 0 && max_val > 0.f, "clamp_scale: bad argument");
  const int64_t total = per_sample * N;
  hipLaunchKernelGGL(clamp_scale_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, x, s, max_val, per_sample, total);
  return mmd_check_launch("clamp_scale");
}

// Adaptive step-size error term (dpm:1088-1149): out[n] += sum_i ((hi - lo) / max(atol, rtol * max(|lo|, |prev|)))^2
// (fp64 atomics; caller zeroes out and takes sqrt(out / per)).
__global__ __launch_bounds__(256) void dpm_err_kernel(const float* __restrict__ hi, const float* __restrict__ lo, const float* __restrict__ prev,
                                                      float atol, float rtol, int64_t per, double* __restrict__ out) {
  __shared__ double red[256];
  const int n = blockIdx.y;
  const int64_t base = (int64_t)n * per;
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
    const float l = lo[base + i];
    const float delta = fmaxf(atol, rtol * fmaxf(fabsf(l), fabsf(prev[base + i])));
    const float e = (hi[base + i] - l) / delta;
    acc += (double)e * (double)e;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out + n, red[0]);
}
extern "C" int mmd_dpm_err(const float* hi, const float* lo, const float* prev, float atol, float rtol, int N, int64_t per_sample,
                           double* out, void* stream) {
  MMD_REQUIRE(hi && lo && prev && out && N > 0 && per_sample > 0, "dpm_err: bad argument");
  const int chunks = (int)((per_sample + 256 * 16 - 1) / (256 * 16));
  hipLaunchKernelGGL(dpm_err_kernel, dim3(chunks < 1 ? 1 : (chunks > 256 ? 256 : chunks), N), dim3(256), 0, (hipStream_t)stream, hi, lo, prev,
                     atol, rtol, per_sample, out);
  return mmd_check_launch("dpm_err");
}

// ----------------------------------------------------------------------------- super-resolution model input
// ImageSuperResModel.forward (image_unet.py:704-715): out[n, 0:C] = x[n], out[n, C:2C] = F.interpolate(low_res[n], (H, W),
// mode="bilinear") (align_corners=False: src = (dst + 0.5) * in/out - 0.5 clamped at 0, neighbours clamped at the edge).
__global__ __launch_bounds__(256) void bilinear_concat_kernel(const float* __restrict__ x, const float* __restrict__ low, float* __restrict__ out,
                                                              int N, int C, int H, int W, int h, int w) {
  const int64_t total = (int64_t)N * 2 * C * H * W;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int xo = (int)(i % W), yo = (int)((i / W) % H);
    const int c = (int)((i / ((int64_t)W * H)) % (2 * C));
    const int64_t n = i / ((int64_t)W * H * 2 * C);
    if (c < C) {
      out[i] = x[((n * C + c) * H + yo) * (int64_t)W + xo];
    } else {
      const float fy = fmaxf(((float)yo + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf(((float)xo + 0.5f) * sw - 0.5f, 0.f);
      const int y0 = (int)fy, x0 = (int)fx;
      const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
      const float ly = fy - (float)y0, lx = fx - (float)x0;
      const float* p = low + (n * C + (c - C)) * (int64_t)h * w;
      out[i] = (1.f - ly) * ((1.f - lx) * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * ((1.f - lx) * p[y1 * w + x0] + lx * p[y1 * w + x1]);
    }
  }
}
extern "C" int mmd_bilinear_concat(const float* x, const float* low, float* out, int N, int C, int H, int W, int h, int w, void* stream) {
  MMD_REQUIRE(x && low && out && N > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0, "bilinear_concat: bad argument");
  hipLaunchKernelGGL(bilinear_concat_kernel, dim3(ew_grid((int64_t)N * 2 * C * H * W)), dim3(256), 0, (hipStream_t)stream, x, low, out, N, C, H,
                     W, h, w);
  return mmd_check_launch("bilinear_concat");
}

// The same input as channels-last ROWS for the implicit-GEMM stem: rows[(n, y, x), 0:C] = x, [C:2C] = bilinear(low), [2C:Cpad] = 0.
// The direct stem kernel spent 3.3 ms per evaluation on the 16 x 256 x 256 frames of a clip (6 -> 192 channels); as a K = 9 * 8
// GEMM on rows the stem is one pass of output-write bandwidth.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_concat_rows_kernel(const float* __restrict__ x, const float* __restrict__ low, char* __restrict__ out,
                                                                   int N, int C, int H, int W, int h, int w, int Cpad) {
  const int64_t rows = (int64_t)N * H * W;
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < rows; m += (int64_t)gridDim.x * 256) {
    const int xo = (int)(m % W), yo = (int)((m / W) % H);
    const int64_t n = m / ((int64_t)W * H);
    const float fy = fmaxf(((float)yo + 0.5f) * sh - 0.5f, 0.f), fx = fmaxf(((float)xo + 0.5f) * sw - 0.5f, 0.f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    for (int c = 0; c < Cpad; ++c) {
      float v = 0.f;
      if (c < C) {
        v = x[((n * C + c) * H + yo) * (int64_t)W + xo];
      } else if (c < 2 * C) {
        const float* p = low + (n * C + (c - C)) * (int64_t)h * w;
        v = (1.f - ly) * ((1.f - lx) * p[y0 * w + x0] + lx * p[y0 * w + x1]) + ly * ((1.f - lx) * p[y1 * w + x0] + lx * p[y1 * w + x1]);
      }
      Elt<T>::st(out, m * Cpad + c, v);
    }
  }
}
extern "C" int mmd_bilinear_concat_rows(int dtype, const float* x, const float* low, void* out, int N, int C, int H, int W, int h, int w,
                                        int Cpad, void* stream) {
  MMD_REQUIRE(x && low && out && N > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0 && Cpad >= 2 * C, "bilinear_concat_rows: bad argument");
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "bilinear_concat_rows: bad dtype");
  const dim3 grid(ew_grid((int64_t)N * H * W));
  if (dtype == MMD_BF16) hipLaunchKernelGGL(bilinear_concat_rows_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, x, low, (char*)out, N, C, H, W, h, w, Cpad);
  else hipLaunchKernelGGL(bilinear_concat_rows_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, low, (char*)out, N, C, H, W, h, w, Cpad);
  return mmd_check_launch("bilinear_concat_rows");
}

// ----------------------------------------------------------------------------- training-loss gradient (learned-range variance)
// Gradient of  sum_n ( dmse[n] * mse[n] + dvb[n] * vb[n] )  w.r.t. the model output [N, F, Cm, HW] (Cm = 2C with flag 4):
//   mean channels c < C     : dmse[n] * 2 (o - target) / per                      (the vb term sees the mean DETACHED, gd:1147-1151)
//   variance channels c >= C: dvb[n] * vb_scale / (per ln 2) * d term / d logvar * (max_log - min_log) / 2
// with term = KL(q || p) for t > 0 and the discretized-Gaussian decoder NLL at t == 0 (losses.py:12-77), exactly the forward
// arithmetic of loss_terms_kernel.
__global__ __launch_bounds__(256) void loss_terms_bwd_kernel(const LossParams p, const float* __restrict__ dmse, const float* __restrict__ dvb,
                                                             float vb_scale, float* __restrict__ g) {
  const int64_t per = (int64_t)p.F * p.C * p.HW;
  const int64_t total = per * p.N;
  const int Cm = (p.flags & 4) ? 2 * p.C : p.C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / per, r = i % per;
    const int hw = (int)(r % p.HW), c = (int)((r / p.HW) % p.C);
    const int64_t f = r / ((int64_t)p.HW * p.C);
    const int ti = (int)p.t[n];
    const int64_t mbase = ((n * (int64_t)p.F + f) * Cm) * (int64_t)p.HW + hw;
    const float o = p.mo[mbase + (int64_t)c * p.HW];
    g[mbase + (int64_t)c * p.HW] = dmse[n] * 2.f * (o - p.target[i]) / (float)per;
    if (p.flags & 4) {
      const float cr = p.tables[ti], crm1 = p.tables[p.T + ti], c1 = p.tables[2 * p.T + ti], c2 = p.tables[3 * p.T + ti];
      const float min_log = p.tables[5 * p.T + ti], max_log = p.tables[6 * p.T + ti];
      const float vv = p.mo[mbase + (int64_t)(c + p.C) * p.HW];
      const float frac = (vv + 1.f) / 2.f;
      const float logvar = frac * max_log + (1.f - frac) * min_log;
      const float xv = p.xt[i], x0 = p.x0[i];
      const float px0 = (p.flags & 2) ? o : cr * xv - crm1 * o;
      const float mean = c1 * px0 + c2 * xv;
      float dterm;                                   // d term / d logvar
      if (ti == 0) {
        const float cx = x0 - mean, inv = expf(-0.5f * logvar);
        const float up = inv * (cx + 1.f / 255.f), um = inv * (cx - 1.f / 255.f);
        const float cdf_p = approx_std_normal_cdf(up), cdf_m = approx_std_normal_cdf(um);
        // d cdf(u) / d logvar = pdf~(u) * (-u / 2),  pdf~ = derivative of the tanh approximation
        auto dcdf = [](float u) {
          const float k = 0.7978845608028654f, a = 0.044715f;
          const float th_ = tanhf(k * (u + a * u * u * u));
          return 0.5f * (1.f - th_ * th_) * k * (1.f + 3.f * a * u * u) * (-0.5f * u);
        };
        const float dp = dcdf(up), dm_ = dcdf(um);
        float dlog;
        if (x0 < -0.999f) dlog = cdf_p > 1e-12f ? dp / cdf_p : 0.f;
        else if (x0 > 0.999f) dlog = (1.f - cdf_m) > 1e-12f ? -dm_ / (1.f - cdf_m) : 0.f;
        else dlog = (cdf_p - cdf_m) > 1e-12f ? (dp - dm_) / (cdf_p - cdf_m) : 0.f;
        dterm = -dlog;
      } else {
        const float dm = (c1 * x0 + c2 * xv) - mean;
        dterm = 0.5f * (1.f - expf(min_log - logvar) - dm * dm * expf(-logvar));
      }
      g[mbase + (int64_t)(c + p.C) * p.HW] = dvb[n] * vb_scale / ((float)per * 0.6931471805599453f) * dterm * 0.5f * (max_log - min_log);
    }
  }
}
extern "C" int mmd_loss_terms_bwd(const float* x0, const float* xt, const float* model_out, const float* target, const float* tables,
                                  const int64_t* t, int T, int N, int F, int C, int HW, int flags, float vb_scale, const float* dmse,
                                  const float* dvb, float* g_model_out, void* stream) {
  MMD_REQUIRE(model_out && target && tables && t && dmse && g_model_out && T > 0 && N > 0 && F > 0 && C > 0 && HW > 0, "loss_terms_bwd: bad argument");
  MMD_REQUIRE(!(flags & 4) || (x0 && xt && dvb), "loss_terms_bwd: the vb term needs x0, x_t and dvb");
  LossParams p;
  p.x0 = x0; p.xt = xt; p.mo = model_out; p.target = target; p.tables = tables; p.t = t; p.partial = nullptr;
  p.T = T; p.N = N; p.F = F; p.C = C; p.HW = HW; p.flags = flags; p.nchunk = 0;
  hipLaunchKernelGGL(loss_terms_bwd_kernel, dim3(ew_grid((int64_t)N * F * C * HW)), dim3(256), 0, (hipStream_t)stream, p, dmse, dvb, vb_scale,
                     g_model_out);
  return mmd_check_launch("loss_terms_bwd");
}
