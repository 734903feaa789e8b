// Implicit-GEMM convolution on the gfx950 matrix cores.
//
// Replaces, for channels-last activations X[rows, Cin] (row stride lda):
//   * VideoConv '2d+1d' spatial 3x3 / temporal k=3   (reference multimodal_unet.py:83-99)
//   * VideoConv '3d' k=1 (ResBlock out conv, skip, cross-attn proj, unet:378,401,609)
//   * AudioConv k=3 dilated / k=1                     (unet:108-131)
//   * every qkv / proj_out 1x1 Conv1d of the attention blocks (unet:272,275,605-606)
// as  Y[m, co] = bias[co] + sum_tap sum_ci X[src(m,tap), ci] * W[co, tap*Cin + ci]  (+ R[m, co])
// where src(m,tap) offsets the row position (p0,p1,p2) of m by the tap and zero-pads out of range.
//
// Mapping: 256 threads = 4 waves in a 2(co) x 2(m) grid, block tile BM x BN, K step = 128 bytes per
// row (64 bf16 / 32 fp32).  Operands are staged global -> VGPR -> LDS (144-byte padded rows: every
// 16-lane ds_read_b128 group is conflict free) with register prefetch of the next K step while the
// current one feeds v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (exact fp32).
// The weights are the MFMA A operand (D rows = co) so each lane owns 4 consecutive output channels;
// the accumulators are staged through LDS in fp32 and written with 16-byte coalesced row stores with
// bias and residual folded in.
#include "mmd_common.h"

struct ConvGemmParams {
  const char* A; int64_t lda;
  const char* W;
  const float* bias;
  const char* R; int64_t ldr;
  char* Y; int64_t ldy;
  int M, Cout, Cin, ntaps;
  int D0, D1, D2;
  // optional GroupNorm(+SiLU) fused into the A loader (1x1 convs only): x' = act(x * a[s(m)] + b[s(m)])
  const float* gn_a; const float* gn_b;      // [S, Cin]; contiguous slices of gn_rows rows (>= BM), Cin <= 256
  int gn_act, gn_S;
  int64_t gn_rows;
  // optional GroupNorm statistics of the OUTPUT for its consumer (mmd_gn_finalize_stats): per (64-row record, column) the sum and
  // sum of squares of the stored values, stats[(m / 64) * stats_ld + column] = float2; M % 64 == 0
  float* stats; int64_t stats_ld;
  int taps[27 * 3];
};

// ---- producer-side GroupNorm statistics.  The epilogue threads (column group cg = tid % (BN/8), row phase rr = tid / (BN/8)) hold
// per-record sums over their rows; lanes of a wave that share cg are folded with xor-shuffles, the four waves through `sP`
// (4 x BM/64 x BN float2; aliases the fp32 staging tile, the caller has a barrier in front), and BM/64 x BN threads write one
// float2 each.  Fixed order: deterministic.
template <int BM, int BN>
__device__ __forceinline__ void epilogue_stats(const ConvGemmParams& p, float (&sum)[BM / 64][8], float (&sq)[BM / 64][8], float* sP, int m0,
                                               int n0, int tid) {
  constexpr int CVN = BN / 8, NREC = BM / 64;
  const int lane = tid & 63, wave = tid >> 6, cg = tid % CVN;
#pragma unroll
  for (int r = 0; r < NREC; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int o = CVN; o < 64; o <<= 1) {
        sum[r][j] += __shfl_xor(sum[r][j], o, 64);
        sq[r][j] += __shfl_xor(sq[r][j], o, 64);
      }
    }
  if (lane < CVN) {
#pragma unroll
    for (int r = 0; r < NREC; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float* d = sP + (((wave * NREC + r) * BN) + cg * 8 + j) * 2;
        d[0] = sum[r][j];
        d[1] = sq[r][j];
      }
  }
  __syncthreads();
  float* sQ = sP + 4 * NREC * BN * 2;                   // the per-(record, column) totals, then folded into QUADS of 4 columns
  for (int t = tid; t < NREC * BN; t += 256) {
    const int r = t / BN, col = t % BN;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += sP[(((w * NREC + r) * BN) + col) * 2];
      b += sP[(((w * NREC + r) * BN) + col) * 2 + 1];
    }
    sQ[t * 2] = a;
    sQ[t * 2 + 1] = b;
  }
  __syncthreads();
  const int ncol = p.Cout - n0 < BN ? p.Cout - n0 : BN, nq = ncol >> 2;
  for (int t = tid; t < NREC * nq; t += 256) {
    const int r = t / nq, q = t - r * nq;
    if (m0 + r * 64 < p.M) {
      const float* v = sQ + (r * BN + 4 * q) * 2;
      const float a = (v[0] + v[2]) + (v[4] + v[6]), b = (v[1] + v[3]) + (v[5] + v[7]);
      float* d = p.stats + ((int64_t)(m0 / 64 + r) * p.stats_ld + (n0 >> 2) + q) * 2;      // quad record of this 64-row record
      d[0] = a;
      d[1] = b;
    }
  }
}

#define ROWB 144   // LDS bytes per staged operand row

// Fragment reads of a K step: hipcc places each ds_read_b128 right in front of the MFMA that consumes it and waits lgkmcnt(0) - eight
// exposed LDS round trips per K step.  With the 16 reads of a step issued first
// and pinned there, every MFMA waits with a counted lgkmcnt and the LDS latency of read n + 1 hides under MFMA n.
#define FRAG_FENCE() __builtin_amdgcn_sched_barrier(0)
// A register that holds a LOAD result, used under a (divergent or uniform) branch, is 'maybe still pending' on the path that skips the use:
// hipcc then waits vmcnt(0) again in front of its next use - behind stores issued meanwhile that means waiting for their acknowledgement
// (an epilogue of 8 passes = 8 serialised store round trips).  Passing the registers through an empty asm once, after the wait, makes
// them plain values.
#define LAUNDER4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))

// Padding taps / out-of-range rows read this zero page instead of branching around the load: every thread then issues a
// STATIC number of global loads per K step, so the compiler can keep the newer register stage in flight with a counted
// s_waitcnt vmcnt(N) (a load under a divergent branch forces vmcnt(0) and serialises the pipeline).
__device__ __attribute__((aligned(16))) uint32_t g_zero_page[8] = {0, 0, 0, 0, 0, 0, 0, 0};
__device__ __attribute__((aligned(16))) uint32_t g_zero_row[32] = {};      // 128 bytes: one K plane of a padding row (strip kernel)


template <typename T> struct Mma;
template <> struct Mma<__bf16> {
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // 16 bytes = 4 consecutive k per half-wave; MFMA j pairs k=j (lanes 0-31) with k=4+j (lanes 32-63).
  __device__ static __forceinline__ void run(const u32x4& a, const u32x4& b, f32x16& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
  }
};

// Epilogue shared by both main loops: accumulators -> LDS (fp32, [m][co]) -> bias + residual -> 16-byte coalesced row stores.
// Caller guarantees every wave is past its last operand read (sC aliases the operand buffers).
template <typename T, int BM, int BN>
__device__ __forceinline__ void gemm_epilogue(const ConvGemmParams& p, f32x16 (&acc)[BN / 64][BM / 64], float* sC, int m0, int n0,
                                              int tid, int wc, int wr, int half, int l31) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  constexpr int TCO = BN / 64, TMM = BM / 64;
  constexpr int LDC = BN + 4;
#pragma unroll
  for (int a = 0; a < TCO; ++a)
#pragma unroll
    for (int b = 0; b < TMM; ++b) {
      const int ml = wr * (BM / 2) + b * 32 + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = wc * (BN / 2) + a * 32 + 8 * q + 4 * half;
        f32x4 v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
        *(f32x4*)(sC + ml * LDC + col) = v;
      }
    }
  constexpr int CVN = BN / 8;          // 8-channel groups per row
  constexpr int RP = 256 / CVN;        // rows per pass
  constexpr int NREC = BM / 64;
  const int cg = tid % CVN, rr = tid / CVN;
  const int co = n0 + cg * 8;
  // bf16: the residual rows of every pass are requested here, in front of the barrier (the pass loop used to load a row, wait, add,
  // store: one serial HBM round trip per pass, and every wait also sat on the previous pass's store)
  constexpr bool PRE = EPV == 8;
  constexpr int PPR = 64 / RP;
  u32x4 rres[PRE ? NREC * PPR : 1];
  if (PRE && p.R) {
#pragma unroll
    for (int i = 0; i < NREC * PPR; ++i) {
      const int m = m0 + (i / PPR) * 64 + rr + (i % PPR) * RP;
      rres[i] = *(const u32x4*)((m < p.M && co < p.Cout) ? p.R + ((int64_t)m * p.ldr + co) * ES : (const char*)g_zero_page);
    }
  }
  __syncthreads();
  if (PRE && p.R) {
#pragma unroll
    for (int i = 0; i < NREC * PPR; ++i) LAUNDER4(rres[i]);
  }
  float ssum[NREC][8], ssq[NREC][8];
#pragma unroll
  for (int r = 0; r < NREC; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) ssum[r][j] = ssq[r][j] = 0.f;
  if (co < p.Cout) {
    float bs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[j] = p.bias ? p.bias[co + j] : 0.f;
#pragma unroll
    for (int rec = 0; rec < NREC; ++rec) {
#pragma unroll
      for (int pp = 0; pp < PPR; ++pp) {
        const int ml = rec * 64 + rr + pp * RP;
        const int m = m0 + ml;
        if (m >= p.M) break;
        float v[8];
        const f32x4 c0 = *(const f32x4*)(sC + ml * LDC + cg * 8);
        const f32x4 c1 = *(const f32x4*)(sC + ml * LDC + cg * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = c0[j] + bs[j]; v[4 + j] = c1[j] + bs[4 + j]; }
        if (p.R) {
#pragma unroll
          for (int h = 0; h < 8 / EPV; ++h) {
            float rf[EPV];
            if constexpr (PRE) Elt<T>::unpack(rres[rec * PPR + pp], rf);
            else Elt<T>::unpack(*(const u32x4*)(p.R + ((int64_t)m * p.ldr + co + h * EPV) * ES), rf);
#pragma unroll
            for (int j = 0; j < EPV; ++j) v[h * EPV + j] += rf[j];
          }
        }
#pragma unroll
        for (int h = 0; h < 8 / EPV; ++h) {
          const u32x4 pk = Elt<T>::pack(v + h * EPV);
          *(u32x4*)(p.Y + ((int64_t)m * p.ldy + co + h * EPV) * ES) = pk;
          if (p.stats) {     // statistics of the values as STORED (what the consumer GroupNorm reads back)
            float rf[EPV];
            Elt<T>::unpack(pk, rf);
#pragma unroll
            for (int j = 0; j < EPV; ++j) { ssum[rec][h * EPV + j] += rf[j]; ssq[rec][h * EPV + j] += rf[j] * rf[j]; }
          }
        }
      }
    }
  }
  if (p.stats) {             // block-uniform
    __syncthreads();                     // every thread is past its last sC read: the wave partials alias the staging tile
    epilogue_stats<BM, BN>(p, ssum, ssq, sC, m0, n0, tid);
  }
}

template <typename T, int BM, int BN, bool GN>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ConvGemmParams p) {
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;            // element size in bytes
  constexpr int AR = BM / 32, WR = BN / 32;   // rows staged per thread
  constexpr int TCO = BN / 64, TMM = BM / 64; // 32x32 tiles per wave
  constexpr int LDC = BN + 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                          // [2][BM][ROWB]
  char* sW = smem + 2 * BM * ROWB;          // [2][BN][ROWB]
  float* sC = (float*)smem;                 // [BM][LDC] (epilogue, aliases the operand buffers)
  // tap table lives behind the operand/epilogue region (all LDS in ONE dynamic array: keeps the base 16-B aligned)
  constexpr int MAIN_B = (2 * (BM + BN) * ROWB > BM * LDC * 4) ? 2 * (BM + BN) * ROWB : BM * LDC * 4;
  int* s_taps = (int*)(smem + MAIN_B);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wc = wave & 1, wr = wave >> 1;
  const int half = lane >> 5, l31 = lane & 31;

  if (tid < p.ntaps * 3) s_taps[tid] = p.taps[tid];

  // XCD-aware tile order: consecutive tiles (sharing activation halos / weight panels) stay on one L2
  const int Nt = (p.Cout + BN - 1) / BN;
  const int nwg = gridDim.x;
  int wgid;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = wgid % Nt, mt = wgid / Nt;
  const int m0 = mt * BM, n0 = nt * BN;

  const int CinV = p.Cin / EPV;
  const int KV = CinV * p.ntaps;
  const int64_t K = (int64_t)p.Cin * p.ntaps;
  const int nit = (KV + 7) >> 3;
  const int cv = tid & 7, r0 = tid >> 3;
  const int D12 = p.D1 * p.D2;

  int pp0[AR], pp1[AR], pp2[AR];
  int64_t arow[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + r0 + 32 * i;
    arow[i] = (int64_t)m;
    if (m < p.M) {
      pp2[i] = m % p.D2;
      pp1[i] = (m / p.D2) % p.D1;
      pp0[i] = (m / D12) % p.D0;
    } else {
      pp0[i] = pp1[i] = pp2[i] = -(1 << 28);
    }
  }
  int tap = cv / CinV, civ = cv % CinV;
  int kv = cv;
  // fused GroupNorm: the block's rows touch at most two slices (gn_rows >= BM); their affine rows are cached in LDS
  float* sGN = (float*)(smem + MAIN_B + 336);        // [2 slices][a|b][Cin]
  int gsel[AR];
  if (GN) {
    const int s0 = (int)(m0 / p.gn_rows);
    for (int i = tid; i < 4 * p.Cin; i += 256) {
      const int sl = i / (2 * p.Cin), ab = (i / p.Cin) & 1, c = i % p.Cin;
      const int sidx = min(s0 + sl, p.gn_S - 1);
      sGN[i] = (ab ? p.gn_b : p.gn_a)[(int64_t)sidx * p.Cin + c];
    }
#pragma unroll
    for (int i = 0; i < AR; ++i) gsel[i] = min((int)(arow[i] / p.gn_rows) - s0, 1) * 2 * p.Cin;
  }

  u32x4 ra0[AR], rw0[WR], ra1[AR], rw1[WR];   // two register stages: tiles it+1 and it+2 are in flight
  __syncthreads();   // s_taps visible

  auto load_tile = [&](u32x4 (&ra)[AR], u32x4 (&rw)[WR], int& meta) {
    int okmask = 0;
    const bool tapok = tap < p.ntaps;
    int o0 = 0, o1 = 0, o2 = 0;
    if (tapok) { o0 = s_taps[tap * 3]; o1 = s_taps[tap * 3 + 1]; o2 = s_taps[tap * 3 + 2]; }
    const int64_t roff = (int64_t)o0 * D12 + o1 * p.D2 + o2;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const bool ok = tapok && (unsigned)(pp0[i] + o0) < (unsigned)p.D0 &&
                      (unsigned)(pp1[i] + o1) < (unsigned)p.D1 && (unsigned)(pp2[i] + o2) < (unsigned)p.D2;
      const char* src = ok ? p.A + ((arow[i] + roff) * p.lda + (int64_t)civ * EPV) * ES : (const char*)g_zero_page;
      u32x4 v = *(const u32x4*)src;
      okmask |= (ok ? 1 : 0) << i;
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      const int co = n0 + r0 + 32 * i;
      const char* src = (co < p.Cout && kv < KV) ? p.W + ((int64_t)co * K + (int64_t)kv * EPV) * ES : (const char*)g_zero_page;
      rw[i] = *(const u32x4*)src;
    }
    meta = okmask | (civ << 8);
  };
  auto store_tile = [&](int buf, const u32x4 (&ra)[AR], const u32x4 (&rw)[WR], int meta) {
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      u32x4 v = ra[i];
      if (GN && ((meta >> i) & 1)) {       // GroupNorm(+FiLM)(+SiLU) on the way into LDS; padding rows stay zero
        float f[EPV];
        Elt<T>::unpack(v, f);
        const float* ap = sGN + gsel[i] + (meta >> 8) * EPV;
        const float* bp = ap + p.Cin;
#pragma unroll
        for (int e = 0; e < EPV; e += 4) {
          const f32x4 av = *(const f32x4*)(ap + e), bv = *(const f32x4*)(bp + e);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float y = f[e + k] * av[k] + bv[k];
            f[e + k] = p.gn_act ? silu_f(y) : y;
          }
        }
        v = Elt<T>::pack(f);
      }
      *(u32x4*)(sA + buf * BM * ROWB + (r0 + 32 * i) * ROWB + cv * 16) = v;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i)
      *(u32x4*)(sW + buf * BN * ROWB + (r0 + 32 * i) * ROWB + cv * 16) = rw[i];
  };
  auto advance = [&]() {
    kv += 8;
    civ += 8;
    while (civ >= CinV) { civ -= CinV; ++tap; }
  };

  f32x16 acc[TCO][TMM];
#pragma unroll
  for (int a = 0; a < TCO; ++a)
#pragma unroll
    for (int b = 0; b < TMM; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto compute = [&](int cur) {
    const char* bW = sW + cur * BN * ROWB + (wc * (BN / 2) + l31) * ROWB + half * 16;
    const char* bA = sA + cur * BM * ROWB + (wr * (BM / 2) + l31) * ROWB + half * 16;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      u32x4 fw[TCO], fa[TMM];
#pragma unroll
      for (int a = 0; a < TCO; ++a) fw[a] = *(const u32x4*)(bW + a * 32 * ROWB + c * 32);
#pragma unroll
      for (int b = 0; b < TMM; ++b) fa[b] = *(const u32x4*)(bA + b * 32 * ROWB + c * 32);
#pragma unroll
      for (int a = 0; a < TCO; ++a)
#pragma unroll
        for (int b = 0; b < TMM; ++b) Mma<T>::run(fw[a], fa[b], acc[a][b]);
    }
  };

  // 3-stage pipeline: LDS[cur] = tile it, one register set = tile it+1 (landing), the other = tile it+2 (issued now)
  int meta0 = 0, meta1 = 0;
  load_tile(ra0, rw0, meta0);
  if (GN) __syncthreads();          // sGN visible
  store_tile(0, ra0, rw0, meta0);
  if (nit > 1) { advance(); load_tile(ra0, rw0, meta0); }
  __syncthreads();
  int cur = 0, it = 0;
  while (true) {
    if (it + 2 < nit) { advance(); load_tile(ra1, rw1, meta1); }
    compute(cur);
    if (it + 1 < nit) store_tile(cur ^ 1, ra0, rw0, meta0);
    __syncthreads();
    cur ^= 1;
    if (++it >= nit) break;
    if (it + 2 < nit) { advance(); load_tile(ra0, rw0, meta0); }
    compute(cur);
    if (it + 1 < nit) store_tile(cur ^ 1, ra1, rw1, meta1);
    __syncthreads();
    cur ^= 1;
    if (++it >= nit) break;
  }

  gemm_epilogue<T, BM, BN>(p, acc, sC, m0, n0, tid, wc, wr, half, l31);
}

// ---------------------------------------------------------------------------------------------------------------------
// Direct-to-LDS main loop (global_load_lds_dwordx4): operands never pass through VGPRs, no ds_write pass.
// LDS tile rows are 128 B unpadded; the 16-byte chunk index is XOR-swizzled with ((row >> 1) & 7) so the
// ds_read_b128 fragment reads stay conflict free.  The DMA writes lane-linear (base + lane*16), so the swizzle is
// applied to the per-lane SOURCE address: lane L of the instruction covering tile rows [8j, 8j+8) lands in row
// 8j + L/8, physical chunk L%8, and therefore fetches logical chunk (L%8) ^ ((row >> 1) & 7).
// FAST (Cin a multiple of one K step = 128 bytes per row): every K step lies inside ONE tap, so the tap offset and
// the channel offset are wave-uniform scalars and a DMA source is  lane-constant row pointer + uniform offset;
// the per-step address arithmetic drops from ~250 to ~60 instructions (it was 1.5x the MFMA issue time).
// NS = LDS ring depth.  NS = 2 (tile 129): the next K step's DMA under this step's MFMAs, two resident blocks per CU cover each
// other's round trips.  NS = 4 (tile 132, FAST only): launches with FEWER TILES THAN THE CHIP HAS BLOCK SLOTS (the ds8 level: M = 4096
// -> 128 tiles) have nothing co-resident to overlap with and pay one L2 round trip per K step (3x3 512 -> 512 at ds8: 72 - 144 steps,
// 73 - 138 us at 260 - 280 TFLOP/s); there the block keeps THREE K steps of DMA in flight over counted s_waitcnt vmcnt(16 / 8 / 0) and
// one raw s_barrier per step (no fence: __syncthreads() would drain the queue).  Same K order, same epilogue: bitwise equal to the
// other tiled loops, so the choice between 129 and 132 is free (ops.conv_gemm: by tile count).
// DESC (FAST only, round 3): the DMA instructions go through wave-uniform buffer descriptors - a lane-constant 32-bit row offset
// that is recomputed only when the tap changes (padding rows: an offset beyond the descriptor, the DMA then writes zeros), and a
// SCALAR offset for the K step - instead of per-step 64-bit pointers with zero-page selects: ~150 -> ~50 non-MFMA instructions per
// step (the tiled loops are issue-bound, see tile 133).  Needs the activation view below 2 GB.
template <typename T, bool FAST, int NS, bool DESC>
__global__ __launch_bounds__(256, (NS > 2 ? 1 : 2)) void conv_gemm_glds_kernel(const ConvGemmParams p) {
  static_assert(!DESC || FAST, "descriptor addressing: uniform-tap K steps only");
  constexpr int BM = 128, BN = 128;
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  constexpr int LDC = BN + 4;
  constexpr int TILE_B = 128 * 128;
  constexpr int MAIN_B = (2 * NS * TILE_B > BM * LDC * 4) ? 2 * NS * TILE_B : BM * LDC * 4;
  static_assert(NS == 2 || (NS == 4 && FAST), "ring depths: 2 (any layer) or 4 (uniform-tap addressing only)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                       // [NS][128 rows][128 B]
  char* sW = smem + NS * TILE_B;
  float* sC = (float*)smem;
  int* s_taps = (int*)(smem + MAIN_B);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wr = wave >> 1;
  const int half = lane >> 5, l31 = lane & 31;
  if (tid < p.ntaps * 3) s_taps[tid] = p.taps[tid];

  const int Nt = (p.Cout + BN - 1) / BN;
  const int nwg = gridDim.x;
  int wgid;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = wgid % Nt, mt = wgid / Nt;
  const int m0 = mt * BM, n0 = nt * BN;

  const int CinV = p.Cin / EPV;
  const int KV = CinV * p.ntaps;
  const int64_t K = (int64_t)p.Cin * p.ntaps;
  const int nit = (KV + 7) >> 3;
  const int D12 = p.D1 * p.D2;

  const int lrow = lane >> 3, pc = lane & 7;
  const int c_par[2] = {pc ^ (lane >> 4), pc ^ (lane >> 4) ^ 4};    // logical chunk for even / odd row groups
  int tapP[2], civP[2], kvP[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) { tapP[q] = c_par[q] / CinV; civP[q] = c_par[q] % CinV; kvP[q] = c_par[q]; }

  int pp0[4], pp1[4], pp2[4];
  int64_t arow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wave * 32 + 8 * i + lrow;
    arow[i] = (int64_t)m;
    if (m < p.M) {
      pp2[i] = m % p.D2;
      pp1[i] = (m / p.D2) % p.D1;
      pp0[i] = (m / D12) % p.D0;
    } else {
      pp0[i] = pp1[i] = pp2[i] = -(1 << 28);
    }
  }
  // FAST path state: lane-constant pointers, uniform (tap, channel-chunk) cursor
  const char* a_ptr[4];
  const char* w_ptr[4];
  bool w_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_ptr[i] = p.A + (arow[i] * p.lda + (int64_t)c_par[i & 1] * EPV) * ES;
    const int co = n0 + wave * 32 + 8 * i + lrow;
    w_ok[i] = co < p.Cout;
    w_ptr[i] = p.W + ((int64_t)(w_ok[i] ? co : 0) * K + (int64_t)c_par[i & 1] * EPV) * ES;
  }
  int u_tap = 0, u_civ = 0;                  // uniform: tap index and first 16-byte chunk of this K step inside the tap
  __syncthreads();   // s_taps visible

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // DESC state: descriptors over the activation view / this block's weight rows, lane-constant offsets
  const uint32_t OOB = 0xfffffff0u;
  const int w_rows = p.Cout - n0 < BN ? p.Cout - n0 : BN;
  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)(((int64_t)p.M - 1) * p.lda * ES + (int64_t)p.Cin * ES), 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * K * ES), 0, (int)((int64_t)w_rows * K * ES), 0x00020000);
  uint32_t a_base[4], w_off[4], a_voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_base[i] = (uint32_t)((arow[i] * p.lda + (int64_t)c_par[i & 1] * EPV) * ES);
    const int row = wave * 32 + 8 * i + lrow;
    w_off[i] = row < w_rows ? (uint32_t)((row * K + (int64_t)c_par[i & 1] * EPV) * ES) : OOB;
    a_voff[i] = OOB;
  }
  auto issue = [&](int buf) {
    if constexpr (DESC) {
      if (u_civ == 0) {                                   // a new tap (uniform): row validity and row shift of this tap
        const int t3 = u_tap * 3;
        const int o0 = __builtin_amdgcn_readfirstlane(s_taps[t3]), o1 = __builtin_amdgcn_readfirstlane(s_taps[t3 + 1]),
                  o2 = __builtin_amdgcn_readfirstlane(s_taps[t3 + 2]);
        const int shift = (int)((((int64_t)o0 * D12 + (int64_t)o1 * p.D2 + o2) * p.lda) * ES);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool ok = (unsigned)(pp0[i] + o0) < (unsigned)p.D0 && (unsigned)(pp1[i] + o1) < (unsigned)p.D1 &&
                          (unsigned)(pp2[i] + o2) < (unsigned)p.D2;
          a_voff[i] = ok ? a_base[i] + (uint32_t)shift : OOB;
        }
      }
      const int soffA = u_civ * 16, soffW = (u_tap * p.Cin + u_civ * EPV) * ES;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lptr_t)(sA + buf * TILE_B + (wave * 32 + 8 * i) * 128), 16, a_voff[i], soffA, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lptr_t)(sW + buf * TILE_B + (wave * 32 + 8 * i) * 128), 16, w_off[i], soffW, 0, 0);
      return;
    }
    if (FAST) {
      const int t3 = u_tap * 3;
      const int o0 = __builtin_amdgcn_readfirstlane(s_taps[t3]), o1 = __builtin_amdgcn_readfirstlane(s_taps[t3 + 1]),
                o2 = __builtin_amdgcn_readfirstlane(s_taps[t3 + 2]);
      const int64_t offA = (((int64_t)o0 * D12 + (int64_t)o1 * p.D2 + o2) * p.lda + (int64_t)u_civ * EPV) * ES;
      const int64_t offW = ((int64_t)u_tap * p.Cin + (int64_t)u_civ * EPV) * ES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = (unsigned)(pp0[i] + o0) < (unsigned)p.D0 && (unsigned)(pp1[i] + o1) < (unsigned)p.D1 &&
                        (unsigned)(pp2[i] + o2) < (unsigned)p.D2;
        const char* src = ok ? a_ptr[i] + offA : (const char*)g_zero_page;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + buf * TILE_B + (wave * 32 + 8 * i) * 128), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* src = w_ok[i] ? w_ptr[i] + offW : (const char*)g_zero_page;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sW + buf * TILE_B + (wave * 32 + 8 * i) * 128), 16, 0, 0);
      }
      return;
    }
    int o0[2], o1[2], o2[2];
    bool tapok[2];
    int64_t roff[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      tapok[q] = tapP[q] < p.ntaps;
      const int t3 = tapok[q] ? tapP[q] * 3 : 0;
      o0[q] = s_taps[t3]; o1[q] = s_taps[t3 + 1]; o2[q] = s_taps[t3 + 2];
      roff[q] = (int64_t)o0[q] * D12 + o1[q] * p.D2 + o2[q];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i & 1;
      const bool ok = tapok[q] && (unsigned)(pp0[i] + o0[q]) < (unsigned)p.D0 &&
                      (unsigned)(pp1[i] + o1[q]) < (unsigned)p.D1 && (unsigned)(pp2[i] + o2[q]) < (unsigned)p.D2;
      const char* src = ok ? p.A + ((arow[i] + roff[q]) * p.lda + (int64_t)civP[q] * EPV) * ES : (const char*)g_zero_page;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + buf * TILE_B + (wave * 32 + 8 * i) * 128), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = i & 1;
      const int co = n0 + wave * 32 + 8 * i + lrow;
      const char* src = (co < p.Cout && kvP[q] < KV) ? p.W + ((int64_t)co * K + (int64_t)kvP[q] * EPV) * ES : (const char*)g_zero_page;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sW + buf * TILE_B + (wave * 32 + 8 * i) * 128), 16, 0, 0);
    }
  };
  auto advance = [&]() {
    if (FAST) {
      u_civ += 8;
      if (u_civ >= CinV) { u_civ = 0; ++u_tap; }
      return;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      kvP[q] += 8;
      civP[q] += 8;
      while (civP[q] >= CinV) { civP[q] -= CinV; ++tapP[q]; }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int xsw = (l31 >> 1) & 7;
  auto compute = [&](int buf) {
    const char* bW = sW + buf * TILE_B + (wc * 64 + l31) * 128;
    const char* bA = sA + buf * TILE_B + (wr * 64 + l31) * 128;
    u32x4 fw[4][2], fa[4][2];
    auto rd = [&](int c) {
      const int phys = ((2 * c + half) ^ xsw) * 16;
#pragma unroll
      for (int a = 0; a < 2; ++a) fw[c][a] = *(const u32x4*)(bW + a * 32 * 128 + phys);
#pragma unroll
      for (int b = 0; b < 2; ++b) fa[c][b] = *(const u32x4*)(bA + b * 32 * 128 + phys);
    };
    auto mm = [&](int c) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) Mma<T>::run(fw[c][a], fa[c][b], acc[a][b]);
    };
    rd(0); rd(1);
    FRAG_FENCE();
    rd(2); mm(0);
    FRAG_FENCE();
    rd(3); mm(1);
    FRAG_FENCE();
    mm(2); mm(3);
    FRAG_FENCE();
  };

  // epilogue operands are fetched ahead of time: the bias here, the residual rows under the last K step's MFMAs -
  // a short-K tile (4-12 steps) otherwise pays both round trips serially after its last barrier
  constexpr int CVN = BN / 8, RP = 256 / CVN, NPASS = BM / RP;
  const int e_cg = tid % CVN, e_rr = tid / CVN;
  const int e_co = n0 + e_cg * 8;
  float bs[8];
  u32x4 rres[NPASS][8 / EPV];
  int cur = 0;
  if constexpr (NS == 2) {
    issue(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[j] = (p.bias && e_co < p.Cout) ? p.bias[e_co + j] : 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it + 1 < nit; ++it) {
      advance();
      issue(cur ^ 1);                                      // DMA of the next K step runs under this step's MFMAs
      compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
  } else {
    // ---- deep ring: K steps it+1 .. it+NS-2 stay in flight while step `it` is consumed.  A wave issues 8 DMA instructions per step,
    // in step order, and the vector-memory counter retires in order, so "at most 8 k outstanding" = "everything up to step
    // (last issued - k) has landed" (the two bias loads in front of the first step only make the first waits stricter).
    constexpr int D = NS - 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[j] = (p.bias && e_co < p.Cout) ? p.bias[e_co + j] : 0.f;
    for (int s0 = 0; s0 < D && s0 < nit; ++s0) {
      issue(s0);
      advance();
    }
    for (int it = 0; it + 1 < nit; ++it) {
      const int ahead = nit - 1 - it < D - 1 ? nit - 1 - it : D - 1;       // younger steps that may stay in flight (uniform)
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // step `it` landed for every wave; every wave is past its fragment reads of step it - 1
      asm volatile("" ::: "memory");
      if (it + D < nit) {                    // refill the slot step it - 1 just left
        issue((it + D) % NS);
        advance();
      }
      compute(it % NS);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    cur = (nit - 1) % NS;
  }
  if (p.R) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int m = m0 + e_rr + ps * RP;
      const bool ok = m < p.M && e_co < p.Cout;
#pragma unroll
      for (int h = 0; h < 8 / EPV; ++h)
        rres[ps][h] = *(const u32x4*)(ok ? p.R + ((int64_t)m * p.ldr + e_co + h * EPV) * ES : (const char*)g_zero_page);
    }
  }
  compute(cur);
  __syncthreads();                                       // every wave is past its last operand read: sC may alias
  if (p.R) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
      for (int h = 0; h < 8 / EPV; ++h) LAUNDER4(rres[ps][h]);
  }

#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
      const int ml = wr * 64 + b2 * 32 + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = wc * 64 + a * 32 + 8 * q + 4 * half;
        f32x4 v = {acc[a][b2][4 * q], acc[a][b2][4 * q + 1], acc[a][b2][4 * q + 2], acc[a][b2][4 * q + 3]};
        *(f32x4*)(sC + ml * LDC + col) = v;
      }
    }
  __syncthreads();
  float ssum[2][8], ssq[2][8];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < 8; ++j) ssum[r][j] = ssq[r][j] = 0.f;
  if (e_co < p.Cout) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int ml = e_rr + ps * RP;
      const int m = m0 + ml;
      if (m < p.M) {
        float v[8];
        const f32x4 c0 = *(const f32x4*)(sC + ml * LDC + e_cg * 8);
        const f32x4 c1 = *(const f32x4*)(sC + ml * LDC + e_cg * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = c0[j] + bs[j]; v[4 + j] = c1[j] + bs[4 + j]; }
        if (p.R) {
#pragma unroll
          for (int h = 0; h < 8 / EPV; ++h) {
            float rf[EPV];
            Elt<T>::unpack(rres[ps][h], rf);
#pragma unroll
            for (int j = 0; j < EPV; ++j) v[h * EPV + j] += rf[j];
          }
        }
#pragma unroll
        for (int h = 0; h < 8 / EPV; ++h) {
          const u32x4 pk = Elt<T>::pack(v + h * EPV);
          *(u32x4*)(p.Y + ((int64_t)m * p.ldy + e_co + h * EPV) * ES) = pk;
          if (p.stats) {     // statistics of the values as STORED (what the consumer GroupNorm reads back)
            float rf[EPV];
            Elt<T>::unpack(pk, rf);
#pragma unroll
            for (int j = 0; j < EPV; ++j) { ssum[ps / (NPASS / 2)][h * EPV + j] += rf[j]; ssq[ps / (NPASS / 2)][h * EPV + j] += rf[j] * rf[j]; }
          }
        }
      }
    }
  }
  if (p.stats) {             // block-uniform
    __syncthreads();                     // every thread is past its last sC read: the wave partials alias the staging tile
    epilogue_stats<128, 128>(p, ssum, ssq, sC, m0, n0, tid);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Halo-tile main loop for spatial 3x3 convolutions (tile code 130, EXPERIMENTAL: opt-in, not in the autotune set yet).
// The direct-to-LDS kernel above is bound by the L2 -> LDS fill rate (~27 B/clk/CU measured against the 64 B/clk/CU the
// MFMA rate would need at 64 flop per staged byte), and a 9-tap conv stages every activation row nine times.  Here a block
// owns an 8x16 pixel patch of ONE frame: per 128-byte channel chunk the (8+2)x(16+2) halo of the patch is staged once
// (23 KB) and the nine taps read shifted windows of it, so a chunk stages 23 + 9*16 = 167 KB instead of 288 KB and padding
// is a zero row of the halo (no per-tap predicates).  K order is chunk-major (the other main loops are tap-major: results
// agree to fp32 rounding, not bitwise).  Halo row r = (ph+1+dh)*18 + (pw+1+dw); the 16-byte chunk swizzle is keyed on r,
// and 16 consecutive pixels of a patch row are 16 consecutive halo rows, so the fragment reads stay conflict free.
// Requires: taps with d0 offset 0 and |dh|,|dw| <= 1, D1 % 8 == 0, D2 % 16 == 0, M = D0*D1*D2, Cin % (128 B) == 0.
// GN = true (3x3 convs, 9 taps): GroupNorm(+FiLM)(+SiLU) of the INPUT is applied to the staged halo in LDS, once per channel chunk -
// the normalised tensor never exists in HBM (the gn_apply pass in front of every ResBlock in-conv: unet:339-340,457-458).  The
// affine rows of a chunk (64 channels of a | b of the block's sample) arrive by DMA in a 1 KB LDS ring one chunk ahead; the halo of
// chunk c + 1 is complete after step 5 of chunk c (six DMA pieces per wave, one per step), and steps 6, 7, 8 transform it in place:
// thread (tid, i) owns the 16 bytes at tid * 16 + i * 4096 of the stage - pixel row tid / 8 + 32 i, physical chunk tid % 8, whose
// LOGICAL chunk (tid % 8) ^ ((tid / 16) % 8) does not depend on i, so a thread needs 8 channels of a | b per chunk.  Padding pixels
// (the zero page) stay zero: the conv pads the normalised activation.  Same expressions as gn_apply: the result is bitwise equal to
// gn_apply followed by the plain halo kernel.
template <typename T, bool GN>
__global__ __launch_bounds__(256, 2) void conv_gemm_halo_kernel(const ConvGemmParams p) {
  constexpr int BM = 128, BN = 128, PH = 8, PW = 16, HWD = PW + 2, HR = (PH + 2) * HWD, HG = (HR + 7) / 8, HJ = (HG + 3) / 4;
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  constexpr int LDC = BN + 4;
  constexpr int A_B = HG * 8 * 128;            // halo stage: 23 groups of 8 rows
  constexpr int W_B = 128 * 128;
  constexpr int MAIN_B = (2 * A_B + 2 * W_B > BM * LDC * 4) ? 2 * A_B + 2 * W_B : BM * LDC * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                             // [2][HG*8 rows][128 B]
  char* sW = smem + 2 * A_B;                   // [2][128 rows][128 B]
  float* sC = (float*)smem;
  int* s_taps = (int*)(smem + MAIN_B);
  float* sGN = (float*)(smem + MAIN_B + 336);  // GN: [2 chunk parities][a (64 channels) | b (64 channels)]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wr = wave >> 1;
  const int half = lane >> 5, l31 = lane & 31;
  if (tid < p.ntaps * 3) s_taps[tid] = p.taps[tid];

  const int Nt = (p.Cout + BN - 1) / BN;
  const int nwg = gridDim.x;
  int wgid;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = wgid % Nt, mt = wgid / Nt;
  const int n0 = nt * BN;
  const int TW = p.D2 / PW, tpf = TW * (p.D1 / PH);
  const int d0 = mt / tpf, trem = mt - d0 * tpf;
  const int h0 = (trem / TW) * PH, w0 = (trem % TW) * PW;
  const int64_t mframe = (int64_t)d0 * p.D1 * p.D2;

  const int CinV = p.Cin / EPV;
  const int nchunk = CinV >> 3;
  const int64_t K = (int64_t)p.Cin * p.ntaps;
  const int nit = nchunk * p.ntaps;

  const int lrow = lane >> 3, pc = lane & 7;
  // halo DMA: wave w stages row groups g = w + 4j; lane-constant source (row pointer + logical chunk), uniform chunk offset
  const char* h_ptr[HJ];
  bool h_ok[HJ];
#pragma unroll
  for (int j = 0; j < HJ; ++j) {
    const int g = wave + 4 * j;
    const int r = 8 * g + lrow;
    const int hr = r / HWD, hc = r - hr * HWD;
    const int hh = h0 - 1 + hr, ww = w0 - 1 + hc;
    h_ok[j] = g < HG && r < HR && (unsigned)hh < (unsigned)p.D1 && (unsigned)ww < (unsigned)p.D2;
    const int logical = pc ^ ((r >> 1) & 7);
    h_ptr[j] = p.A + ((mframe + (int64_t)hh * p.D2 + ww) * p.lda + (int64_t)logical * EPV) * ES;
  }
  const char* w_ptr[4];
  bool w_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int logical = pc ^ (((8 * i + lrow) >> 1) & 7);
    const int co = n0 + wave * 32 + 8 * i + lrow;
    w_ok[i] = co < p.Cout;
    w_ptr[i] = p.W + ((int64_t)(w_ok[i] ? co : 0) * K + (int64_t)logical * EPV) * ES;
  }
  __syncthreads();   // s_taps visible

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue_w = [&](int buf, int t, int c) {
    const int64_t offW = ((int64_t)t * p.Cin + (int64_t)c * 8 * EPV) * ES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* src = w_ok[i] ? w_ptr[i] + offW : (const char*)g_zero_page;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sW + buf * W_B + (wave * 32 + 8 * i) * 128), 16, 0, 0);
    }
  };
  auto issue_h = [&](int buf, int c, int j) {
    if (wave + 4 * j < HG) {                                   // wave-uniform
      const char* src = h_ok[j] ? h_ptr[j] + (int64_t)c * 128 : (const char*)g_zero_page;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sA + buf * A_B + (wave + 4 * j) * 1024), 16, 0, 0);
    }
  };
  // ---- fused input GroupNorm (GN): affine ring DMA, in-place transform of a landed halo stage
  constexpr int NSLOT = (HG * 64 + 255) / 256;                 // 16-byte slots of a stage per thread (the last one is partial)
  unsigned gvalid = 0;                                         // bit i: slot i is a pixel inside the frame (else padding: stays zero)
  const float* gn_src = nullptr;                               // waves 0 / 1: this lane's float of the sample's a / b row
  const int glc = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;          // first channel (inside a chunk) of this thread's 16 bytes
  if (GN) {
    const int sidx = min((int)(mframe / p.gn_rows), p.gn_S - 1);
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      const int r = (tid >> 3) + 32 * i;
      const int hr = r / HWD, hc = r - hr * HWD;
      const bool ok = r < HR && (unsigned)(h0 - 1 + hr) < (unsigned)p.D1 && (unsigned)(w0 - 1 + hc) < (unsigned)p.D2;
      gvalid |= (ok ? 1u : 0u) << i;
    }
    gn_src = (wave == 0 ? p.gn_a : p.gn_b) + (int64_t)sidx * p.Cin + lane;
  }
  auto issue_gn = [&](int c) {                                 // 64 floats of a (wave 0) and of b (wave 1) -> ring slot c & 1
    if (wave < 2)                                              // wave-uniform
      __builtin_amdgcn_global_load_lds((gptr_t)(gn_src + c * 64), (lptr_t)(sGN + (c & 1) * 128 + wave * 64), 4, 0, 0);
  };
  static_assert(!GN || EPV == 8, "fused input GroupNorm: bf16 stages only (64 channels per 128-byte chunk)");
  auto transform = [&](int buf, int c, int part) {             // part 0..2: a third of the slots; part < 0: all of them
    constexpr int PER = (NSLOT + 2) / 3;
    const float* ap = sGN + (c & 1) * 128 + glc;
    const f32x4 a0 = *(const f32x4*)ap, a1 = *(const f32x4*)(ap + 4), b0 = *(const f32x4*)(ap + 64), b1 = *(const f32x4*)(ap + 68);
    const float av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      if (part >= 0 && i / PER != part) continue;              // (block-uniform)
      if (i * 256 + 256 <= HG * 64 || tid < HG * 64 - i * 256) {     // the partial last slot: wave-uniform (HG * 64 % 64 == 0)
        char* q = sA + buf * A_B + tid * 16 + i * 4096;
        const u32x4 v = *(const u32x4*)q;
        float f[EPV];
        Elt<T>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          const float w = f[e] * av[e] + bv[e];
          f[e] = p.gn_act ? silu_f(w) : w;
        }
        const u32x4 y = Elt<T>::pack(f);
        *(u32x4*)q = ((gvalid >> i) & 1u) ? y : v;
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

  const int xsw = (l31 >> 1) & 7;
  int rb[2];                                   // halo row of this lane's pixel (tap 0,0) per m sub-tile
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int px = wr * 64 + b * 32 + l31;
    rb[b] = ((px >> 4) + 1) * HWD + (px & 15) + 1;
  }
  auto compute = [&](int bufw, int bufa, int t) {
    const int o1 = __builtin_amdgcn_readfirstlane(s_taps[t * 3 + 1]), o2 = __builtin_amdgcn_readfirstlane(s_taps[t * 3 + 2]);
    const int toff = o1 * HWD + o2;
    const char* bW = sW + bufw * W_B + (wc * 64 + l31) * 128;
    const char* bA[2];
    int key[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int r = rb[b] + toff;
      key[b] = (r >> 1) & 7;
      bA[b] = sA + bufa * A_B + r * 128;
    }
    u32x4 fw[4][2], fa[4][2];
    auto rd = [&](int c) {
#pragma unroll
      for (int a = 0; a < 2; ++a) fw[c][a] = *(const u32x4*)(bW + a * 32 * 128 + (((2 * c + half) ^ xsw) * 16));
#pragma unroll
      for (int b = 0; b < 2; ++b) fa[c][b] = *(const u32x4*)(bA[b] + (((2 * c + half) ^ key[b]) * 16));
    };
    auto mm = [&](int c) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) Mma<T>::run(fw[c][a], fa[c][b], acc[a][b]);
    };
    // two k-groups of reads ahead of the MFMAs that consume them (pinned: see FRAG_FENCE)
    rd(0); rd(1);
    FRAG_FENCE();
    rd(2); mm(0);
    FRAG_FENCE();
    rd(3); mm(1);
    FRAG_FENCE();
    mm(2); mm(3);
    FRAG_FENCE();
  };

  // prologue: whole halo of chunk 0 + weights of step 0
#pragma unroll
  for (int j = 0; j < HJ; ++j) issue_h(0, 0, j);
  issue_w(0, 0, 0);
  if (GN) issue_gn(0);
  constexpr int CVN = BN / 8, RP = 256 / CVN, NPASS = BM / RP;
  const int e_cg = tid % CVN, e_rr = tid / CVN;
  const int e_co = n0 + e_cg * 8;
  float bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bs[j] = (p.bias && e_co < p.Cout) ? p.bias[e_co + j] : 0.f;
  u32x4 rres[NPASS][8 / EPV];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (GN) {                                                  // chunk 0: the whole stage is normalised before the first MFMA
    transform(0, 0, -1);
    __syncthreads();
  }
  int c = 0, t = 0;
  for (int it = 0; it + 1 < nit; ++it) {
    int tn = t + 1, cn = c;
    if (tn == p.ntaps) { tn = 0; ++cn; }
    issue_w((it + 1) & 1, tn, cn);                           // next step's weights under this step's MFMAs
    if (c + 1 < nchunk) {                                    // next chunk's halo, spread over this chunk's steps
#pragma unroll
      for (int j = 0; j < HJ; ++j)
        if (j % p.ntaps == t) issue_h((c + 1) & 1, c + 1, j);
      if (GN && t == 0) issue_gn(c + 1);                     // (its ring slot was last read in steps 6-8 of chunk c - 1)
    }
    compute(it & 1, c & 1, t);
    if (GN && c + 1 < nchunk && t >= HJ && t < HJ + 3)       // the next chunk's halo landed with the barrier of step HJ - 1:
      transform((c + 1) & 1, c + 1, t - HJ);                 // a third of its slots in each of the three remaining steps (9 taps)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    t = tn;
    c = cn;
  }
  auto row_of = [&](int ml) { return mframe + (int64_t)(h0 + (ml >> 4)) * p.D2 + w0 + (ml & 15); };
  if (p.R) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int64_t m = row_of(e_rr + ps * RP);
#pragma unroll
      for (int h = 0; h < 8 / EPV; ++h)
        rres[ps][h] = *(const u32x4*)(e_co < p.Cout ? p.R + (m * p.ldr + e_co + h * EPV) * ES : (const char*)g_zero_page);
    }
  }
  compute((nit - 1) & 1, c & 1, t);
  __syncthreads();                                           // every wave is past its last operand read: sC may alias
  if (p.R) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
      for (int h = 0; h < 8 / EPV; ++h) LAUNDER4(rres[ps][h]);
  }

#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
      const int ml = wr * 64 + b2 * 32 + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = wc * 64 + a * 32 + 8 * q + 4 * half;
        f32x4 v = {acc[a][b2][4 * q], acc[a][b2][4 * q + 1], acc[a][b2][4 * q + 2], acc[a][b2][4 * q + 3]};
        *(f32x4*)(sC + ml * LDC + col) = v;
      }
    }
  __syncthreads();
  if (e_co < p.Cout) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int ml = e_rr + ps * RP;
      const int64_t m = row_of(ml);
      float v[8];
      const f32x4 c0 = *(const f32x4*)(sC + ml * LDC + e_cg * 8);
      const f32x4 c1 = *(const f32x4*)(sC + ml * LDC + e_cg * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = c0[j] + bs[j]; v[4 + j] = c1[j] + bs[4 + j]; }
      if (p.R) {
#pragma unroll
        for (int h = 0; h < 8 / EPV; ++h) {
          float rf[EPV];
          Elt<T>::unpack(rres[ps][h], rf);
#pragma unroll
          for (int j = 0; j < EPV; ++j) v[h * EPV + j] += rf[j];
        }
      }
#pragma unroll
      for (int h = 0; h < 8 / EPV; ++h)
        *(u32x4*)(p.Y + (m * p.ldy + e_co + h * EPV) * ES) = Elt<T>::pack(v + h * EPV);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Halo-tile main loop, 16 x 16 pixel patches (tile code 133, bf16): the 8 x 16 kernel above runs two 4-wave blocks per CU, each
// streaming its OWN copy of the weights (per chunk 23 KB of halo + 9 x 16 KB of weights for 128 pixels) with one K step of DMA in
// flight; measured 31 GB/s of LDS fill per CU at 0.74 - 0.98 PFLOP/s: neither the fill path (~27 B/clk/CU) nor the matrix pipe
// (41 % busy) is the limit, the serial DMA -> barrier -> MFMA step is.  Here ONE 8-wave block per CU owns a 16 x 16 patch: every
// weight byte feeds twice the pixels (per chunk 41 KB + 9 x 16 KB for 256 pixels: 204 flop per staged byte, was 113), which frees
// the LDS for a THREE-slot weight ring (two K steps of DMA in flight over a counted s_waitcnt and one raw s_barrier per step,
// like tile 132) beside the double-buffered halo.  Every wave issues exactly THREE DMA instructions per step (two weight row
// groups + a halo piece of the next chunk / an affine row / a dummy), so "vmcnt(3)" = "everything but the newest step has landed".
// K order (chunk, tap, k) and epilogue equal tile 130's: bitwise the same output.  GN as in tile 130: the in-place transform of
// halo slot j (= the row groups {w + 8 j}, DMA piece j of every wave, issued at step j) runs at step j + 2 of the previous chunk.
// Requires D1 % 16 == 0, D2 % 16 == 0, nine spatial taps, full frames, Cin % 64 == 0.
// Issue-side diet (round 3): the first version of this kernel issued ~290 non-MFMA instructions per K step and wave (64-bit per-lane
// DMA addresses with zero-page selects, register arrays indexed by the runtime tap, tap offsets fetched from LDS) beside its 16
// MFMAs - it ran exactly as fast as tile 130, which has the same per-step overhead: both were bound by instruction issue, not by
// the matrix pipe, the LDS fill or the DMA depth.  Now: the nine taps are unrolled (canonical 3 x 3 order is required, so the halo
// shift of a tap is a literal), every DMA is a buffer_load ... lds through a wave-uniform descriptor with a lane-constant 32-bit
// offset and a SCALAR step offset (padding rows: an offset beyond the descriptor's range - they are never fetched, the stage's
// padding rows are zeroed once).
template <bool GN>
__global__ __launch_bounds__(512, 2) void conv_gemm_halo16_kernel(const ConvGemmParams p) {
  typedef __bf16 T;
  constexpr int BM = 256, BN = 128, PH = 16, PW = 16, HWD = PW + 2, HR = (PH + 2) * HWD, HG = (HR + 7) / 8, HJ = (HG + 7) / 8;
  constexpr int NT = 9;
  constexpr int EPV = 8, ES = 2;
  constexpr int LDC = BN + 4;
  constexpr int A_B = HG * 8 * 128;            // halo stage: 41 groups of 8 rows
  constexpr int W_B = 128 * 128, NWS = 3;
  constexpr int OPS_B = 2 * A_B + NWS * W_B;
  constexpr int MAIN_B = (OPS_B > BM * LDC * 4) ? OPS_B : BM * LDC * 4;
  static_assert(HJ == 6, "halo pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                             // [2][HG*8 rows][128 B]
  char* sW = smem + 2 * A_B;                   // [3][128 rows][128 B]
  float* sC = (float*)smem;
  float* sGN = (float*)(smem + MAIN_B);        // [2 chunk parities][a (64 channels) | b (64 channels)]
  char* sDummy = smem + MAIN_B + 1024;         // 256 B: target of the DMA instructions that only keep the per-step count uniform

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave & 1, wr = wave >> 1;
  const int half = lane >> 5, l31 = lane & 31;

  const int Nt = (p.Cout + BN - 1) / BN;
  const int nwg = gridDim.x;
  int wgid;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nt = wgid % Nt, mt = wgid / Nt;
  const int n0 = nt * BN;
  const int TW = p.D2 / PW, tpf = TW * (p.D1 / PH);
  const int d0 = mt / tpf, trem = mt - d0 * tpf;
  const int h0 = (trem / TW) * PH, w0 = (trem % TW) * PW;
  const int64_t mframe = (int64_t)d0 * p.D1 * p.D2;

  const int nchunk = p.Cin >> 6;
  const int K = p.Cin * NT;

  const int lrow = lane >> 3, pc = lane & 7;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // ---- DMA descriptors (wave-uniform) and lane-constant byte offsets.  A: the frame's rows; W: this block's 128 weight rows.
  const uint32_t OOB = 0xfffffff0u;            // beyond every descriptor: the lane fetches nothing and writes nothing useful (see below)
  const int64_t a_bytes = ((int64_t)p.D1 * p.D2 - 1) * p.lda * ES + (int64_t)p.Cin * ES;
  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + mframe * p.lda * ES), 0, (int)(a_bytes < 0x7fffffff ? a_bytes : 0x7fffffff), 0x00020000);
  const int w_rows = p.Cout - n0 < BN ? p.Cout - n0 : BN;
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * K * ES), 0, w_rows * K * ES, 0x00020000);
  uint32_t h_off[HJ];                          // wave w stages halo row groups g = w + 8 j
  unsigned h_okmask = 0;
#pragma unroll
  for (int j = 0; j < HJ; ++j) {
    const int g = wave + 8 * j;
    const int r = 8 * g + lrow;
    const int hr = r / HWD, hc = r - hr * HWD;
    const int hh = h0 - 1 + hr, ww = w0 - 1 + hc;
    const bool ok = g < HG && r < HR && (unsigned)hh < (unsigned)p.D1 && (unsigned)ww < (unsigned)p.D2;
    const int logical = pc ^ ((r >> 1) & 7);
    h_off[j] = ok ? (uint32_t)((((int64_t)hh * p.D2 + ww) * p.lda + logical * EPV) * ES) : OOB;
    h_okmask |= (ok ? 1u : 0u) << j;
  }
  uint32_t w_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (wave + 8 * i) + lrow;
    const int logical = pc ^ ((row >> 1) & 7);
    w_off[i] = row < w_rows ? (uint32_t)((row * K + logical * EPV) * ES) : OOB;     // rows past Cout: their outputs are never stored
  }
  // ---- fused input GroupNorm
  constexpr int NSLOT = (HG * 64 + 511) / 512;                 // 16-byte slots of a halo stage per thread (6; the last one: wave 0 only)
  unsigned gvalid = 0;
  const float* gn_src = nullptr;
  const int glc = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;          // first channel (inside a chunk) of this thread's 16 bytes: the same in every slot
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int r = (tid >> 3) + 64 * i;
    const int hr = r / HWD, hc = r - hr * HWD;
    const bool ok = r < HR && (unsigned)(h0 - 1 + hr) < (unsigned)p.D1 && (unsigned)(w0 - 1 + hc) < (unsigned)p.D2;
    gvalid |= (ok ? 1u : 0u) << i;
  }
  if (GN) {
    const int sidx = min((int)(mframe / p.gn_rows), p.gn_S - 1);
    gn_src = (wave == 0 ? p.gn_a : p.gn_b) + (int64_t)sidx * p.Cin + lane;
  }
  // padding rows of both halo stages are zeroed ONCE: an out-of-range DMA lane may or may not write its (zero) result, the
  // in-place transform leaves padding slots alone, and the validity of a slot does not depend on the chunk
#pragma unroll
  for (int i = 0; i < NSLOT; ++i)
    if ((i * 512 + 512 <= HG * 64 || tid < HG * 64 - i * 512) && !((gvalid >> i) & 1u)) {
      *(u32x4*)(sA + tid * 16 + i * 8192) = u32x4{0u, 0u, 0u, 0u};
      *(u32x4*)(sA + A_B + tid * 16 + i * 8192) = u32x4{0u, 0u, 0u, 0u};
    }
  __syncthreads();

  auto dma_dummy = [&]() { __builtin_amdgcn_global_load_lds((gptr_t)g_zero_page, (lptr_t)sDummy, 4, 0, 0); };
  auto issue_w1 = [&](int slot, int c2, int t2, int i) {       // weights of (chunk c2, tap t2): piece i of two per wave
    const int soff = (t2 * p.Cin + c2 * 64) * ES;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lptr_t)(sW + slot * W_B + (wave + 8 * i) * 1024), 16, w_off[i], soff, 0, 0);
  };
  auto issue_w = [&](int slot, int c2, int t2) { issue_w1(slot, c2, t2, 0); issue_w1(slot, c2, t2, 1); };
  auto issue_h = [&](int buf, int c, int j) {                  // exactly one DMA instruction (j is a literal at every call site)
    if (wave + 8 * j < HG)                                     // wave-uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lptr_t)(sA + buf * A_B + (wave + 8 * j) * 1024), 16, h_off[j], c * 128, 0, 0);
    else
      dma_dummy();
  };
  auto issue_gn = [&](int c) {                                 // exactly one DMA instruction: 64 floats of a (wave 0) / b (wave 1) -> ring slot c & 1
    if (GN && wave < 2) __builtin_amdgcn_global_load_lds((gptr_t)(gn_src + c * 64), (lptr_t)(sGN + (c & 1) * 128 + wave * 64), 4, 0, 0);
    else dma_dummy();
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int xsw = (l31 >> 1) & 7;
  int rb[2];                                   // halo row of this lane's pixel (tap 0,0) per m sub-tile
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int px = wr * 64 + b * 32 + l31;
    rb[b] = ((px >> 4) + 1) * HWD + (px & 15) + 1;
  }
  const char* bWl = sW + (wc * 64 + l31) * 128;                // + slot * W_B
  auto compute = [&](int wslot, int bufa, int toff, auto&& dma, auto&& tr) {   // toff: the tap's halo-row shift (a literal); dma(i): the step's i-th DMA instruction; tr(g): norm piece behind MFMA group g
    const char* bW = bWl + wslot * W_B;
    const char* bA[2];
    int key[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int r = rb[b] + toff;
      key[b] = (r >> 1) & 7;
      bA[b] = sA + bufa * A_B + r * 128;
    }
    u32x4 fw[4][2], fa[4][2];
    auto rd = [&](int c) {
#pragma unroll
      for (int a = 0; a < 2; ++a) fw[c][a] = *(const u32x4*)(bW + a * 32 * 128 + (((2 * c + half) ^ xsw) * 16));
#pragma unroll
      for (int b = 0; b < 2; ++b) fa[c][b] = *(const u32x4*)(bA[b] + (((2 * c + half) ^ key[b]) * 16));
    };
    auto mm = [&](int c) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) Mma<T>::run(fw[c][a], fa[c][b], acc[a][b]);
    };
    // (round 4) the step's three DMA instructions go out BEHIND MFMA groups, not in front of the first fragment read: ~100 issue
    // cycles apiece that every wave used to spend between the barrier and its first MFMA
    rd(0); rd(1);
    FRAG_FENCE();
    rd(2); mm(0); dma(0); tr(0);
    FRAG_FENCE();
    rd(3); mm(1); dma(1); tr(1);
    FRAG_FENCE();
    mm(2); dma(2); tr(2);
    FRAG_FENCE();
    mm(3); tr(3);
    FRAG_FENCE();
  };
  // halo slot i of stage buf (chunk c): act(x a + b) in place, dwords d0 .. d0 + nd - 1 of its 16 bytes (two channels each).  Inside
  // the main loop a slot is normalised in four quarters, one behind each MFMA group of a step (round 4): a whole slot (~65 VALU
  // instructions) behind the step's sixteen MFMAs was paid in full - a wave hides ~5 instructions in the shadow of one MFMA.
  u32x4 tv;
  uint32_t ty[4];
  auto transform = [&](int buf, int c, int i, int d0, int nd) {
    if (i * 512 + 512 <= HG * 64 || tid < HG * 64 - i * 512) { // the partial last slot: wave-uniform
      const float* ap = sGN + (c & 1) * 128 + glc;
      char* q = sA + buf * A_B + tid * 16 + i * 8192;
      if (d0 == 0) tv = *(const u32x4*)q;
#pragma unroll
      for (int d = d0; d < d0 + nd; ++d) {
        const float w0 = __uint_as_float(tv[d] << 16) * ap[2 * d] + ap[64 + 2 * d];
        const float w1 = __uint_as_float(tv[d] & 0xffff0000u) * ap[2 * d + 1] + ap[65 + 2 * d];
        const float y[2] = {p.gn_act ? silu_f(w0) : w0, p.gn_act ? silu_f(w1) : w1};
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        const bf16x2 pk = {(__bf16)y[0], (__bf16)y[1]};
        ty[d] = __builtin_bit_cast(uint32_t, pk);
      }
      if (d0 + nd == 4) {
        const u32x4 y4 = {ty[0], ty[1], ty[2], ty[3]};
        *(u32x4*)q = ((gvalid >> i) & 1u) ? y4 : tv;
      }
    }
  };

  // ---- prologue: halo of chunk 0 (six groups of three DMA instructions with the affine rows of chunks 0 and 1 and the weights of
  //      steps 0 and 1 in the free positions), then everything of chunk 0 is normalised before the first MFMA
#pragma unroll
  for (int j = 0; j < HJ; ++j) issue_h(0, 0, j);
  issue_gn(0);
  if (nchunk > 1) issue_gn(1); else dma_dummy();
  dma_dummy();
  issue_w(0, 0, 0);
  dma_dummy();
  issue_w(1, 0, 1);
  dma_dummy();
  constexpr int CVN = BN / 8, RP = 512 / CVN, NPASS = BM / RP;
  const int e_cg = tid % CVN, e_rr = tid / CVN;
  const int e_co = n0 + e_cg * 8;
  float bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bs[j] = (p.bias && e_co < p.Cout) ? p.bias[e_co + j] : 0.f;
  u32x4 rres[NPASS];                                          // residual rows of the epilogue passes: requested in front of the LAST step's MFMAs
  auto row_of = [&](int ml) { return mframe + (int64_t)(h0 + (ml >> 4)) * p.D2 + w0 + (ml & 15); };
  if (GN) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (also the weights of steps 0 / 1: the loop's first waits are then no-ops)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) transform(0, 0, i, 0, 4);
    // (the loop's first barrier orders these LDS writes before the first fragment reads)
  }
  for (int c = 0; c < nchunk; ++c) {
    const bool more = c + 1 < nchunk;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); // everything but the newest group of three has landed: this step is complete
      __builtin_amdgcn_s_barrier();                            // ... for every wave; every wave is past its fragment reads of the previous step
      asm volatile("" ::: "memory");
      const int wslot = (c * NT + t) % NWS;
      if (t == NT - 1 && !more && p.R) {   // (the pass loop used to load a row, wait, add, store - a serial HBM round trip per pass; the loads
        //                                    sit in front of this step's DMA group in the in-order queue and are drained by the final wait)
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
          rres[ps] = *(const u32x4*)(e_co < p.Cout ? p.R + (row_of(e_rr + ps * RP) * p.ldr + e_co) * ES : (const char*)g_zero_page);
      }
      // the DMA group of this step: weights of the step after next into the slot the previous step left + one instruction for a later chunk
      auto group = [&](int i) {
        const int t2 = (t + 2) % NT, c2 = c + (t + 2) / NT;
        if (i < 2) {
          if (c2 < nchunk) issue_w1((wslot + 2) % NWS, c2, t2, i);
          else dma_dummy();
        } else if (t < HJ) { if (more) issue_h((c + 1) & 1, c + 1, t); else dma_dummy(); }
        else if (t == NT - 1 && c + 2 < nchunk) issue_gn(c + 2); // (its ring slot was last read in steps 2 .. 7 of chunk c - 1)
        else dma_dummy();
      };
      constexpr int TOFF[9] = {-19, -18, -17, -1, 0, 1, 17, 18, 19};   // (dh, dw) in row-major 3 x 3 order -> halo-row shift dh * 18 + dw
      int toff = TOFF[t];
      asm volatile("" : "+s"(toff));       // keep the 72 per-tap fragment addresses out of the registers: recomputed per step (~20 VALU)
      const bool tr_on = GN && more && t >= 2 && t < 2 + NSLOT;                        // halo piece t - 2 landed with this step's wait
      compute(wslot, c & 1, toff, group, [&](int g) { if (tr_on) transform((c + 1) & 1, c + 1, t - 2, g, 1); });
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                             // every wave is past its last operand read: sC may alias
  if (p.R) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) LAUNDER4(rres[ps]);
  }

#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
      const int ml = wr * 64 + b2 * 32 + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = wc * 64 + a * 32 + 8 * q + 4 * half;
        f32x4 v = {acc[a][b2][4 * q], acc[a][b2][4 * q + 1], acc[a][b2][4 * q + 2], acc[a][b2][4 * q + 3]};
        *(f32x4*)(sC + ml * LDC + col) = v;
      }
    }
  __syncthreads();
  if (e_co < p.Cout) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int ml = e_rr + ps * RP;
      const int64_t m = row_of(ml);
      float v[8];
      const f32x4 c0 = *(const f32x4*)(sC + ml * LDC + e_cg * 8);
      const f32x4 c1 = *(const f32x4*)(sC + ml * LDC + e_cg * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = c0[j] + bs[j]; v[4 + j] = c1[j] + bs[4 + j]; }
      if (p.R) {
        float rf[EPV];
        Elt<T>::unpack(rres[ps], rf);
#pragma unroll
        for (int j = 0; j < EPV; ++j) v[j] += rf[j];
      }
      *(u32x4*)(p.Y + (m * p.ldy + e_co) * ES) = Elt<T>::pack(v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-strip main loop for 1x1 convolutions (tile code 131, bf16, Cin = 128 / 256 / 384).
// The tiled main loops above are latency-bound on the 1x1 convs: K is 2-6 K steps, one K step is in flight per block, and every
// output tile pays prologue + 2-6 dependent L2 round trips + epilogue (qkv conv 256 -> 768 at ds2: 65 us against ~20 us of HBM
// time).  Here the ACTIVATIONS are stationary: a wave loads its 32*RF rows ONCE, straight from global memory into MFMA B-operand
// fragments (16 bytes per lane per 16-channel k-step; all loads of the strip in flight together), applies GroupNorm(+FiLM)(+SiLU)
// once in registers (the tiled GN loader redoes it for every column tile, which is why wide convs could not fuse it), and then
// walks the output channels in chunks of CC: the chunk's weights [CC][Cin] stream through a two-stage LDS ring by
// global_load_lds (same 128-byte-row, XOR-swizzled image as the direct-to-LDS loop, one plane per 64 channels), are shared by
// the four waves and come from L2.  The epilogue never touches LDS: v_permlane32_swap pairs the two half-waves' 4-channel groups
// into 8 consecutive channels per lane (16-byte stores / residual loads), and the stores of a chunk's last sub-tile are issued
// after the chunk's barrier so that the wait for the weight DMA does not sit on their acknowledgements.
// K order and epilogue order equal the tiled loops': Y is bitwise identical to tiles 64 / 128 / 129.  The output statistics
// (RF = 2: a wave owns whole 64-row records) are folded in this kernel's own fixed order (rows f = 0, 1 of a lane, then a
// recursive-halving butterfly over the 32 lanes of a half-wave), so launches that emit statistics are chosen by layer geometry,
// never by timing (ops.strip_tile_pinned).
// Grid = row strips x nsplit column ranges (nsplit fills the chip when M is small; results do not depend on it).
// Round 6: (1) the weight DMA is issued through asm and every wait of the loop is explicit: with an LDS DMA the compiler KNOWS of in flight it
// waits vmcnt(0) in front of every LDS read that might alias it (the bias reads of each epilogue unit = the acknowledgement of the store
// before it; in straight-line code also the first fragment read behind the issue = the prefetch itself); (2) the residual operand is a
// template parameter and rows past M store to a sink: no divergent branch in the epilogue; (3) the K >= 256 instances run a software
// pipeline over the sub-tiles (see PIPE below): ds2 qkv 46.7 -> 39.1 us, ds4 qkv 30.2 -> 25.2 us (profiles/r06_strip_pipeline.txt).
// (halfwave_total - the DPP fold of the quad statistics - lives in mmd_common.h: the fused VideoConv kernel shares it.)
// The pipelined strip loop requests its residual pieces one block ahead and waits for them with exact counts.  A compiler-visible load would
// be waited for by the compiler's count, which does not know the weight DMA (asm); an asm load with an OUTPUT operand is 'ready' as far as the
// compiler knows, and it copies the registers (operand matching, loop edges) before the wait - measured: wrong results, a fault when the
// destination was dead.  So the pieces live in registers the compiler never allocates: the kernels with a residual are limited to 248 VGPRs
// (amdgpu_num_vgpr) and v[248:255] belong to these two statements; the second waits and unpacks into ordinary outputs.
#define STRIP_RES_LOAD(J2, off, base)                                                                                        \
  do {                                                                                                                       \
    if ((J2) == 0) asm volatile("global_load_dwordx4 v[248:251], %0, %1" :: "v"(off), "s"(base) : "v248", "v249", "v250", "v251");  \
    else asm volatile("global_load_dwordx4 v[252:255], %0, %1" :: "v"(off), "s"(base) : "v252", "v253", "v254", "v255");       \
  } while (0)
#define STRIP_RES_UNPACK_ASM(A, B, C, D)                                                                                     \
  "s_waitcnt vmcnt(%8)\n\tv_lshlrev_b32 %0, 16, " A "\n\tv_and_b32 %1, 0xffff0000, " A "\n\tv_lshlrev_b32 %2, 16, " B "\n\tv_and_b32 %3, 0xffff0000, " B \
  "\n\tv_lshlrev_b32 %4, 16, " C "\n\tv_and_b32 %5, 0xffff0000, " C "\n\tv_lshlrev_b32 %6, 16, " D "\n\tv_and_b32 %7, 0xffff0000, " D
#define STRIP_RES_WAIT(J2, rf, n)                                                                                            \
  do {                                                                                                                       \
    if ((J2) == 0)                                                                                                           \
      asm volatile(STRIP_RES_UNPACK_ASM("v248", "v249", "v250", "v251")                                                      \
                   : "=v"(rf[0]), "=v"(rf[1]), "=v"(rf[2]), "=v"(rf[3]), "=v"(rf[4]), "=v"(rf[5]), "=v"(rf[6]), "=v"(rf[7]) : "n"(n));  \
    else                                                                                                                     \
      asm volatile(STRIP_RES_UNPACK_ASM("v252", "v253", "v254", "v255")                                                      \
                   : "=v"(rf[0]), "=v"(rf[1]), "=v"(rf[2]), "=v"(rf[3]), "=v"(rf[4]), "=v"(rf[5]), "=v"(rf[6]), "=v"(rf[7]) : "n"(n));  \
  } while (0)
// s_waitcnt immediate of gfx9: vmcnt(n) lgkmcnt(0), expcnt untouched
__host__ __device__ constexpr int wait_imm(int vm) { return (vm & 15) | ((vm >> 4) << 14) | 0x0070; }
template <int KS, int RF, int CC, int GNM, int STM, bool HR>   // HR: residual operand (compile time: behind a runtime branch hipcc waits vmcnt(0) in
                                             // front of EVERY use of the residual registers, i.e. for the acknowledgement of the store issued just before). STM: output statistics 0 none / 1 quad records (compile
                                             // time: the runtime branches cost the K = 128 instance 30 spilled registers).  GNM: 0 no GroupNorm, 1 fused affine, 2 fused affine + SiLU (compile time: two copies of the
                                             // normalisation in one kernel spill ~100 registers around the branch)
__device__ __forceinline__ void conv1x1_strip_body(const ConvGemmParams& p, const int nsplit) {
  constexpr int K = 64 * KS;                 // input channels
  constexpr int NCG = 4 * KS;                // 16-channel k-steps (one MFMA each)
  constexpr int BR = 128 * RF;               // rows per block: 4 waves x RF fragments of 32 rows
  constexpr int NA = CC / 32;                // 32-channel output sub-tiles per chunk
  constexpr int GP = CC / 32;                // weight DMA instructions per wave per 64-channel plane (CC / 8 row groups over 4 waves)
  constexpr int PLANE_B = CC * 128, STAGE_B = KS * PLANE_B;
  // software-pipelined sub-tile loop (below): the instances whose launches run one workgroup per CU (K >= 256: the ds2 / ds4 / ds8 levels; the K = 128
  // launches of the ds1 level put 3 - 4 workgroups on a CU and the hardware overlaps their waves).  K = 256 with a residual does not fit the
  // register file (two accumulator sets + the rows' 128 operand registers + the residual pieces: ~45 spilled registers)
  constexpr bool PIPE = KS >= 4 && !(KS == 4 && HR);
  constexpr bool DEFER = KS > 2 || RF == 1;  // K = 128 / 256 with two fragments: 1-4 chunks per block, the 16 registers buy a third wave per SIMD
  static_assert(RF == 2 || CC == 32, "one row fragment per wave: one sub-tile per chunk, deferred epilogue");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                           // [2 stages][KS planes][CC rows][128 B], 16-byte chunks XOR-swizzled by (row >> 1) & 7

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  // the nsplit blocks of one row strip get consecutive ids inside one XCD's contiguous range: they share the strip's rows in that L2
  const int nwg = gridDim.x;
  int wgid;
  {
    const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int sp = wgid % nsplit, mt = wgid / nsplit;
  const int Cs = p.Cout / nsplit, cbase = sp * Cs, nchunk = Cs / CC;
  const int m0 = mt * BR;
  float* sBias = (float*)(smem + 2 * STAGE_B);          // [Cs] bias of this block's column range
  float* sGN = sBias + ((Cs + 3) & ~3);                 // [2 slices][a | b][K] fused GroupNorm affine
  float* sRec = sGN + (GNM != 0 ? 4 * K : 0);           // RF = 1: [2 chunk parities][4 waves][NA * 2][32] half-record statistics

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int lrow = lane >> 3, pc = lane & 7;
  const char* w_ptr[GP];
#pragma unroll
  for (int ih = 0; ih < GP; ++ih) {
    const int row = 8 * (ih * 4 + wave) + lrow;          // row of the chunk this lane fetches 16 bytes of
    const int logical = pc ^ ((row >> 1) & 7);
    w_ptr[ih] = p.W + ((int64_t)(cbase + row) * K + logical * 8) * 2;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)sW;
  auto issue = [&](int stage, int ci) {
    const int64_t off = (int64_t)ci * CC * K * 2;
#pragma unroll
    for (int pl = 0; pl < KS; ++pl)
#pragma unroll
      for (int ih = 0; ih < GP; ++ih)
        // asm, not __builtin_amdgcn_global_load_lds: with an LDS DMA the compiler KNOWS of in flight, it puts s_waitcnt vmcnt(0) in front of
        // every LDS read that might alias it - the bias reads of each epilogue unit (= the acknowledgement of the store issued just
        // before: four serialised store round trips per sub-tile) and, in straight-line code, the first weight fragment read after the
        // issue (= the prefetch).  Untracked VMEM instructions only make the compiler's own counted waits conservative (in-order return).
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                     :: "v"(w_ptr[ih] + off + pl * 128), "s"(lds0 + stage * STAGE_B + pl * PLANE_B + (ih * 4 + wave) * 1024) : "memory", "m0");
  };

  issue(0, 0);
  // the strip's activations: fragment f, k-step cg = 8 channels [16 cg + 8 half, +8) of row l31 - the B operand of the MFMA
  int rowc[RF];
  bool rok[RF];
  uint32_t yoff[RF], roff[RF];                            // byte offset of the lane's row (+ its half's 16 bytes) in Y / R: < 4 GB (dispatch)
  u32x4 xa[RF][NCG];
#pragma unroll
  for (int f = 0; f < RF; ++f) {
    const int row = m0 + wave * (32 * RF) + f * 32 + l31;   // m0 + BR may pass M by less than one block: no overflow (M < 2^31 - 256)
    rok[f] = row < p.M;
    // rows past M compute row M - 1 once more - same operands, same bits - and STORE it there as well: the stores are unconditional (a
    // store behind a divergent branch costs the straight-line epilogue its exact wait counts: the compiler re-waits, vmcnt(0), for loads
    // 'pending' on the skipped path); uniform base + 32-bit lane offset: no per-lane 64-bit address arithmetic
    rowc[f] = rok[f] ? row : p.M - 1;
    yoff[f] = (uint32_t)rowc[f] * (uint32_t)(p.ldy * 2) + 16 * half;
    roff[f] = (uint32_t)rowc[f] * (uint32_t)(p.ldr * 2) + 16 * half;
    if (p.ntaps == 1) {                                  // block-uniform
      const char* ap = p.A + ((int64_t)rowc[f] * p.lda + half * 8) * 2;
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) xa[f][cg] = *(const u32x4*)(ap + cg * 32);
    } else {
      // several taps (temporal / audio k=3 at 128 channels): K index = tap * Cin + ci, Cin a multiple of 64, so a 64-channel plane
      // lies inside one tap; its source row is the row shifted by the tap, or the zero row outside (D0, D1, D2)
      const int D12 = p.D1 * p.D2;
      const int q2 = rowc[f] % p.D2, q1 = (rowc[f] / p.D2) % p.D1, q0 = (rowc[f] / D12) % p.D0;
#pragma unroll
      for (int pl = 0; pl < KS; ++pl) {
        const int tap = (64 * pl) / p.Cin, ci0 = 64 * pl - tap * p.Cin;
        const int o0 = p.taps[3 * tap], o1 = p.taps[3 * tap + 1], o2 = p.taps[3 * tap + 2];
        const bool ok = (unsigned)(q0 + o0) < (unsigned)p.D0 && (unsigned)(q1 + o1) < (unsigned)p.D1 && (unsigned)(q2 + o2) < (unsigned)p.D2;
        const int64_t src = (int64_t)rowc[f] + (int64_t)o0 * D12 + o1 * p.D2 + o2;
        const char* ap = ok ? p.A + (src * p.lda + ci0 + half * 8) * 2 : (const char*)g_zero_row + half * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) xa[f][4 * pl + c] = *(const u32x4*)(ap + c * 32);
      }
    }
  }
  // bias of the column range and the GroupNorm affine rows of the (at most two) slices of the strip -> LDS.  All global loads
  // first (branch-free: clamped indices, the zero page when there is no bias), then the LDS writes: one round trip for everything
  constexpr bool gn = GNM != 0;
  float bias_v[8];                                       // Cs <= 2048
  {
    const float* bsrc = p.bias ? p.bias + cbase : (const float*)g_zero_page;
    const int bmul = p.bias ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) bias_v[i] = bsrc[min(tid + 256 * i, Cs - 1) * bmul];
  }
  int gsel[RF];
#pragma unroll
  for (int f = 0; f < RF; ++f) gsel[f] = 0;
  float tv[KS];                                          // entry i = tid + 256 e of [2 slices][a | b][K]
  if (gn) {
    const int gnr = (int)p.gn_rows;                      // < 2^31: gn_S * gn_rows == M
    const int s0 = m0 / gnr;                             // gn_rows >= BR: the strip touches at most two slices
#pragma unroll
    for (int e = 0; e < KS; ++e) {
      const int i = tid + 256 * e;
      const int sl = i / (2 * K), ab = (i / K) & 1, c = i % K;
      const int sidx = min(s0 + sl, p.gn_S - 1);
      tv[e] = (ab ? p.gn_b : p.gn_a)[(int64_t)sidx * K + c];
    }
#pragma unroll
    for (int f = 0; f < RF; ++f) gsel[f] = min(rowc[f] / gnr - s0, 1) * 2 * K;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (tid + 256 * i < Cs) sBias[tid + 256 * i] = bias_v[i];
  if (gn) {
#pragma unroll
    for (int e = 0; e < KS; ++e) sGN[tid + 256 * e] = tv[e];
  }
  // (the builtin, not asm: the compiler's own scoreboard must see the wait, or it re-waits - vmcnt(0) - for 'pending' loads at their next
  // use, which by then sits behind the next chunk's DMA issue)
  __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) lgkmcnt(0)
  __builtin_amdgcn_s_barrier();                          // chunk 0 landed; bias / affine tables visible
  if (gn) {
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        float x[8];
        Elt<__bf16>::unpack(xa[f][cg], x);
        const float* ap = sGN + gsel[f] + cg * 16 + half * 8;
        const float* bp = ap + K;
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
          const f32x4 av = *(const f32x4*)(ap + e), bv = *(const f32x4*)(bp + e);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float y = x[e + k] * av[k] + bv[k];
            x[e + k] = GNM == 2 ? silu_f(y) : y;
          }
        }
        u32x4 y = Elt<__bf16>::pack(x);
        // pin the result here: otherwise the arithmetic is sunk to its first use (the MFMA loop) while all 8 KS table reads stay up
        // front, and the kernel spills hundreds of registers
        asm volatile("" : "+v"(y.x), "+v"(y.y), "+v"(y.z), "+v"(y.w));
        xa[f][cg] = y;
      }
  }

  const int xsw = (l31 >> 1) & 7;
  const bool wave_ok = (int64_t)m0 + wave * (32 * RF) < p.M;      // wave-uniform: statistics records are whole waves (RF = 2)
  const int64_t rec = ((int64_t)m0 + wave * (32 * RF)) / 64;
  if constexpr (PIPE) {
    // ---- software pipeline over the 32-column sub-tiles (round 6).  Block s = the MFMAs of sub-tile s into one accumulator set with the
    // epilogue of sub-tile s - 1 (the other set) between them: a lone wave per SIMD - the launches of one workgroup per CU - used to run
    // the two back to back, the matrix pipe idle under ~300 epilogue VALU instructions per chunk and the VALU idle under 2048 MFMA cycles.
    // Every vector-memory LOAD of the loop is asm (weight DMA, residual rows) and every wait explicit and counted: vmcnt retires in order,
    // the stores are the only instructions the compiler tracks and nothing ever waits for them.
    //   per epilogue unit (j2, f): [wait for the unit's residual piece] ... store, [request the same piece of the NEXT sub-tile into the
    //   same registers]: VM_U instructions, so behind a residual request sit 2 (NU - 1) guaranteed instructions (+ D when a chunk's DMA
    //   issue lies between) until its use one block later; behind a chunk's DMA NA * NU * VM_U until the chunk boundary.  The conditional
    //   statistics stores only add to the real count: the waits get stricter by instructions that are old anyway.
    constexpr int NU = 2 * RF, D = KS * GP, VM_U = HR ? 2 : 1;
    const int NS = nchunk * NA;
    f32x16 acc[2][RF];
    static_assert(!HR || RF == 1, "two residual pieces in v[248:255]");
    float erf[8];                                        // the unit's residual piece, unpacked
    float keep[2][2][2];                                 // [block parity][j2][sum | sum of squares] of the last epilogue's records
    if constexpr (HR) {
#pragma unroll
      for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) STRIP_RES_LOAD(j2, roff[f], p.R + (int64_t)(cbase + 16 * j2) * 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (sub-tile 0's residual: complete before the loop, its waits below are then no-ops)
    }
    // one MFMA of sub-tile s_: index i = (pl * 4 + c) * RF + f; the weight fragment of k-step (pl, c) is read PF k-steps ahead
    constexpr int NM = KS * 4 * RF, NKS = KS * 4, PF = RF == 2 ? 1 : 2;
    u32x4 fwr[PF + 1];
    auto frag_read = [&](int s_, int ks) __attribute__((always_inline)) {
      const int st = (s_ / NA) & 1, a = s_ % NA;
      fwr[ks % (PF + 1)] = *(const u32x4*)(sW + st * STAGE_B + (a * 32 + l31) * 128 + (ks >> 2) * PLANE_B + (((2 * (ks & 3) + half) ^ xsw) * 16));
    };
    auto mfma_one = [&](int i, f32x16 (&ac)[RF]) __attribute__((always_inline)) {
      const int ks = i / RF, f = i % RF;
      if (ks == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ac[f][r] = 0.f;
      }
      Mma<__bf16>::run(fwr[ks % (PF + 1)], xa[f][ks], ac[f]);
    };
    // the epilogue of a sub-tile as a list of steps of 3 - 8 VALU instructions each (all indices are constants after unrolling):
    //   per j2, per f: four steps "pair q" (permlane swap, bias, residual), one step "pack, store, next residual request", four
    //   statistics steps; per j2: four half-wave folds and one step that keeps / parks the record.
    constexpr int SPU = 5 + (STM ? 4 : 0), PJ = RF * SPU + (STM ? 5 : 0), NSTEP = 2 * PJ;
    float ev[8], eu[4], et[4];
    u32x4 epk;
    f32x4 eb[2][2];
    auto epi_begin = [&](int s_) __attribute__((always_inline)) {
      eb[0][0] = *(const f32x4*)(sBias + s_ * 32 + 8 * half);
      eb[0][1] = *(const f32x4*)(sBias + s_ * 32 + 8 * half + 4);
    };
    auto epi_step = [&](int s_, f32x16 (&ac)[RF], float (&kp)[2][2], int t, bool dma_between, bool last) __attribute__((always_inline)) {
      const int j2 = t / PJ, r = t % PJ;
      const int cb = s_ * 32;
      const int a = s_ % NA, par = (s_ / NA) & 1;
      if (r == 0) {
        if (j2 == 0) {
          eb[1][0] = *(const f32x4*)(sBias + cb + 16 + 8 * half);
          eb[1][1] = *(const f32x4*)(sBias + cb + 16 + 8 * half + 4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) eu[k] = 0.f;
      }
      if (r < RF * SPU) {
        const int f = r / SPU, k = r % SPU;
        if (k < 4) {
          if (HR && k == 0) {
            // last sub-tile (no MFMAs, no further requests): behind this unit's request sit the 2 (NU - 1 - u) instructions of the units
            // behind it in the previous block and the u stores of this epilogue
            if (last) STRIP_RES_WAIT(j2, erf, 2 * (NU - 1 - (j2 * RF + f)) + (j2 * RF + f));
            else if (dma_between) STRIP_RES_WAIT(j2, erf, 2 * (NU - 1) + D);
            else STRIP_RES_WAIT(j2, erf, 2 * (NU - 1));
          }
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ac[f][8 * j2 + k]), __float_as_uint(ac[f][8 * j2 + 4 + k]), false, false);
          ev[k] = __uint_as_float(sw[0]) + eb[j2][0][k];
          ev[4 + k] = __uint_as_float(sw[1]) + eb[j2][1][k];
          if constexpr (HR) {
            ev[k] += erf[k];
            ev[4 + k] += erf[4 + k];
          }
        } else if (k == 4) {
          epk = Elt<__bf16>::pack(ev);
          *(u32x4*)(p.Y + (int64_t)(cbase + cb + 16 * j2) * 2 + yoff[f]) = epk;
          if (HR && !last) STRIP_RES_LOAD(j2, roff[f], p.R + (int64_t)(cbase + cb + 32 + 16 * j2) * 2);
        } else {                                         // statistics of the values as STORED: word q = values 2 q, 2 q + 1 of the lane's two quads
          const int q = k - 5;
          const float r0 = __uint_as_float(epk[q] << 16), r1 = __uint_as_float(epk[q] & 0xffff0000u);
          eu[q >> 1] += r0;
          eu[q >> 1] += r1;
          eu[2 + (q >> 1)] += r0 * r0;
          eu[2 + (q >> 1)] += r1 * r1;
        }
      } else {
        const int h = r - RF * SPU;
        if (h < 4) et[h] = halfwave_total(eu[h]);
        else {
          // lanes 16 / 17 of each half hold the (sum, sum of squares) of quad 0 / 1 of the lane group's 8 channels
          kp[j2][0] = (l31 & 1) ? et[1] : et[0];
          kp[j2][1] = (l31 & 1) ? et[3] : et[2];
          if (RF == 1) {
            // a wave holds HALF a record (32 rows): park the partial for the even wave of the pair, which adds (own + partner) behind the
            // next chunk boundary; the buffer alternates with the chunk parity of the sub-tile.  Unconditional (lanes that hold nothing
            // write a dump slot): a divergent branch here would cut the block's straight-line code in two
            float* d = sRec + ((l31 >> 1) == 8 ? (((par * 4 + wave) * (NA * 2) + a * 2 + j2) * 2 + half) * 4 + (l31 & 1) * 2 : 128 + lane * 2);
            d[0] = kp[j2][0];
            d[1] = kp[j2][1];
          }
        }
      }
    };
    // block: the MFMAs of sub-tile s_ into an, the epilogue steps of sub-tile s_ - 1 (from ap) spread evenly between them; sched_barrier
    // pins the order (left alone hipcc issues the MFMAs back to back and the epilogue behind them)
    auto block = [&](int s_, f32x16 (&an)[RF], f32x16 (&ap)[RF], float (&kp)[2][2], bool dma_between) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < PF; ++ks) frag_read(s_, ks);
      epi_begin(s_ - 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        if (i % RF == 0 && i / RF + PF < NKS) frag_read(s_, i / RF + PF);
        mfma_one(i, an);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = NSTEP * i / NM; t < NSTEP * (i + 1) / NM; ++t) epi_step(s_ - 1, ap, kp, t, dma_between, false);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto mfma_sub = [&](int s_, f32x16 (&ac)[RF]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < PF; ++ks) frag_read(s_, ks);
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        if (i % RF == 0 && i / RF + PF < NKS) frag_read(s_, i / RF + PF);
        mfma_one(i, ac);
      }
    };
    auto epi = [&](int s_, f32x16 (&ac)[RF], float (&kp)[2][2], bool dma_between, bool last) __attribute__((always_inline)) {
      epi_begin(s_);
#pragma unroll
      for (int t = 0; t < NSTEP; ++t) epi_step(s_, ac, kp, t, dma_between, last);
    };
    // the records of sub-tile s_: RF = 2 right behind its epilogue, RF = 1 behind the chunk boundary after it (partner's half from LDS)
    auto commit = [&](int s_, float (&kp)[2][2]) __attribute__((always_inline)) {
      if (STM == 1 && wave_ok && (RF == 2 || (wave & 1) == 0) && (l31 >> 1) == 8) {
        const int a = s_ % NA, par = (s_ / NA) & 1;
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2) {
          const int col = cbase + s_ * 32 + 16 * j2 + 8 * half;
          float* d = p.stats + (rec * p.stats_ld + (col >> 2) + (l31 & 1)) * 2;
          float o0 = 0.f, o1 = 0.f;
          if (RF == 1) {
            const float* o = sRec + ((((par * 4 + wave + 1) * (NA * 2) + a * 2 + j2) * 2 + half) * 4 + (l31 & 1) * 2);
            o0 = o[0];
            o1 = o[1];
          }
          d[0] = kp[j2][0] + o0;
          d[1] = kp[j2][1] + o1;
        }
      }
    };
    // entering chunk c >= 1: its weights have landed for every wave, every wave is past its fragment reads of chunk c - 1
    auto boundary = [&](int c) __attribute__((always_inline)) {
      if (c == 1) __builtin_amdgcn_s_waitcnt(wait_imm((NA - 1) * NU * VM_U));      // (chunk 0's first block has no epilogue)
      else __builtin_amdgcn_s_waitcnt(wait_imm(NA * NU * VM_U));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue((c + 1) & 1, c + 1 < nchunk ? c + 1 : nchunk - 1);      // (unconditional: the counts above assume it)
    };
    issue(1, nchunk > 1 ? 1 : 0);
    mfma_sub(0, acc[0]);
    for (int s1 = 1; s1 < NS; s1 += 2) {
      if (NA == 1) {
        boundary(s1);
        if (RF == 1 && s1 >= 2) commit(s1 - 2, keep[1]);
      }
      block(s1, acc[1], acc[0], keep[0], NA == 1);
      if (RF == 2) commit(s1 - 1, keep[0]);
      if (s1 + 1 < NS) {
        boundary((s1 + 1) / NA);
        if (RF == 1) commit(s1 - 1, keep[0]);
        block(s1 + 1, acc[0], acc[1], keep[1], true);
        if (RF == 2) commit(s1, keep[1]);
      }
    }
    // the last sub-tile's epilogue (its parity is uniform but not a constant)
    auto tail = [&](f32x16 (&ac)[RF], float (&kp)[2][2], float (&kq)[2][2]) __attribute__((always_inline)) {
      epi(NS - 1, ac, kp, false, true);
      if (RF == 2) commit(NS - 1, kp);
      if (RF == 1 && STM == 1) {                          // block-uniform
        __builtin_amdgcn_s_waitcnt(wait_imm(63));        // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        if (NS >= 2) commit(NS - 2, kq);
        commit(NS - 1, kp);
      }
    };
    if ((NS - 1) & 1) tail(acc[1], keep[1], keep[0]);
    else tail(acc[0], keep[0], keep[1]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the last boundary's weight DMA (a repeat of the last chunk) must not outlive the workgroup's LDS
  } else {
  for (int ci = 0; ci < nchunk; ++ci) {
    const int st = ci & 1;
    if (ci + 1 < nchunk) issue(st ^ 1, ci + 1);         // next chunk's weights land under this chunk's MFMAs
    u32x4 outv[RF][2];                                  // the LAST sub-tile's stores wait until after the barrier (see above)
    float srec[2][2];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int cb = ci * CC + a * 32;                   // first column of the sub-tile inside this block's range
      u32x4 rres[RF][2];
      if constexpr (HR) {
#pragma unroll
        for (int f = 0; f < RF; ++f)
#pragma unroll
          for (int j2 = 0; j2 < 2; ++j2)
            rres[f][j2] = *(const u32x4*)(p.R + (int64_t)(cbase + cb + 16 * j2) * 2 + roff[f]);
        // (the requests stay in FRONT of the MFMAs: without the branch that used to sit here the scheduler sinks them to their first use -
        // request, vmcnt(0), use - to save the registers: + 4 % on the ds1 out conv)
        __builtin_amdgcn_sched_barrier(0);
      }
      f32x16 acc[RF];
#pragma unroll
      for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
      const char* bW = sW + st * STAGE_B + (a * 32 + l31) * 128;
#pragma unroll
      for (int pl = 0; pl < KS; ++pl)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const u32x4 fw = *(const u32x4*)(bW + pl * PLANE_B + (((2 * c + half) ^ xsw) * 16));
#pragma unroll
          for (int f = 0; f < RF; ++f) Mma<__bf16>::run(fw, xa[f][4 * pl + c], acc[f]);
        }
      // acc[f][4 q + j] = channel 8 q + 4 half + j of row l31.  Pair q = 2 j2 (vdst) with q = 2 j2 + 1 (src): afterwards this lane
      // holds the 8 consecutive channels 16 j2 + 8 half .. + 8 of its row
#pragma unroll
      for (int j2 = 0; j2 < 2; ++j2) {
        const int col = cbase + cb + 16 * j2 + 8 * half;
        const f32x4 b0 = *(const f32x4*)(sBias + cb + 16 * j2 + 8 * half), b1 = *(const f32x4*)(sBias + cb + 16 * j2 + 8 * half + 4);
        float u[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] = 0.f;
#pragma unroll
        for (int f = 0; f < RF; ++f) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[f][8 * j2 + j]), __float_as_uint(acc[f][8 * j2 + 4 + j]),
                                                             false, false);
            v[j] = __uint_as_float(sw[0]);
            v[4 + j] = __uint_as_float(sw[1]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
          if constexpr (HR) {
            float rf[8];
            Elt<__bf16>::unpack(rres[f][j2], rf);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += rf[j];
          }
          const u32x4 pk = Elt<__bf16>::pack(v);
          if (DEFER && a == NA - 1) outv[f][j2] = pk;
          else *(u32x4*)(p.Y + (int64_t)(cbase + cb + 16 * j2) * 2 + yoff[f]) = pk;
          if (STM != 0) {                                // statistics of the values as STORED: the lane's 8 channels = two QUADS
            float rf[8];
            Elt<__bf16>::unpack(pk, rf);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              u[0] += rf[j];
              u[1] += rf[4 + j];
              u[2] += rf[j] * rf[j];
              u[3] += rf[4 + j] * rf[4 + j];
            }
          }
        }
        if (STM == 1) {                                   // block-uniform: quad records (sum, sum of squares) per 64 rows
          // lanes 16 / 17 of each half end up with the (sum, sum of squares) of quad 0 / 1 of the lane group's 8 channels
          const float t0 = halfwave_total(u[0]), t1 = halfwave_total(u[1]), t2 = halfwave_total(u[2]), t3 = halfwave_total(u[3]);
          const float msum = (l31 & 1) ? t1 : t0, msq = (l31 & 1) ? t3 : t2;
          if (RF == 1) {
            // a wave holds HALF a record (32 rows): park the partial for the even wave of the pair, which adds (own + partner) after
            // the chunk's barrier; the buffer alternates with the chunk parity, so a wave that runs ahead into the next chunk cannot
            // overwrite what its partner has not read yet
            if ((l31 >> 1) == 8) {
              float* d = sRec + ((((ci & 1) * 4 + wave) * (NA * 2) + a * 2 + j2) * 2 + half) * 4 + (l31 & 1) * 2;
              d[0] = msum;
              d[1] = msq;
            }
            srec[j2][0] = msum;                          // RF = 1 has one sub-tile per chunk (CC = 32)
            srec[j2][1] = msq;
          } else if (wave_ok && (l31 >> 1) == 8) {
            float* d = p.stats + (rec * p.stats_ld + (col >> 2) + (l31 & 1)) * 2;
            d[0] = msum;
            d[1] = msq;
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();                        // next chunk landed; every wave is past its reads of this stage
#pragma unroll
    for (int j2 = 0; DEFER && j2 < 2; ++j2) {
      const int col = cbase + ci * CC + (NA - 1) * 32 + 16 * j2 + 8 * half;
#pragma unroll
      for (int f = 0; f < RF; ++f)
        *(u32x4*)(p.Y + (int64_t)(cbase + ci * CC + (NA - 1) * 32 + 16 * j2) * 2 + yoff[f]) = outv[f][j2];
      if (RF == 1) {
        if (STM == 1 && wave_ok && (wave & 1) == 0 && (l31 >> 1) == 8) {
          const float* o = sRec + ((((ci & 1) * 4 + wave + 1) * (NA * 2) + (NA - 1) * 2 + j2) * 2 + half) * 4 + (l31 & 1) * 2;
          float* d = p.stats + (rec * p.stats_ld + (col >> 2) + (l31 & 1)) * 2;
          d[0] = srec[j2][0] + o[0];
          d[1] = srec[j2][1] + o[1];
        }
      }
    }
  }
  }
}

#define STRIP_LAUNCH_BOUNDS __launch_bounds__(256, (KS <= 2 ? (RF == 1 ? 4 : 3) : 2))
template <int KS, int RF, int CC, int GNM, int STM>
__global__ STRIP_LAUNCH_BOUNDS void conv1x1_strip_kernel(const ConvGemmParams p, const int nsplit) {
  conv1x1_strip_body<KS, RF, CC, GNM, STM, false>(p, nsplit);
}
// with a residual operand: v[248:255] are not the compiler's (STRIP_RES_LOAD / STRIP_RES_WAIT above)
template <int KS, int RF, int CC, int GNM, int STM>
__global__ STRIP_LAUNCH_BOUNDS __attribute__((amdgpu_num_vgpr(248))) void conv1x1_strip_res_kernel(const ConvGemmParams p, const int nsplit) {
  conv1x1_strip_body<KS, RF, CC, GNM, STM, true>(p, nsplit);
}

template <int KS, int RF, int CC, int GNM, int STM, bool HR>
static int launch_conv1x1_strip_res(const ConvGemmParams& p, hipStream_t st) {
  constexpr int BR = 128 * RF, STAGE_B = KS * CC * 128;
  const int rowblocks = cdiv(p.M, BR), nch = p.Cout / CC;
  // column split: the smallest divisor of the chunk count that gives the chip ONE block per CU (the strip's rows are then loaded nsplit
  // times, from L2 after the first); results do not depend on it.  256 since round 5 (448 = ~1.75 blocks per CU before): a launch of the
  // small levels that fills BOTH block slots of every CU leaves no room for the other launch chain's blocks, and the two chains' small,
  // latency-bound launches then queue behind each other instead of running side by side - same-call A/B, two boxes, two passes each:
  // 11.14 -> 10.80 ms and 11.18 -> 10.94 ms per step; 224: the same, 192: 10.96, 128: 11.40, 288 / 320 (1.5 blocks per CU at ds4): 11.3 / 11.0
  // (profiles/r05_launch_width_ab.txt)
  static const int want_blocks = [] {                     // tuning switch (read once): MMD_STRIP_BLOCKS, default 256
    const char* e = getenv("MMD_STRIP_BLOCKS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 256;
  }();
  int nsplit = 1;
  for (int d = 1; d <= nch && d <= 16; ++d)
    if (nch % d == 0) {
      nsplit = d;
      if ((int64_t)rowblocks * d >= want_blocks) break;
    }
  const int Cs = p.Cout / nsplit;
  constexpr size_t REC_B = RF == 1 ? 2 * 4 * (CC / 32) * 2 * 32 * sizeof(float) : 0;
  const size_t lds = 2 * (size_t)STAGE_B + (size_t)((Cs + 3) & ~3) * 4 + (p.gn_a ? 4 * (size_t)(64 * KS) * 4 : 0) + REC_B;
  const size_t lds_max = 2 * (size_t)STAGE_B + 2048 * 4 + 4 * (size_t)(64 * KS) * 4 + REC_B;
  if (lds > lds_max) return mmd_set_error(MMD_ERR_UNSUPPORTED, "conv_gemm tile 131 (strip): %d output channels per block", Cs);
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    const void* kfn = HR ? (const void*)conv1x1_strip_res_kernel<KS, RF, CC, GNM, STM> : (const void*)conv1x1_strip_kernel<KS, RF, CC, GNM, STM>;
    hipError_t e = hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "conv1x1_strip: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  if (HR) hipLaunchKernelGGL((conv1x1_strip_res_kernel<KS, RF, CC, GNM, STM>), dim3(rowblocks * nsplit), dim3(256), lds, st, p, nsplit);
  else hipLaunchKernelGGL((conv1x1_strip_kernel<KS, RF, CC, GNM, STM>), dim3(rowblocks * nsplit), dim3(256), lds, st, p, nsplit);
  return mmd_check_launch("conv1x1_strip");
}

template <int KS, int RF, int CC, int GNM, int STM>
static int launch_conv1x1_strip_mode(const ConvGemmParams& p, hipStream_t st) {
  return p.R ? launch_conv1x1_strip_res<KS, RF, CC, GNM, STM, true>(p, st) : launch_conv1x1_strip_res<KS, RF, CC, GNM, STM, false>(p, st);
}

template <int KS, int RF, int CC, int STM>
static int launch_conv1x1_strip_st(const ConvGemmParams& p, hipStream_t st) {
  if (!p.gn_a) return launch_conv1x1_strip_mode<KS, RF, CC, 0, STM>(p, st);
  return p.gn_act ? launch_conv1x1_strip_mode<KS, RF, CC, 2, STM>(p, st) : launch_conv1x1_strip_mode<KS, RF, CC, 1, STM>(p, st);
}

template <int KS, int RF, int CC>
static int launch_conv1x1_strip(const ConvGemmParams& p, hipStream_t st) {
  return p.stats ? launch_conv1x1_strip_st<KS, RF, CC, 1>(p, st) : launch_conv1x1_strip_st<KS, RF, CC, 0>(p, st);
}

// tile 131: bf16 convs whose whole K = ntaps * Cin is 128 / 256 (two row fragments per wave) or 384 / 512 (one fragment): the 1x1
// convs at 128-512 channels and the k=3 temporal / audio convs at 128 channels.  Cin % 64 == 0; fused GroupNorm only with one tap
static int dispatch_conv1x1_strip(const ConvGemmParams& p, hipStream_t st) {
  const int K = p.Cin * p.ntaps;
  const int rf = K <= 256 ? 2 : 1, cc = K <= 256 ? 64 : 32;
  if ((K != 128 && K != 256 && K != 384 && K != 512) || p.Cin % 64 != 0 || p.Cout % cc != 0 ||
      (int64_t)p.M * p.ldy * 2 >= 0xffffffffLL || (p.R && (int64_t)p.M * p.ldr * 2 >= 0xffffffffLL) || (p.gn_a && (p.ntaps != 1 || p.gn_rows < 128 * rf)) ||
      (p.ntaps == 1 && (p.taps[0] || p.taps[1] || p.taps[2])))
    return mmd_set_error(MMD_ERR_UNSUPPORTED, "conv_gemm tile 131 (strip): needs ntaps * Cin in {128, 256, 384, 512}, Cin %% 64 == 0, Cout %% %d == 0, "
                         "fused GroupNorm only for 1x1 convs with slices of >= %d rows (got Cin=%d ntaps=%d Cout=%d)", cc, 128 * rf, p.Cin, p.ntaps, p.Cout);
  if (K == 128) {
    // round 6: the ResBlock out conv of the ds1 levels (norm + SiLU + 1x1 + skip, 65536 / 25600 rows per sample) on 128-row blocks, four
    // per CU, instead of 256-row blocks, three per CU - 1024 blocks on 768 slots ran 1.33 rounds.  Chosen by the layer's rows per
    // SAMPLE (the record fold of a one-fragment wave differs in the last bit: the choice must not move with the batch size).
    // Same-call A/B (profiles/r06_lanes_width_aconv_call11.txt): the graded ResBlock 0.2042 -> 0.1990 ms, the step unchanged (10.85 / 10.84 ms).
    // MMD_STRIP_K128_RF1=0: the two-fragment instance (A/B).
    static const bool rf1 = [] { const char* e = getenv("MMD_STRIP_K128_RF1"); return !(e && e[0] == '0'); }();
    if (rf1 && p.gn_a && p.gn_rows >= 16384 && p.ntaps == 1 && p.Cout % 32 == 0) return launch_conv1x1_strip<2, 1, 32>(p, st);
    return launch_conv1x1_strip<2, 2, 64>(p, st);
  }
  if (K == 256) return launch_conv1x1_strip<4, 2, 64>(p, st);
  if (K == 384) return launch_conv1x1_strip<6, 1, 32>(p, st);
  return launch_conv1x1_strip<8, 1, 32>(p, st);
}

template <typename T, bool GN>
static int launch_conv_gemm_halo(const ConvGemmParams& p, hipStream_t st) {
  bool taps_ok = p.ntaps <= 9;
  for (int t = 0; t < p.ntaps && taps_ok; ++t)
    taps_ok = p.taps[t * 3] == 0 && p.taps[t * 3 + 1] >= -1 && p.taps[t * 3 + 1] <= 1 && p.taps[t * 3 + 2] >= -1 && p.taps[t * 3 + 2] <= 1;
  if (!taps_ok || p.D1 % 8 != 0 || p.D2 % 16 != 0 || (int64_t)p.D0 * p.D1 * p.D2 != p.M || p.Cin % (8 * Elt<T>::EPV) != 0)
    return mmd_set_error(MMD_ERR_UNSUPPORTED, "conv_gemm tile 130 (halo): needs spatial taps (|dh|,|dw| <= 1), D1 %% 8 == 0, D2 %% 16 == 0, "
                         "full frames and Cin a multiple of one 128-byte K step");
  if (GN && (p.ntaps != 9 || p.gn_rows % ((int64_t)p.D1 * p.D2) != 0))
    return mmd_set_error(MMD_ERR_UNSUPPORTED, "conv_gemm tile 130 with fused GroupNorm: needs the nine spatial taps and slices of whole frames");
  const size_t lds = 2 * (size_t)(23 * 8 * 128) + 2 * (size_t)(128 * 128) + 336 + (GN ? 1024 : 0);     // (two blocks per CU: <= 81920 B)
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_gemm_halo_kernel<T, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "conv_gemm_halo: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int grid = (p.M / 128) * cdiv(p.Cout, 128);
  hipLaunchKernelGGL((conv_gemm_halo_kernel<T, GN>), dim3(grid), dim3(256), lds, st, p);
  return mmd_check_launch("conv_gemm_halo");
}

// tile 133: halo-tile main loop on 16 x 16 patches, 8 waves, three-slot weight ring (bf16, nine spatial taps)
template <bool GN>
static int launch_conv_gemm_halo16(const ConvGemmParams& p, hipStream_t st) {
  bool taps_ok = p.ntaps == 9;
  for (int t = 0; t < p.ntaps && taps_ok; ++t)       // canonical order (dh, dw) row-major: the kernel's tap shifts are literals
    taps_ok = p.taps[t * 3] == 0 && p.taps[t * 3 + 1] == t / 3 - 1 && p.taps[t * 3 + 2] == t % 3 - 1;
  if (!taps_ok || p.D1 % 16 != 0 || p.D2 % 16 != 0 || (int64_t)p.D0 * p.D1 * p.D2 != p.M || p.Cin % 64 != 0 ||
      ((int64_t)p.D1 * p.D2 - 1) * p.lda * 2 + (int64_t)p.Cin * 2 >= 0x7fffffffLL || (int64_t)128 * p.Cin * 9 * 2 >= 0x7fffffffLL)
    return mmd_set_error(MMD_ERR_UNSUPPORTED, "conv_gemm tile 133 (halo, 16 x 16 patches): needs the nine spatial taps in row-major (dh, dw) order, "
                         "D1 %% 16 == 0, D2 %% 16 == 0, full frames, Cin %% 64 == 0 and frames below 2 GB");
  if (GN && p.gn_rows % ((int64_t)p.D1 * p.D2) != 0)
    return mmd_set_error(MMD_ERR_UNSUPPORTED, "conv_gemm tile 133 with fused GroupNorm: needs slices of whole frames");
  constexpr size_t OPS_B = 2 * (size_t)(41 * 8 * 128) + 3 * (size_t)(128 * 128), C_B = 256 * 132 * sizeof(float);
  const size_t lds = (OPS_B > C_B ? OPS_B : C_B) + 1024 + 256;
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_gemm_halo16_kernel<GN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "conv_gemm_halo16: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int grid = (p.M / 256) * cdiv(p.Cout, 128);
  hipLaunchKernelGGL((conv_gemm_halo16_kernel<GN>), dim3(grid), dim3(512), lds, st, p);
  return mmd_check_launch("conv_gemm_halo16");
}

// descriptor addressing of the direct-to-LDS loops: 32-bit byte offsets into the activation view and the block's 128 weight rows;
// MMD_GEMM_DESC=0 keeps the 64-bit pointer path (A/B)
template <typename T>
static bool glds_desc_ok(const ConvGemmParams& p) {
  static const bool on = [] { const char* e = getenv("MMD_GEMM_DESC"); return !(e && e[0] == '0'); }();
  constexpr int ES = 16 / Elt<T>::EPV;
  int maxshift = 0;
  for (int t = 0; t < p.ntaps; ++t) {
    const int64_t sh = ((int64_t)p.taps[3 * t] * p.D1 * p.D2 + (int64_t)p.taps[3 * t + 1] * p.D2 + p.taps[3 * t + 2]) * p.lda * ES;
    if (sh > 0x3fffffff || sh < -0x3fffffff) return false;
    (void)maxshift;
  }
  return on && ((int64_t)p.M * p.lda + p.Cin) * ES < 0x7fffffffLL && (int64_t)128 * p.Cin * p.ntaps * ES < 0x7fffffffLL;
}

template <typename T>
static int launch_conv_gemm_glds(const ConvGemmParams& p, hipStream_t st) {
  const size_t lds = 128 * 132 * sizeof(float) + 336;
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_gemm_glds_kernel<T, false, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_gemm_glds_kernel<T, true, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_gemm_glds_kernel<T, true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "conv_gemm_glds: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int grid = cdiv(p.M, 128) * cdiv(p.Cout, 128);
  if (p.Cin % (8 * Elt<T>::EPV) != 0) hipLaunchKernelGGL((conv_gemm_glds_kernel<T, false, 2, false>), dim3(grid), dim3(256), lds, st, p);
  else if (glds_desc_ok<T>(p)) hipLaunchKernelGGL((conv_gemm_glds_kernel<T, true, 2, true>), dim3(grid), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((conv_gemm_glds_kernel<T, true, 2, false>), dim3(grid), dim3(256), lds, st, p);
  return mmd_check_launch("conv_gemm_glds");
}

// tile 132: the direct-to-LDS loop with a four-slot ring (three K steps of DMA in flight), one block per CU
template <typename T>
static int launch_conv_gemm_ring(const ConvGemmParams& p, hipStream_t st) {
  if (p.Cin % (8 * Elt<T>::EPV) != 0)
    return mmd_set_error(MMD_ERR_UNSUPPORTED, "conv_gemm tile 132 (deep ring): Cin must be a multiple of one 128-byte K step");
  const size_t lds = 8 * (size_t)(128 * 128) + 336;
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_gemm_glds_kernel<T, true, 4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_gemm_glds_kernel<T, true, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "conv_gemm_ring: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int grid = cdiv(p.M, 128) * cdiv(p.Cout, 128);
  if (glds_desc_ok<T>(p)) hipLaunchKernelGGL((conv_gemm_glds_kernel<T, true, 4, true>), dim3(grid), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((conv_gemm_glds_kernel<T, true, 4, false>), dim3(grid), dim3(256), lds, st, p);
  return mmd_check_launch("conv_gemm_ring");
}

template <typename T, int BM, int BN, bool GN>
static int launch_conv_gemm(const ConvGemmParams& p, hipStream_t st) {
  const size_t lds_ops = 2 * (size_t)(BM + BN) * ROWB;
  const size_t lds_c = (size_t)BM * (BN + 4) * sizeof(float);
  const size_t lds = (lds_ops > lds_c ? lds_ops : lds_c) + 336 + (GN ? 16 * 256 : 0);   // + tap table (+ GN affine cache, Cin <= 256)
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_gemm_kernel<T, BM, BN, GN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "conv_gemm: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int grid = cdiv(p.M, BM) * cdiv(p.Cout, BN);
  hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, GN>), dim3(grid), dim3(256), lds, st, p);
  return mmd_check_launch("conv_gemm");
}

template <typename T>
static int dispatch_conv_gemm(const ConvGemmParams& p, int tile, hipStream_t st) {
  if (tile == 133) {
    if constexpr (Elt<T>::EPV == 8) return p.gn_a ? launch_conv_gemm_halo16<true>(p, st) : launch_conv_gemm_halo16<false>(p, st);
    else return mmd_set_error(MMD_ERR_UNSUPPORTED, "conv_gemm tile 133: bf16 only");
  }
  if (p.gn_a) {
    if (tile == 130) {
      if constexpr (Elt<T>::EPV == 8) return launch_conv_gemm_halo<T, true>(p, st);
      else return mmd_set_error(MMD_ERR_UNSUPPORTED, "conv_gemm tile 130 with fused GroupNorm: bf16 only");
    }
    if (tile == 128) return launch_conv_gemm<T, 128, 128, true>(p, st);
    return launch_conv_gemm<T, 64, 64, true>(p, st);
  }
  if (tile == 129) return launch_conv_gemm_glds<T>(p, st);
  if (tile == 132) return launch_conv_gemm_ring<T>(p, st);
  if (tile == 130) return launch_conv_gemm_halo<T, false>(p, st);
  if (tile == 128) return launch_conv_gemm<T, 128, 128, false>(p, st);
  return launch_conv_gemm<T, 64, 64, false>(p, st);
}

static int conv_gemm_impl(int dtype, const void* A, int64_t lda, const void* W, const float* bias, const void* R, int64_t ldr,
                          void* Y, int64_t ldy, int M, int Cout, int Cin, int ntaps, const int* taps, int D0, int D1, int D2,
                          int tile, const float* gn_a, const float* gn_b, int gn_act, int gn_S, int64_t gn_rows, float* stats,
                          int64_t stats_ld, void* stream) {
  const int epv = dtype == MMD_BF16 ? 8 : 4;
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "conv_gemm: bad dtype %d", dtype);
  MMD_REQUIRE(A && W && Y && M > 0 && Cout > 0 && Cin > 0, "conv_gemm: null/empty argument");
  MMD_REQUIRE(ntaps >= 1 && ntaps <= 27 && taps, "conv_gemm: ntaps %d out of [1,27]", ntaps);
  MMD_REQUIRE(Cin % epv == 0, "conv_gemm: Cin %d must be a multiple of %d", Cin, epv);
  MMD_REQUIRE(Cout % 8 == 0, "conv_gemm: Cout %d must be a multiple of 8", Cout);
  MMD_REQUIRE(lda % epv == 0 && ldy % epv == 0 && (!R || ldr % epv == 0), "conv_gemm: row strides must be 16-byte multiples");
  MMD_REQUIRE(((uintptr_t)A | (uintptr_t)W | (uintptr_t)Y | (uintptr_t)R) % 16 == 0, "conv_gemm: pointers must be 16-byte aligned");
  MMD_REQUIRE(D0 > 0 && D1 > 0 && D2 > 0, "conv_gemm: bad position dims");
  MMD_REQUIRE(!gn_a || (gn_b && (ntaps == 1 || tile == 130 || tile == 133) && gn_S > 0 && gn_rows >= 128 && (Cin <= 256 || tile == 131 || tile == 130 || tile == 133) &&
                        (int64_t)gn_S * gn_rows == M),
              "gn_conv1x1 / gn_conv_gemm: needs contiguous slices of >= 128 rows covering M, Cin <= 256 unless tile 130 / 131, taps only with tile 130 "
              "(got S=%d rows=%ld Cin=%d M=%d ntaps=%d tile=%d)", gn_S, (long)gn_rows, Cin, M, ntaps, tile);
  ConvGemmParams p;
  p.A = (const char*)A; p.lda = lda; p.W = (const char*)W; p.bias = bias;
  p.R = (const char*)R; p.ldr = ldr; p.Y = (char*)Y; p.ldy = ldy;
  p.M = M; p.Cout = Cout; p.Cin = Cin; p.ntaps = ntaps; p.D0 = D0; p.D1 = D1; p.D2 = D2;
  p.gn_a = gn_a; p.gn_b = gn_b; p.gn_act = gn_act; p.gn_S = gn_S; p.gn_rows = gn_rows;
  MMD_REQUIRE(!stats || (M % 64 == 0 && Cout % 4 == 0 && stats_ld >= Cout / 4 && tile != 130 && tile != 133 && (uintptr_t)stats % 8 == 0),
              "conv_gemm: output statistics need M %% 64 == 0, Cout %% 4 == 0, stats_ld >= Cout / 4 (quads) and a row-tiled main loop (not tiles 130 / 133)");
  p.stats = stats; p.stats_ld = stats_ld;
  for (int i = 0; i < ntaps * 3; ++i) p.taps[i] = taps[i];
  hipStream_t st = (hipStream_t)stream;
  if (tile == 0) tile = (int64_t)cdiv(M, 128) * cdiv(Cout, 128) >= 320 ? 128 : 64;
  MMD_REQUIRE(tile == 64 || tile == 128 || ((tile == 129 || tile == 132) && !gn_a) || (tile == 130 && (!gn_a || dtype == MMD_BF16)) ||
                  ((tile == 131 || tile == 133) && dtype == MMD_BF16),
              "conv_gemm: tile must be 0, 64, 128, 129 (128 direct-to-LDS), 130 (halo-tile 3x3), 131 (row strip, bf16 1x1 convs) or 132 (129 with a "
              "four-slot ring for launches of few tiles)");
  if (tile == 131) return dispatch_conv1x1_strip(p, st);
  return dtype == MMD_BF16 ? dispatch_conv_gemm<__bf16>(p, tile, st) : dispatch_conv_gemm<float>(p, tile, st);
}

extern "C" int mmd_conv_gemm(int dtype, const void* A, int64_t lda, const void* W, const float* bias, const void* R, int64_t ldr,
                             void* Y, int64_t ldy, int M, int Cout, int Cin, int ntaps, const int* taps, int D0, int D1, int D2,
                             int tile, void* stream) {
  return conv_gemm_impl(dtype, A, lda, W, bias, R, ldr, Y, ldy, M, Cout, Cin, ntaps, taps, D0, D1, D2, tile, nullptr, nullptr, 0, 0,
                        0, nullptr, 0, stream);
}

// As mmd_conv_gemm, and the epilogue also leaves the GroupNorm statistics of the output for its consumer: per (64-row record, QUAD
// of 4 columns) the sum and the sum of squares of the values as stored, stats[(m / 64) * stats_ld + quad] = float2 (stats points at
// the first quad this launch writes, so producers of a channel-concatenated tensor fill column slices of one record buffer).
// mmd_gn_finalize_stats turns the records into the fused affine; the statistics pass over the tensor disappears.
extern "C" int mmd_conv_gemm_stats(int dtype, const void* A, int64_t lda, const void* W, const float* bias, const void* R, int64_t ldr,
                                   void* Y, int64_t ldy, int M, int Cout, int Cin, int ntaps, const int* taps, int D0, int D1, int D2,
                                   int tile, float* stats, int64_t stats_ld, void* stream) {
  MMD_REQUIRE(stats, "conv_gemm_stats: null statistics buffer");
  return conv_gemm_impl(dtype, A, lda, W, bias, R, ldr, Y, ldy, M, Cout, Cin, ntaps, taps, D0, D1, D2, tile, nullptr, nullptr, 0, 0,
                        0, stats, stats_ld, stream);
}

// 1x1 conv of GroupNorm32(+FiLM)(+SiLU)'d rows: Y = act(A * gn_a[s(m)] + gn_b[s(m)]) W^T + bias (+ R); gn_a/gn_b [S, Cin] from
// mmd_gn_stats over S contiguous slices of `rows_per_slice` rows (s(m) = m / rows_per_slice).
extern "C" int mmd_gn_conv1x1(int dtype, const void* A, int64_t lda, const float* gn_a, const float* gn_b, int act, int S,
                              int64_t rows_per_slice, const void* W, const float* bias, const void* R, int64_t ldr, void* Y,
                              int64_t ldy, int M, int Cout, int Cin, int tile, void* stream) {
  static const int tap0[3] = {0, 0, 0};
  MMD_REQUIRE(gn_a && gn_b, "gn_conv1x1: null GroupNorm affine");
  return conv_gemm_impl(dtype, A, lda, W, bias, R, ldr, Y, ldy, M, Cout, Cin, 1, tap0, 1, 1, 1, tile, gn_a, gn_b, act, S,
                        rows_per_slice, nullptr, 0, stream);
}

// mmd_gn_conv1x1 that also emits the output statistics (see mmd_conv_gemm_stats).
extern "C" int mmd_gn_conv1x1_stats(int dtype, const void* A, int64_t lda, const float* gn_a, const float* gn_b, int act, int S,
                                    int64_t rows_per_slice, const void* W, const float* bias, const void* R, int64_t ldr, void* Y,
                                    int64_t ldy, int M, int Cout, int Cin, int tile, float* stats, int64_t stats_ld, void* stream) {
  static const int tap0[3] = {0, 0, 0};
  MMD_REQUIRE(gn_a && gn_b && stats, "gn_conv1x1_stats: null GroupNorm affine / statistics buffer");
  return conv_gemm_impl(dtype, A, lda, W, bias, R, ldr, Y, ldy, M, Cout, Cin, 1, tap0, 1, 1, 1, tile, gn_a, gn_b, act, S,
                        rows_per_slice, stats, stats_ld, stream);
}

// Spatial 3x3 conv of GroupNorm32(+FiLM)(+SiLU)'d rows on the halo tile (tile 130, bf16): the normalisation is applied to the staged
// halo in LDS, so the normalised tensor never exists in HBM (mmd_gn_apply + mmd_conv_gemm in one launch; bitwise equal to the pair).
extern "C" int mmd_gn_conv_gemm(int dtype, const void* A, int64_t lda, const float* gn_a, const float* gn_b, int act, int S,
                                int64_t rows_per_slice, const void* W, const float* bias, const void* R, int64_t ldr, void* Y,
                                int64_t ldy, int M, int Cout, int Cin, int ntaps, const int* taps, int D0, int D1, int D2, int tile,
                                void* stream) {
  MMD_REQUIRE(gn_a && gn_b, "gn_conv_gemm: null GroupNorm affine");
  MMD_REQUIRE(tile == 130 || tile == 133, "gn_conv_gemm: the fused input GroupNorm of a conv with taps exists on the halo tiles (130 / 133) only");
  return conv_gemm_impl(dtype, A, lda, W, bias, R, ldr, Y, ldy, M, Cout, Cin, ntaps, taps, D0, D1, D2, tile, gn_a, gn_b, act, S,
                        rows_per_slice, nullptr, 0, stream);
}

