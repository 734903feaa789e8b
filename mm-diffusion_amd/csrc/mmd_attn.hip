// Attention kernels on channels-last qkv rows.
//
// One problem description covers every attention of the reference hot path:
//   * SingleModalQKVAttention spatial / audio self-attention   (multimodal_unet.py:212-240)
//   * QKVAttention = random-shift windowed cross-modal attention (RS-MMA), both directions
//     (multimodal_unet.py:507-564) with the window addressed arithmetically
//     keys(group i) = rows k_batch + ((i + shift) * k_per_group + j) mod k_mod, j < win * k_per_group
//     (the reference materialises [F*HW x win*L/F] int64 index matrices, unet:614-647, to read F rows)
//   * the last query group owns the remainder rows (unet:547-548)
// q/k/v live in the qkv GEMM output rows: q at column q_off + h*ch, k at k_off + h*ch, v at v_off + h*ch.
//
// attn_mfma_kernel (bf16, ch in {16,32,48,64,96,128}): flash-style, 4 waves x 32 queries, 64-key tiles.
//   S^T = K Q^T on v_mfma_f32_32x32x16_bf16 (keys = D rows, queries = D cols: every lane owns ONE query,
//   so running max / sum / rescale are lane-local plus one lane^32 exchange), P stays in registers and is
//   the B operand of O^T = V^T P^T (the 32x32 C layout of S^T is exactly the k-slot order we feed, with V^T
//   staged key-permuted to match), K row-major / V transposed in LDS with conflict-free strides.
// attn_generic_kernel (fp32 math, any ch <= 192): LDS-tiled VALU flash attention - fp32 mode and odd shapes.
// attn_small_kernel: one wave per (slice, head) for short sequences (temporal attention, T = F <= 32).
#include "mmd_common.h"
#include <type_traits>

struct AttnParams {
  const char* Q; int64_t ldq;
  const char* KV; int64_t ldkv;
  char* O; int64_t ldo;
  int q_off, k_off, v_off;
  int heads, ch;
  int nb, G;
  int64_t q_rows_per_batch;
  int q_per_group;
  int64_t k_rows_per_batch;   // = k_mod
  int k_per_group, win;
  const int* shift_ptr;
  float scale;                // ch^-1/2  (= (ch^-1/4)^2, reference scales q and k separately)
  float* lse2;                // optional [q rows, heads]: log2-domain log-sum-exp of scale*log2e*scores (training forward)
};

struct GroupInfo {
  int64_t q_row0, k_row0;
  int q_count, k_count, k_start;
  int k_mod;
};

__device__ __forceinline__ GroupInfo group_info(const AttnParams& p, int bg) {
  GroupInfo gi;
  const int n = bg / p.G, g = bg % p.G;
  gi.q_row0 = (int64_t)n * p.q_rows_per_batch + (int64_t)g * p.q_per_group;
  gi.q_count = (g == p.G - 1) ? (int)(p.q_rows_per_batch - (int64_t)g * p.q_per_group) : p.q_per_group;
  gi.k_row0 = (int64_t)n * p.k_rows_per_batch;
  gi.k_mod = (int)p.k_rows_per_batch;
  gi.k_count = p.win * p.k_per_group;
  const int shift = p.shift_ptr ? *p.shift_ptr : 0;
  gi.k_start = (int)(((int64_t)(g + shift) * p.k_per_group) % gi.k_mod);
  return gi;
}
// XCD-aware block remap.  Workgroups are dealt round-robin to the 8 XCDs by flat id, and blockIdx.x (the query tile) is
// the fastest index: by default the query tiles of one (head, group) land on 8 DIFFERENT XCDs and each private L2 fetches
// the same K/V window from HBM again.  Remap so all query tiles of a (head, group) share flat-id mod 8 (same XCD, and
// adjacent in dispatch order).  Returns (query tile, head, batch-group).
__device__ __forceinline__ void attn_block_coords(int& qt, int& h, int& bg) {
  const int nx = gridDim.x, ny = gridDim.y, nz = gridDim.z;
  const int hz_count = ny * nz;
  int hz;
  if ((hz_count & 7) == 0 && nx > 1) {
    const int f = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int k = f >> 3, r = f & 7;
    qt = k % nx;
    hz = r + 8 * (k / nx);
  } else {
    qt = blockIdx.x;
    hz = blockIdx.y + ny * blockIdx.z;
  }
  h = hz % ny;
  bg = hz / ny;
}

__device__ __forceinline__ int64_t key_row(const GroupInfo& gi, int kk) {
  int r = gi.k_start + kk;
  if (r >= gi.k_mod) r -= gi.k_mod;
  return gi.k_row0 + r;
}

// ============================================================================= MFMA flash attention (bf16)
__device__ __attribute__((aligned(16))) uint32_t g_attn_zero[4] = {0, 0, 0, 0};   // (-DATTN_BRANCHFREE_KV) what a key past the window reads
#define SVT_STRIDE 136   // bytes per V^T row (64 keys * 2 B + 8): (stride/8) odd -> conflict-free ds_read_b64

// launch bound 2 waves/SIMD (<= 256 registers): keeps the S / O accumulators in the unified VGPR file - with the default
// bound hipcc parks them in AGPRs and every softmax / rescale touch costs a v_accvgpr_read + write pair (224 moves per tile).
template <int D>
__global__ __launch_bounds__(256, (D <= 96 ? 2 : 1)) void attn_mfma_kernel(const AttnParams p) {
  constexpr int DV = D / 8;            // 16-byte vecs per row
  constexpr int SK = D * 2 + 16;       // bytes per K row in LDS ((SK/16) odd)
  constexpr int KST = D / 16;          // k-steps of the S MFMA
  constexpr int DT = (D + 31) / 32;    // 32-wide d tiles of O^T
  constexpr int NK = (64 * DV + 255) / 256;   // staged vecs per thread per operand
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                     // [64][SK]
  char* sVt = smem + 64 * SK;          // [DT*32][SVT_STRIDE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  int qt, h, bg;
  attn_block_coords(qt, h, bg);
  const GroupInfo gi = group_info(p, bg);
  const int q0 = qt * 128;
  if (q0 >= gi.q_count) return;        // uniform per block

  // zero the V^T rows beyond D once (D=16/48: upper half of a 32-row tile is never staged)
  if (D % 32 != 0) {
    for (int i = tid; i < (DT * 32 - D) * SVT_STRIDE / 4; i += 256) ((uint32_t*)(sVt + D * SVT_STRIDE))[i] = 0u;
  }

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q = l31, half) holds d = 16*s + 8*half + [0,8)
  const int qi = q0 + wave * 32 + l31;
  const bool qok = qi < gi.q_count;
  u32x4 qf[KST];
  {
    const char* qp = p.Q + ((gi.q_row0 + qi) * p.ldq + p.q_off + h * D) * 2;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qok) v = *(const u32x4*)(qp + (s * 16 + half * 8) * 2);
      qf[s] = v;
    }
  }

  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float sc = p.scale * 1.4426950408889634f;   // scores kept in log2 domain

  u32x4 rk[NK], rv[NK];
  const int ntiles = (gi.k_count + 63) >> 6;

  auto load_kv = [&](int kt0) {
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int id = tid + 256 * i;
      // K: row-major, DV lanes per key row (coalesced)
      {
        const int j = id / DV, v = id % DV;
        u32x4 x = {0u, 0u, 0u, 0u};
        if (id < 64 * DV && kt0 + j < gi.k_count)
          x = *(const u32x4*)(p.KV + (key_row(gi, kt0 + j) * p.ldkv + p.k_off + h * D + v * 8) * 2);
        rk[i] = x;
      }
      // V: 32 keys x 2 d-vecs per wave instruction (conflict-free transposed ds_write_b16)
      {
        const int j = (id & 31) + 32 * ((id >> 6) & 1);
        const int v = 2 * (id >> 7) + ((id >> 5) & 1);
        u32x4 x = {0u, 0u, 0u, 0u};
        if (id < 64 * DV && kt0 + j < gi.k_count)
          x = *(const u32x4*)(p.KV + (key_row(gi, kt0 + j) * p.ldkv + p.v_off + h * D + v * 8) * 2);
        rv[i] = x;
      }
    }
  };
  auto store_kv = [&]() {
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int id = tid + 256 * i;
      if (id < 64 * DV) {
        {
          const int j = id / DV, v = id % DV;
          *(u32x4*)(sK + j * SK + v * 16) = rk[i];
        }
        {
          const int j = (id & 31) + 32 * ((id >> 6) & 1);
          const int v = 2 * (id >> 7) + ((id >> 5) & 1);
          uint16_t* dst = (uint16_t*)(sVt + (8 * v) * SVT_STRIDE + 2 * j);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            dst[e * (SVT_STRIDE / 2)] = (uint16_t)((rv[i][e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        }
      }
    }
  };

  load_kv(0);
  // full 64-key tiles never need the key mask: the body is instantiated twice so the (wave-uniform) ragged last tile
  // does not leave 68 v_cmp/v_cndmask per tile in the hot loop (hipcc if-converts a plain `if`)
  auto tile_body = [&](int t, auto ragged) {
    __syncthreads();          // previous tile fully consumed
    store_kv();
    __syncthreads();
    if (t + 1 < ntiles) load_kv((t + 1) * 64);   // in flight during the MFMAs below

    // ---- S^T = K Q^T : two 32-key sub-tiles
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
    for (int st = 0; st < KST; ++st)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {      // two independent accumulator chains interleaved
        const u32x4 kf = *(const u32x4*)(sK + (kt * 32 + l31) * SK + half * 16 + st * 32);
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[st]), s[kt], 0, 0, 0);
      }
    // ---- online softmax (lane-local row; partner lane^32 holds the other 32 keys).  VALU diet: the key mask is applied only
    //      in the (wave-uniform) ragged last tile, the softmax scale rides in the exp2 fma, and the O rescale is skipped
    //      when no lane's running max moved (alpha == 1 for the whole wave).
    if constexpr (decltype(ragged)::value) {
      const int kbase = t * 64 + 4 * half;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = kbase + 32 * kt + (r & 3) + 8 * (r >> 2);
          s[kt][r] = kk < gi.k_count ? s[kt][r] : -3e38f;
        }
    }
    float mx = -3e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * sc);        // sc > 0: max commutes with the scale
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], sc, -m_new));
        s[kt][r] = e;
        ps += e;
      }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    if (__any(m_new != m_run)) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    m_run = m_new;
    // ---- O^T += V^T P^T : P fragments straight from the S^T registers (k-slot j <-> reg 8*st + j)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)s[kt][8 * st + j];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const char* vb = sVt + (dt * 32 + l31) * SVT_STRIDE + (32 * kt + 16 * st + 4 * half) * 2;
          const u32x2 v0 = *(const u32x2*)(vb);
          const u32x2 v1 = *(const u32x2*)(vb + 16);
          u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf, o[dt], 0, 0, 0);
        }
      }
  };
  const int nfull = gi.k_count >> 6;
  for (int t = 0; t < nfull; ++t) tile_body(t, std::false_type{});
  if (nfull < ntiles) tile_body(nfull, std::true_type{});
  // ---- normalise and store: lane owns query qi, d = 32*dt + (r&3) + 8*(r>>2) + 4*half
  if (qok) {
    const float inv = 1.f / l_run;
    if (p.lse2 && half == 0) p.lse2[(gi.q_row0 + qi) * p.heads + h] = m_run + __builtin_amdgcn_logf(l_run);   // v_log_f32 = log2
    char* op = p.O + ((gi.q_row0 + qi) * p.ldo + h * D) * 2;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = dt * 32 + 8 * q4 + 4 * half;
        if (d < D) {
          bf16x4 w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = (__bf16)(o[dt][4 * q4 + j] * inv);
          *(bf16x4*)(op + d * 2) = w;
        }
      }
  }
}

// max / sum over the two lanes (l, l ^ 32) that share a query: v_permlane32_swap (VALU) instead of ds_bpermute (an LDS round trip
// queued behind the fragment reads).  swap(x, x) leaves {lower-half values, upper-half values} in the two results for every lane.
__device__ __forceinline__ float half_pair_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_pair_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ============================================================================= DMA-staged MFMA attention (bf16, head width 64; round 3)
// attn_mfma_kernel above issues ~360 instructions per 64-key tile and wave for its 16 MFMAs: ~110 of them stage K / V (per-lane 64-bit
// addresses, bounds branches, register round trip, eight 2-byte transposing LDS writes per lane), 32 re-zero the score accumulators, and
// a tile costs two barriers.  The tiled GEMMs turned out to be bound by exactly this kind of issue overhead (tile 133); the same
// diet here:
//   * K and V tiles go global -> LDS by buffer_load ... lds through ONE descriptor: a lane-constant 32-bit offset per 8-row group
//     (circular window row, bounds -> an out-of-range offset, which lands zeros), no register staging, no ds_write at all;
//   * V stays ROW-MAJOR in LDS ([key][d], 16-byte chunks XOR-swizzled with bit 1 of the key so the four key rows of a transposing
//     read fall on different bank quarters) and the V^T fragments of O^T += V^T P^T come out of ds_read_b64_tr_b16: lane i of a
//     16-lane group supplies the 8 bytes V[key0 + i / 4][d0 + 4 (i % 4) ..] and receives V[key0 .. key0 + 3][d0 + i];
//   * K row-major with the GEMMs' swizzle (chunk ^ (row >> 1) & 7), fragments by ds_read_b128;
//   * the first S^T MFMA of a tile takes a zero accumulator operand instead of 32 v_mov;
//   * tile t + 1 is in flight while tile t is consumed: one raw s_barrier per tile behind a counted s_waitcnt.
// Same arithmetic as attn_mfma_kernel (S^T = K Q^T in the log2 domain, lane-local online softmax, P as the B operand straight from the
// S^T registers): bitwise the same output.
// Round 4 measured a VALU diet of this loop (q pre-multiplied by scale * log2 e, -m as the C operand of the first S^T MFMA, thresholded
// defer-max, v_max3 folds: 15 -> ~7 VALU instructions per MFMA): 105.1 vs 104.3 us on the spatial ds2 shape, 51.3 vs 48.7 and 49.7 vs
// 51.3 us on the two RS cross-attention shapes (profiles/r04_attn_fast_softmax_ab.txt) - nothing: the loop is bound by the dependent
// chain S^T MFMAs -> row max -> exp -> P V MFMAs of the three waves a SIMD holds, not by what it issues.  Not adopted (it re-rounds q).
template <int D>
__global__ __launch_bounds__(256, 2) void attn_dma_kernel(const AttnParams p) {
  static_assert(D == 64, "DMA-staged attention: head width 64 (one 128-byte LDS row per key)");
  constexpr int KST = D / 16, DT = D / 32, TILE_B = 64 * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                     // [2 stages][64 keys][128 B]
  char* sV = smem + 2 * TILE_B;        // [2 stages][64 keys][128 B]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  int qt, h, bg;
  attn_block_coords(qt, h, bg);
  const GroupInfo gi = group_info(p, bg);
  const int q0 = qt * 128;
  if (q0 >= gi.q_count) return;        // uniform per block
  const float sc = p.scale * 1.4426950408889634f;   // scores kept in log2 domain

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q = l31, half) holds d = 16 s + 8 half + [0, 8)
  const int qi = q0 + wave * 32 + l31;
  const bool qok = qi < gi.q_count;
  u32x4 qf[KST];
  {
    const char* qp = p.Q + ((gi.q_row0 + qi) * p.ldq + p.q_off + h * D) * 2;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qok) v = *(const u32x4*)(qp + (s * 16 + half * 8) * 2);
      qf[s] = v;
    }
  }
  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // ---- DMA: wave w stages row groups {w, w + 4} of K and of V; lane L of a group covers row 8 g + L / 8, physical chunk L % 8
  typedef __attribute__((address_space(3))) void* lptr_t;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.KV, 0, (int)((int64_t)p.nb * p.k_rows_per_batch * p.ldkv * 2), 0x00020000);
  const int lrow = lane >> 3, pc = lane & 7;
  int row_in_tile[2];
  uint32_t kcol[2], vcol[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (wave + 4 * i) + lrow;
    row_in_tile[i] = row;
    kcol[i] = (uint32_t)((p.k_off + h * D) * 2 + ((pc ^ ((row >> 1) & 7)) * 16));
    vcol[i] = (uint32_t)((p.v_off + h * D) * 2 + ((pc ^ (((row >> 1) & 1) << 2)) * 16));
  }
  const uint32_t ldb = (uint32_t)(p.ldkv * 2);
  auto issue = [&](int stage, int kt0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kk = kt0 + row_in_tile[i];
      int r = gi.k_start + kk;
      r = r >= gi.k_mod ? r - gi.k_mod : r;
      const uint32_t rowoff = (uint32_t)(gi.k_row0 + r) * ldb;
      const bool ok = kk < gi.k_count;
      const uint32_t ko = ok ? rowoff + kcol[i] : 0xfffffff0u, vo = ok ? rowoff + vcol[i] : 0xfffffff0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(sK + stage * TILE_B + (wave + 4 * i) * 1024), 16, ko, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(sV + stage * TILE_B + (wave + 4 * i) * 1024), 16, vo, 0, 0, 0);
    }
  };
  // fragment addresses.  K: row 32 kt + l31, logical chunk 2 st + half.  V^T by transposing reads: group (kt, st, u) of 4 keys
  // key0 = 32 kt + 16 st + 8 u + 4 half; this lane supplies row key0 + (lane & 15) / 4, columns dt * 32 + 16 ((lane >> 4) & 1) + 4 (lane & 3)
  const int kx = (l31 >> 1) & 7;
  const char* kbase = sK + l31 * 128;
  const int vrow0 = 4 * half + ((lane & 15) >> 2);                   // + 32 kt + 16 st + 8 u  (bit 1 of the row = bit 1 of vrow0's low part)
  const int vcolb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;   // byte column inside a 64-byte d tile: chunk = vcolb / 16 + 4 dt
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((address_space(3))) s16x4* lp4;

  const int ntiles = (gi.k_count + 63) >> 6;
  issue(0, 0);
  // S^T = K Q^T of the tile in `stage`: two 32-key sub-tiles, the first k-step on a zero accumulator
  auto compute_s = [&](int stage, f32x16 (&s)[2]) {
    const char* kb = kbase + stage * TILE_B;
    u32x4 kf[KST][2];
#pragma unroll
    for (int st = 0; st < KST; ++st)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) kf[st][kt] = *(const u32x4*)(kb + kt * 32 * 128 + (((2 * st + half) ^ kx) * 16));
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[0][kt]), __builtin_bit_cast(bf16x8, qf[0]), z, 0, 0, 0);
    }
#pragma unroll
    for (int st = 1; st < KST; ++st)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[st][kt]), __builtin_bit_cast(bf16x8, qf[st]), s[kt], 0, 0, 0);
  };
  // V^T fragments of the tile in `stage` (transposing reads)
  auto read_v = [&](int stage, bf16x8 (&vf)[2][2][DT]) {
    const char* vb = sV + stage * TILE_B;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          s16x4 lo, hi;
          {
            const int row = 32 * kt + 16 * st + vrow0;
            const int ch = ((vcolb >> 4) + 4 * dt) ^ (((row >> 1) & 1) << 2);
            lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(vb + row * 128 + ch * 16 + (vcolb & 15)));
          }
          {
            const int row = 32 * kt + 16 * st + 8 + vrow0;
            const int ch = ((vcolb >> 4) + 4 * dt) ^ (((row >> 1) & 1) << 2);
            hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(vb + row * 128 + ch * 16 + (vcolb & 15)));
          }
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          vf[kt][st][dt] = __builtin_bit_cast(bf16x8, both);
        }
  };
  // online softmax of tile t (scores in s, lane-local row; partner lane ^ 32 holds the other 32 keys), then O^T += V^T P^T
  auto softmax_pv = [&](int t, f32x16 (&s)[2], bf16x8 (&vf)[2][2][DT], auto ragged) {
    if constexpr (decltype(ragged)::value) {
      const int kbase2 = t * 64 + 4 * half;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = kbase2 + 32 * kt + (r & 3) + 8 * (r >> 2);
          s[kt][r] = kk < gi.k_count ? s[kt][r] : -3e38f;
        }
    }
    float ps = 0.f;
    {
      float mx = -3e38f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
      mx = half_pair_max(mx);       // (v_permlane32_swap: a ds_bpermute here drains lgkmcnt - the V fragment reads in flight - first)
      const float m_new = fmaxf(m_run, mx * sc);        // sc > 0: max commutes with the scale
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], sc, -m_new));
          s[kt][r] = e;
          ps += e;
        }
      ps = half_pair_sum(ps);
      l_run = l_run * alpha + ps;
      if (__any(m_new != m_run)) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
      m_run = m_new;
    }
    // P fragments straight from the S^T registers (k-slot j <-> reg 8*st + j)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)s[kt][8 * st + j];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt][st][dt], pf, o[dt], 0, 0, 0);
      }
  };
  const int nfull = gi.k_count >> 6;
  auto tile_body = [&](int t, auto ragged) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this tile's DMA (the only one in flight) has landed
    __builtin_amdgcn_s_barrier();                                    // ... for every wave; everyone is past the other stage's reads
    asm volatile("" ::: "memory");
    f32x16 s[2];
    compute_s(t & 1, s);
    bf16x8 vf[2][2][DT];
    read_v(t & 1, vf);                                               // requested under the softmax
    // the next tile's DMA goes out BEHIND the transposing reads: hipcc cannot tell that ds_read_b64_tr_b16 does not touch the stage an
    // outstanding LDS-DMA writes and put s_waitcnt vmcnt(0) in front of the first one - the wave sat out the whole L2 round trip of
    // the tile it had just requested (round 3 ISA).  In flight during this tile's softmax and P V, and every other wave's tile.
    if (t + 1 < ntiles) issue((t + 1) & 1, (t + 1) * 64);
    softmax_pv(t, s, vf, ragged);
  };
  for (int t = 0; t < nfull; ++t) tile_body(t, std::false_type{});
  if (nfull < ntiles) tile_body(nfull, std::true_type{});
  // ---- normalise, transpose through LDS (stage 0 of K / V: free after the last tile) and store whole 128-byte head rows
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    constexpr int SO = D * 2 + 16;           // padded row of the transposed tile
    const float inv = 1.f / l_run;
    if (p.lse2 && half == 0 && qok) p.lse2[(gi.q_row0 + qi) * p.heads + h] = m_run + __builtin_amdgcn_logf(l_run);
    char* so = smem + (wave * 32) * SO;      // this wave's 32 rows; written and read by this wave only (4 x 4.6 KB < 32 KB)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = dt * 32 + 8 * q4 + 4 * half;
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (__bf16)(o[dt][4 * q4 + e] * inv);
        *(bf16x4*)(so + l31 * SO + d * 2) = w;
      }
    const int tq = q0 + wave * 32;
#pragma unroll
    for (int ps2 = 0; ps2 < 32 * (D / 8) / 64; ++ps2) {
      const int idx = ps2 * 64 + lane;
      const int row = idx / (D / 8), v = idx % (D / 8);
      if (tq + row < gi.q_count) {
        const u32x4 x = *(const u32x4*)(so + row * SO + v * 16);
        *(u32x4*)(p.O + ((gi.q_row0 + tq + row) * p.ldo + h * D + v * 8) * 2) = x;
      }
    }
  }
}

// ============================================================================= hand-placed pipelined attention (bf16, head width 64; round 6)
// attn_dma_kernel above leaves the ORDER of a tile's ~300 instructions to hipcc, which emits them as three blocks: 8 S^T MFMAs, ~190 VALU
// instructions of softmax, 8 P V MFMAs (ISA of round 5).  A wave issues in order: during an MFMA block it stalls 32 cycles per MFMA with
// nothing else to issue, during the VALU block the matrix pipe has nothing from this wave - SQ counters: MFMA pipe busy 32.6 %, and five
// compiler-scheduled restructurings (rounds 3 - 4) measured nothing.  Here the order is fixed by hand:
//   * software pipeline over key tiles: iteration t runs the softmax of tile t (VALU) and, BETWEEN its instructions, the S^T MFMAs of tile
//     t + 1 and the P V MFMAs of tile t - one MFMA per ~10 VALU slots, so each MFMA's 32-cycle shadow is filled by the wave's own VALU
//     work instead of a stall;
//   * the whole iteration is ONE asm statement with literal registers (tools/gen_attn_pipe.py writes it: a list scheduler over the
//     iteration's dependency graph under the machine's hazard rules, counted `s_waitcnt lgkmcnt(N)` computed from the order of the LDS
//     reads, a linear-scan allocation of the temporaries); the compiler only sees the interface registers;
//   * the O rescale is unconditional (alpha == 1.0 exactly where the running max did not move): no branch in the stream;
//   * key tiles by descriptor DMA as above, but K runs TWO tiles ahead and V one (S^T of tile t + 1 needs K(t + 1) while P V still reads
//     V(t)); a full tile with no wrap of the circular window inside takes a lane-constant offset + a scalar tile offset (no per-tile
//     address VALU at all), the ragged / wrapping tile the per-lane form;
//   * waves whose 32 queries lie beyond the group (the 16-query tail block of the 400-query audio groups) only stage and synchronise.
// Same arithmetic, same order of every sum as attn_mfma_kernel: bitwise the same output (tests/test_round6_gpu.py).
#include "mmd_attn_pipe_body.inc"
// One pipelined iteration = one asm statement (generated, literal registers): softmax of the tile whose scores are in `sa` and its P V
// into o, S^T of the next tile into `sb`.  VAR = A: sa = v[0:31], sb = v[32:63], K stage 1, V stage 0;  B: the two sets and stages swapped.
#define ATTN_PIPE_ITER(VAR)                                                                                                            \
  asm volatile(ATTN_PIPE_ASM_##VAR                                                                                                     \
               : "+{v[0:15]}"(s0[0]), "+{v[16:31]}"(s0[1]), "+{v[32:47]}"(s1[0]), "+{v[48:63]}"(s1[1]), "+{v[64:79]}"(o[0]),           \
                 "+{v[80:95]}"(o[1]), "+{v119}"(m_run), "+{v120}"(l_run)                                                               \
               : "{v[96:99]}"(qf[0]), "{v[100:103]}"(qf[1]), "{v[104:107]}"(qf[2]), "{v[108:111]}"(qf[3]), "{v112}"(ka[0]),            \
                 "{v113}"(ka[1]), "{v114}"(ka[2]), "{v115}"(ka[3]), "{v116}"(va[0]), "{v117}"(va[1]), "{v118}"(sc)                     \
               : ATTN_PIPE_CLOBBERS, "memory")

template <int D>
__global__ __launch_bounds__(256, 2) void attn_pipe_kernel(const AttnParams p) {
  static_assert(D == 64, "pipelined attention: head width 64 (one 128-byte LDS row per key)");
  constexpr int KST = D / 16, DT = D / 32, TILE_B = 64 * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                     // [2 stages][64 keys][128 B]
  char* sV = smem + 2 * TILE_B;        // [2 stages][64 keys][128 B]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  int qt, h, bg;
  attn_block_coords(qt, h, bg);
  const GroupInfo gi = group_info(p, bg);
  const int q0 = qt * 128;
  if (q0 >= gi.q_count) return;        // uniform per block
  const float sc = p.scale * 1.4426950408889634f;   // scores kept in log2 domain
  const bool active = q0 + wave * 32 < gi.q_count;  // (wave-uniform) a wave without queries stages its share of K / V and keeps the barriers

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q = l31, half) holds d = 16 s + 8 half + [0, 8)
  const int qi = q0 + wave * 32 + l31;
  const bool qok = qi < gi.q_count;
  u32x4 qf[KST];
  {
    const char* qp = p.Q + ((gi.q_row0 + qi) * p.ldq + p.q_off + h * D) * 2;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qok) v = *(const u32x4*)(qp + (s * 16 + half * 8) * 2);
      qf[s] = v;
    }
  }
  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  // ---- DMA: wave w stages row groups {w, w + 4} of K and of V; lane L of a group covers row 8 g + L / 8, physical chunk L % 8
  typedef __attribute__((address_space(3))) void* lptr_t;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.KV, 0, (int)((int64_t)p.nb * p.k_rows_per_batch * p.ldkv * 2), 0x00020000);
  const int lrow = lane >> 3, pc = lane & 7;
  const uint32_t ldb = (uint32_t)(p.ldkv * 2);
  const uint32_t kcol_s = (uint32_t)((p.k_off + h * D) * 2), vcol_s = (uint32_t)((p.v_off + h * D) * 2);   // (scalar) column of this head
  int row_in_tile[2];
  uint32_t kswz[2], vswz[2], kfast[2], vfast[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (wave + 4 * i) + lrow;
    row_in_tile[i] = row;
    kswz[i] = (uint32_t)((pc ^ ((row >> 1) & 7)) * 16);
    vswz[i] = (uint32_t)((pc ^ (((row >> 1) & 1) << 2)) * 16);
    kfast[i] = (uint32_t)row * ldb + kswz[i];
    vfast[i] = (uint32_t)row * ldb + vswz[i];
  }
  // tile T of the window into `stage`: K rows (which == 0) or V rows (which == 1)
  auto issue = [&](int which, int stage, int T) {
    const int kt0 = T * 64;
    int r0 = gi.k_start + kt0;
    r0 = r0 >= gi.k_mod ? r0 - gi.k_mod : r0;
    char* dst = (which ? sV : sK) + stage * TILE_B;
    if (kt0 + 64 <= gi.k_count && r0 + 64 <= gi.k_mod) {      // (uniform) all 64 rows inside the window, no wrap inside the tile
      const uint32_t so = (uint32_t)(gi.k_row0 + r0) * ldb + (which ? vcol_s : kcol_s);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + (wave + 4 * i) * 1024), 16, which ? vfast[i] : kfast[i], so, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int kk = kt0 + row_in_tile[i];
        int r = gi.k_start + kk;
        r = r >= gi.k_mod ? r - gi.k_mod : r;
        const uint32_t off = (uint32_t)(gi.k_row0 + r) * ldb + (which ? vcol_s + vswz[i] : kcol_s + kswz[i]);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + (wave + 4 * i) * 1024), 16, kk < gi.k_count ? off : 0xfffffff0u, 0, 0, 0);
      }
    }
  };
  // fragment addresses (LDS byte addresses for the asm reads).  K: row 32 kt + l31, logical chunk 2 st + half.  V^T by transposing reads:
  // group (kt, st, u) of 4 keys, key0 = 32 kt + 16 st + 8 u + 4 half; this lane supplies row key0 + (lane & 15) / 4, columns
  // dt * 32 + 16 ((lane >> 4) & 1) + 4 (lane & 3); the chunk swizzle of a V row depends on bit 1 of the row = bit 1 of vrow0
  const int kx = (l31 >> 1) & 7;
  const char* kbase = sK + l31 * 128;
  const int vrow0 = 4 * half + ((lane & 15) >> 2);
  const int vcolb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  uint32_t ka[KST], va[DT];
#pragma unroll
  for (int st = 0; st < KST; ++st) ka[st] = lds0 + (uint32_t)(l31 * 128 + (((2 * st + half) ^ kx) * 16));
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
    va[dt] = lds0 + (uint32_t)(vrow0 * 128 + ((((vcolb >> 4) + 4 * dt) ^ ((vrow0 & 2) << 1)) * 16) + (vcolb & 15));
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((address_space(3))) s16x4* lp4;

  const int ntiles = (gi.k_count + 63) >> 6;
  issue(0, 0, 0);
  issue(1, 0, 0);
  if (ntiles > 1) issue(0, 1, 1);
  // ---- prologue: S^T of tile 0 (compiler-scheduled, as in attn_dma_kernel)
  f32x16 s0[2], s1[2];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (active) {
    u32x4 kf[KST][2];
#pragma unroll
    for (int st = 0; st < KST; ++st)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) kf[st][kt] = *(const u32x4*)(kbase + kt * 32 * 128 + (((2 * st + half) ^ kx) * 16));
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      s0[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[0][kt]), __builtin_bit_cast(bf16x8, qf[0]), z, 0, 0, 0);
    }
#pragma unroll
    for (int st = 1; st < KST; ++st)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        s0[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[st][kt]), __builtin_bit_cast(bf16x8, qf[st]), s0[kt], 0, 0, 0);
    // the loop's first statements read these accumulators from asm, where hipcc does not pad the MFMA -> VALU distance
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(s0[0]), "+v"(s0[1]));
  }
  // ---- pipelined iterations t = 0 .. ntiles - 2 (unrolled by two: the stage offsets are literals)
  auto top = [&](int t) {
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");    // K(t + 1), V(t) landed for everyone; everyone is past K(t), V(t - 1)
    if (t + 2 < ntiles) issue(0, t & 1, t + 2);
    if (t + 1 < ntiles) issue(1, (t + 1) & 1, t + 1);
    asm volatile("" ::: "memory");
  };
  int t = 0;
  for (; t + 2 < ntiles; t += 2) {
    top(t);
    if (active) ATTN_PIPE_ITER(A);
    top(t + 1);
    if (active) ATTN_PIPE_ITER(B);
  }
  if (t + 1 < ntiles) {
    top(t);
    if (active) {
      ATTN_PIPE_ITER(A);
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(s1[0]), "+v"(s1[1]));
      s0[0] = s1[0];
      s0[1] = s1[1];
    }
    ++t;
  }
  // ---- drain: softmax and P V of the last tile (t == ntiles - 1; the only one that can be ragged), compiler-scheduled
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (active) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(o[0]), "+v"(o[1]), "+v"(s0[0]), "+v"(s0[1]));
    const char* vb = sV + (t & 1) * TILE_B;
    bf16x8 vf[2][2][DT];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          s16x4 lo, hi;
          {
            const int row = 32 * kt + 16 * st + vrow0;
            const int ch = ((vcolb >> 4) + 4 * dt) ^ (((row >> 1) & 1) << 2);
            lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(vb + row * 128 + ch * 16 + (vcolb & 15)));
          }
          {
            const int row = 32 * kt + 16 * st + 8 + vrow0;
            const int ch = ((vcolb >> 4) + 4 * dt) ^ (((row >> 1) & 1) << 2);
            hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp4)(vb + row * 128 + ch * 16 + (vcolb & 15)));
          }
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          vf[kt][st][dt] = __builtin_bit_cast(bf16x8, both);
        }
    if (gi.k_count & 63) {                                           // (uniform) ragged last tile: keys beyond the window never win
      const int kbase2 = t * 64 + 4 * half;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = kbase2 + 32 * kt + (r & 3) + 8 * (r >> 2);
          s0[kt][r] = kk < gi.k_count ? s0[kt][r] : -3e38f;
        }
    }
    float ps = 0.f;
    float mx = -3e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s0[kt][r]);
    mx = half_pair_max(mx);
    const float m_new = fmaxf(m_run, mx * sc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[kt][r], sc, -m_new));
        s0[kt][r] = e;
        ps += e;
      }
    ps = half_pair_sum(ps);
    l_run = l_run * alpha + ps;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    m_run = m_new;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)s0[kt][8 * st + j];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[kt][st][dt], pf, o[dt], 0, 0, 0);
      }
  }
  // ---- normalise, transpose through LDS (stage 0 of K / V: free after the last tile) and store whole 128-byte head rows
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (active) {
    constexpr int SO = D * 2 + 16;           // padded row of the transposed tile
    const float inv = 1.f / l_run;
    if (p.lse2 && half == 0 && qok) p.lse2[(gi.q_row0 + qi) * p.heads + h] = m_run + __builtin_amdgcn_logf(l_run);
    char* so = smem + (wave * 32) * SO;      // this wave's 32 rows; written and read by this wave only (4 x 4.6 KB < 32 KB)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = dt * 32 + 8 * q4 + 4 * half;
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (__bf16)(o[dt][4 * q4 + e] * inv);
        *(bf16x4*)(so + l31 * SO + d * 2) = w;
      }
    const int tq = q0 + wave * 32;
#pragma unroll
    for (int ps2 = 0; ps2 < 32 * (D / 8) / 64; ++ps2) {
      const int idx = ps2 * 64 + lane;
      const int row = idx / (D / 8), v = idx % (D / 8);
      if (tq + row < gi.q_count) {
        const u32x4 x = *(const u32x4*)(so + row * SO + v * 16);
        *(u32x4*)(p.O + ((gi.q_row0 + tq + row) * p.ldo + h * D + v * 8) * 2) = x;
      }
    }
  }
}

// Round 4 also built a 64-queries-per-wave form of the kernel above (8 waves, two 32-query sub-tiles per wave sharing every K / V fragment
// read; key tiles streamed, or - video <- audio - the whole key window resident in LDS with no barrier in the loop), bitwise equal and
// tested, and measured it: 109.0 vs 108.9 us (spatial ds2), 50.7 vs 51.2 us (v <- a ds2, resident), 48.9 vs 50.8 us (a <- v ds2), 1.5 -
// 2.7x slower at ds4 (profiles/r04_attn_wide_kernel_bench.txt; the code is in the history at 22912f0).  Neither the per-tile barrier +
// DMA, nor the fragment traffic, nor the chains per wave bound this loop.  Not kept.

// ============================================================================= staged-window MFMA attention (bf16)
// For the long windows of the ds-2 level (spatial self-attention 1024 x 1024, RS cross-attention 1024 x 400 / 400 x 1024: ~80 % of
// the attention FLOPs of a step).  attn_mfma_kernel above re-stages K/V for every 128 queries, pays two barriers per 64 keys with
// one tile in flight, and issues every fragment read right in front of the MFMA that consumes it (the LDS latency of all 16 reads
// of a sub-tile is exposed): ~15-20 % of the MFMA peak.  Here:
//   * a block is 8 waves x one 32-query tile = 256 queries; keys go through LDS in stages of ATS_KEYS = 256 keys (K row-major with
//     padded rows, V transposed and key-permuted so a P.V fragment is ONE ds_read_b128); the NEXT stage's rows are fetched into
//     registers while this stage is consumed (issue early / write late): two barriers per 256 keys instead of two per 64;
//   * inside a stage a wave walks four 64-key sub-tiles with no barrier, in three clean phases per sub-tile: 8 back-to-back
//     S^T = K Q^T MFMAs on fragments that were read a phase earlier; the online softmax (pure VALU) under which the V fragments of
//     this sub-tile and the K fragments of the next one are fetched; 8 back-to-back O^T += V^T P^T MFMAs.  With two waves per SIMD
//     one wave's MFMA phase runs under the other's softmax phase.
//   * the running (m, l, O^T) stay in registers across stages; lane owns a query, P never leaves the registers.
#define ATS_KEYS 256
#define ATS_VT_STRIDE (ATS_KEYS * 2 + 16)   // bytes per V^T row ((stride/16) odd -> conflict-free ds_read_b128)

// position of key j inside its 16-key group of a V^T row: the 4-key quads are stored in the order 0, 2, 1, 3, so the eight keys
// {0-3, 8-11} + 4 half that a lane feeds to one MFMA (k-slot e <-> S^T register 8 st + e) are 16 contiguous bytes
__device__ __forceinline__ int ats_vt_col(int j) {
  const int q = (j >> 2) & 3;
  return (j & ~12) | ((((q & 1) << 1) | (q >> 1)) << 2);
}

template <int D>
__global__ __launch_bounds__(512, 1) void attn_stage_kernel(const AttnParams p) {
  constexpr int DV = D / 8;
  constexpr int SK = D * 2 + 16;
  constexpr int KST = D / 16;
  constexpr int DT = D / 32;
  constexpr int NK = (ATS_KEYS * DV + 511) / 512;
  static_assert(D % 32 == 0, "staged attention: head width must be a multiple of 32");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                         // [ATS_KEYS][SK]
  char* sVt = smem + ATS_KEYS * SK;        // [D][ATS_VT_STRIDE]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  int qc, h, bg;
  attn_block_coords(qc, h, bg);
  const GroupInfo gi = group_info(p, bg);
  const int q0 = qc * 256;
  if (q0 >= gi.q_count) return;            // uniform per block

  // ---- Q fragments (B operand of S^T = K Q^T): lane (q = l31, half) holds d = 16 s + 8 half + [0, 8)
  const int tq = q0 + wave * 32;
  const bool live = tq < gi.q_count;       // wave-uniform: a dead wave only helps staging
  const int qi = tq + l31;
  u32x4 qf[KST];
  {
    const char* qp = p.Q + ((gi.q_row0 + qi) * p.ldq + p.q_off + h * D) * 2;
#pragma unroll
    for (int s = 0; s < KST; ++s) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qi < gi.q_count) v = *(const u32x4*)(qp + (s * 16 + half * 8) * 2);
      qf[s] = v;
    }
  }
  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  const float sc = p.scale * 1.4426950408889634f;   // scores kept in log2 domain

  u32x4 rk[NK], rv[NK];
  auto load_stage = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int id = tid + 512 * i;
      {   // K: row-major, DV lanes per key row (coalesced)
        const int j = id / DV, v = id % DV;
        u32x4 x = {0u, 0u, 0u, 0u};
        if (id < ATS_KEYS * DV && k0 + j < gi.k_count)
          x = *(const u32x4*)(p.KV + (key_row(gi, k0 + j) * p.ldkv + p.k_off + h * D + v * 8) * 2);
        rk[i] = x;
      }
      {   // V: per 64-key block, 32 keys x 2 d-vecs per wave instruction (transposed 2-byte writes)
        const int kb = id / (64 * DV), idl = id % (64 * DV);
        const int j = kb * 64 + (idl & 31) + 32 * ((idl >> 6) & 1);
        const int v = 2 * (idl >> 7) + ((idl >> 5) & 1);
        u32x4 x = {0u, 0u, 0u, 0u};
        if (id < ATS_KEYS * DV && k0 + j < gi.k_count)
          x = *(const u32x4*)(p.KV + (key_row(gi, k0 + j) * p.ldkv + p.v_off + h * D + v * 8) * 2);
        rv[i] = x;
      }
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < NK; ++i) {
      const int id = tid + 512 * i;
      if (id < ATS_KEYS * DV) {
        {
          const int j = id / DV, v = id % DV;
          *(u32x4*)(sK + j * SK + v * 16) = rk[i];
        }
        {
          const int kb = id / (64 * DV), idl = id % (64 * DV);
          const int j = kb * 64 + (idl & 31) + 32 * ((idl >> 6) & 1);
          const int v = 2 * (idl >> 7) + ((idl >> 5) & 1);
          uint16_t* dst = (uint16_t*)(sVt + (8 * v) * ATS_VT_STRIDE + 2 * ats_vt_col(j));
#pragma unroll
          for (int e = 0; e < 8; ++e)
            dst[e * (ATS_VT_STRIDE / 2)] = (uint16_t)((rv[i][e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        }
      }
    }
  };

  // fragment addresses: K rows (sub*64 + kt*32 + l31), 16-byte chunk (half + 2 st); V^T rows (dt*32 + l31), keys of k-slot group
  // (kt, st) of this half = 16 contiguous bytes at column (sub*64 + 32 kt + 16 st) + 8 half
  const char* kbase = sK + l31 * SK + half * 16;
  const char* vbase = sVt + l31 * ATS_VT_STRIDE + half * 16;
  u32x4 kf[2 * KST];
  auto read_k = [&](int sub) {
#pragma unroll
    for (int st = 0; st < KST; ++st)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) kf[st * 2 + kt] = *(const u32x4*)(kbase + (sub * 64 + kt * 32) * SK + st * 32);
  };

  auto sub_tile = [&](int k0, int sub, bool more, auto ragged) {
    // ---- phase 1: S^T = K Q^T, two independent accumulator chains, fragments already in registers
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
    for (int st = 0; st < KST; ++st)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[st * 2 + kt]), __builtin_bit_cast(bf16x8, qf[st]), s[kt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    // ---- fetch under the softmax: V fragments of this sub-tile, K fragments of the next one
    u32x4 vf[4][DT];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) vf[g][dt] = *(const u32x4*)(vbase + dt * 32 * ATS_VT_STRIDE + (sub * 64 + 16 * g) * 2);
    if (more) read_k(sub + 1);
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2: online softmax (lane-local row; partner lane^32 holds the other 32 keys)
    if constexpr (decltype(ragged)::value) {
      const int kb = k0 + sub * 64 + 4 * half;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = kb + 32 * kt + (r & 3) + 8 * (r >> 2);
          s[kt][r] = kk < gi.k_count ? s[kt][r] : -3e38f;
        }
    }
    float mxp[4] = {-3e38f, -3e38f, -3e38f, -3e38f};        // four independent chains (a single 32-deep chain is latency-bound)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mxp[r & 3] = fmaxf(mxp[r & 3], s[kt][r]);
    const float mx = half_pair_max(fmaxf(fmaxf(mxp[0], mxp[1]), fmaxf(mxp[2], mxp[3])));
    const float m_new = fmaxf(m_run, mx * sc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], sc, -m_new));
        s[kt][r] = e;
        psp[r & 3] += e;
      }
    const float ps = half_pair_sum((psp[0] + psp[1]) + (psp[2] + psp[3]));
    l_run = l_run * alpha + ps;
    if (__any(m_new != m_run)) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    m_run = m_new;
    bf16x8 pf[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[g][e] = (__bf16)s[g >> 1][8 * (g & 1) + e];
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 3: O^T += V^T P^T (k-slot group g = 2 kt + st)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[g][dt]), pf[g], o[dt], 0, 0, 0);
  };

  load_stage(0);
  for (int k0 = 0; k0 < gi.k_count; k0 += ATS_KEYS) {
    __syncthreads();                       // the previous stage is fully consumed
    store_stage();
    __syncthreads();
    if (k0 + ATS_KEYS < gi.k_count) load_stage(k0 + ATS_KEYS);    // in flight during this stage's MFMAs
    if (live) {
      const int kn = min(gi.k_count - k0, ATS_KEYS);
      const int nfull = kn >> 6, nsub = (kn + 63) >> 6;
      read_k(0);
      for (int sub = 0; sub < nfull; ++sub) sub_tile(k0, sub, sub + 1 < nsub, std::false_type{});
      if (nfull < nsub) sub_tile(k0, nfull, false, std::true_type{});
    }
  }

  // ---- normalise, transpose through LDS (the K region, free after the last stage) and store whole 128-byte head rows: a lane
  //      owns a query ROW, so direct stores would be 8-byte pieces at a row stride (64 lines per wave instruction - the store tail
  //      cost 15 % of the kernel); staged, 8 lanes cover one row and a wave instruction writes 8 complete rows
  __syncthreads();                         // every wave is past its last K / V^T read
  {
    const float inv = 1.f / l_run;
    if (p.lse2 && half == 0 && qi < gi.q_count) p.lse2[(gi.q_row0 + qi) * p.heads + h] = m_run + __builtin_amdgcn_logf(l_run);
    char* so = smem + (wave * 32) * SK;    // this wave's 32 rows x (D*2 + 16) bytes; written and read by this wave only
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = dt * 32 + 8 * q4 + 4 * half;
        bf16x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (__bf16)(o[dt][4 * q4 + e] * inv);
        *(bf16x4*)(so + l31 * SK + d * 2) = w;
      }
#pragma unroll
    for (int ps = 0; ps < 32 * DV / 64; ++ps) {
      const int idx = ps * 64 + lane;
      const int row = idx / DV, v = idx % DV;
      if (tq + row < gi.q_count) {
        const u32x4 x = *(const u32x4*)(so + row * SK + v * 16);
        *(u32x4*)(p.O + ((gi.q_row0 + tq + row) * p.ldo + h * D + v * 8) * 2) = x;
      }
    }
  }
}

// ============================================================================= generic VALU flash attention
// KT keys per tile, OB output columns per lane (16 lanes across a row): <64, 8> up to head width 128; <32, 12> up to 192 (the SR
// U-Net's 768 channels / 4 heads in fp32 mode) - half the key tile so Q, K, V and S still fit the CU's LDS.
template <typename T, int KT, int OB>
__global__ __launch_bounds__(256) void attn_generic_kernel(const AttnParams p) {
  constexpr int KB = KT / 16, SS = KT + 1, KP = KT / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ch = p.ch;
  const int LQ = ch + 1;
  float* sQ = (float*)smem;            // [64][ch+1]  (pre-scaled)
  float* sK = sQ + 64 * LQ;            // [KT][ch+1]
  float* sV = sK + KT * LQ;            // [KT][ch]
  float* sS = sV + KT * ch;            // [64][KT+1]
  float* sM = sS + 64 * SS;            // [64] running max
  float* sL = sM + 64;                 // [64] running sum
  float* sAl = sL + 64;                // [64] rescale of this tile

  const int tid = threadIdx.x;
  int qt, h, bg;
  attn_block_coords(qt, h, bg);
  const GroupInfo gi = group_info(p, bg);
  const int q0 = qt * 64;
  if (q0 >= gi.q_count) return;
  const int ty = tid >> 4, tx = tid & 15;

  for (int i = tid; i < 64 * ch; i += 256) {
    const int r = i / ch, d = i % ch;
    float v = 0.f;
    if (q0 + r < gi.q_count) v = Elt<T>::ld(p.Q, (gi.q_row0 + q0 + r) * p.ldq + p.q_off + h * ch + d) * p.scale;
    sQ[r * LQ + d] = v;
  }
  if (tid < 64) { sM[tid] = -1e30f; sL[tid] = 0.f; }
  float oacc[4][OB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < OB; ++b) oacc[a][b] = 0.f;

  const int ntiles = (gi.k_count + KT - 1) / KT;
  for (int t = 0; t < ntiles; ++t) {
    __syncthreads();
    for (int i = tid; i < KT * ch; i += 256) {
      const int r = i / ch, d = i % ch;
      float kv = 0.f, vv = 0.f;
      if (t * KT + r < gi.k_count) {
        const int64_t row = key_row(gi, t * KT + r);
        kv = Elt<T>::ld(p.KV, row * p.ldkv + p.k_off + h * ch + d);
        vv = Elt<T>::ld(p.KV, row * p.ldkv + p.v_off + h * ch + d);
      }
      sK[r * LQ + d] = kv;
      sV[r * ch + d] = vv;
    }
    __syncthreads();
    // S block: rows ty*4.., keys tx*KB..
    float sacc[4][KB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < KB; ++b) sacc[a][b] = 0.f;
    for (int d = 0; d < ch; ++d) {
      float qa[4], kb[KB];
#pragma unroll
      for (int a = 0; a < 4; ++a) qa[a] = sQ[(ty * 4 + a) * LQ + d];
#pragma unroll
      for (int b = 0; b < KB; ++b) kb[b] = sK[(tx * KB + b) * LQ + d];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < KB; ++b) sacc[a][b] += qa[a] * kb[b];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < KB; ++b) {
        const int kk = t * KT + tx * KB + b;
        sS[(ty * 4 + a) * SS + tx * KB + b] = kk < gi.k_count ? sacc[a][b] : -1e30f;
      }
    __syncthreads();
    // row softmax update: 4 threads per row
    {
      const int r = tid >> 2, part = tid & 3;
      float mx = -1e30f;
      for (int k = part * KP; k < part * KP + KP; ++k) mx = fmaxf(mx, sS[r * SS + k]);
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      const float m_old = sM[r];
      const float m_new = fmaxf(m_old, mx);
      float sum = 0.f;
      for (int k = part * KP; k < part * KP + KP; ++k) {
        const float e = __expf(sS[r * SS + k] - m_new);
        sS[r * SS + k] = e;
        sum += e;
      }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      __syncthreads();   // everyone has read sM[r] before it is overwritten
      if (part == 0) {
        const float al = __expf(m_old - m_new);
        sAl[r] = al;
        sL[r] = sL[r] * al + sum;
        sM[r] = m_new;
      }
    }
    __syncthreads();
    // O update: rows ty*4.., columns tx + 16*b
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float al = sAl[ty * 4 + a];
#pragma unroll
      for (int b = 0; b < OB; ++b) oacc[a][b] *= al;
    }
    for (int k = 0; k < KT; ++k) {
      float pv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) pv[a] = sS[(ty * 4 + a) * SS + k];
#pragma unroll
      for (int b = 0; b < OB; ++b) {
        const int d = tx + 16 * b;
        if (d < ch) {
          const float vv = sV[k * ch + d];
#pragma unroll
          for (int a = 0; a < 4; ++a) oacc[a][b] += pv[a] * vv;
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = ty * 4 + a;
    if (q0 + r < gi.q_count) {
      const float inv = 1.f / sL[r];
#pragma unroll
      for (int b = 0; b < OB; ++b) {
        const int d = tx + 16 * b;
        if (d < ch) Elt<T>::st(p.O, (gi.q_row0 + q0 + r) * p.ldo + h * ch + d, oacc[a][b] * inv);
      }
    }
  }
}

// ============================================================================= short-sequence attention
struct SmallAttnParams {
  const char* QKV; int64_t ld;
  char* O; int64_t ldo;
  int C, heads, ch;
  int S, Tn, inner;
  int64_t outer_stride, inner_stride, tstride;
  float scale;
};

// One wave per (slice, head); lane = (query qi = lane/4 [+16], d-quarter dq = lane%4).
template <typename T, int CHQ>
__global__ __launch_bounds__(256) void attn_small_kernel(const SmallAttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ch = CHQ * 4;
  const int Tn = p.Tn;
  float* sK = (float*)smem + wave * 2 * 32 * ch;   // [Tn][ch]
  float* sV = sK + 32 * ch;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  const bool active = item < (int64_t)p.S * p.heads;
  const int s = active ? (int)(item / p.heads) : 0, h = active ? (int)(item % p.heads) : 0;
  const int64_t base = (int64_t)(s / p.inner) * p.outer_stride + (int64_t)(s % p.inner) * p.inner_stride;
  constexpr int EPV = Elt<T>::EPV;
  constexpr int ES = 16 / EPV;
  const bool vec_ok = (p.ld % EPV == 0) && (p.C % EPV == 0) && (((uintptr_t)p.QKV) % 16 == 0) && (ch % EPV == 0);
  if (active) {
    if (vec_ok) {
      const int cvn = ch / EPV;
      for (int i = lane; i < Tn * cvn; i += 64) {
        const int j = i / cvn, v = i % cvn;
        const int64_t row = base + (int64_t)j * p.tstride;
        float fk[EPV], fv[EPV];
        Elt<T>::unpack(*(const u32x4*)(p.QKV + (row * p.ld + p.C + h * ch + v * EPV) * ES), fk);
        Elt<T>::unpack(*(const u32x4*)(p.QKV + (row * p.ld + 2 * p.C + h * ch + v * EPV) * ES), fv);
#pragma unroll
        for (int e = 0; e < EPV; ++e) { sK[j * ch + v * EPV + e] = fk[e]; sV[j * ch + v * EPV + e] = fv[e]; }
      }
    } else {
      for (int i = lane; i < Tn * ch; i += 64) {
        const int j = i / ch, d = i % ch;
        const int64_t row = base + (int64_t)j * p.tstride;
        sK[j * ch + d] = Elt<T>::ld(p.QKV, row * p.ld + p.C + h * ch + d);
        sV[j * ch + d] = Elt<T>::ld(p.QKV, row * p.ld + 2 * p.C + h * ch + d);
      }
    }
  }
  __syncthreads();
  if (!active) return;
  const int dq = lane & 3;
  for (int qb = 0; qb < Tn; qb += 16) {
    const int qi = qb + (lane >> 2);
    const bool ok = qi < Tn;
    float q[CHQ];
    {
      const int64_t row = base + (int64_t)(ok ? qi : 0) * p.tstride;
      if (vec_ok && CHQ % EPV == 0) {
#pragma unroll
        for (int d = 0; d < CHQ; d += EPV) {
          float f[EPV];
          Elt<T>::unpack(*(const u32x4*)(p.QKV + (row * p.ld + h * ch + dq * CHQ + d) * ES), f);
#pragma unroll
          for (int e = 0; e < EPV; ++e) q[(d + e) < CHQ ? (d + e) : 0] = f[e] * p.scale;
        }
      } else {
#pragma unroll
        for (int d = 0; d < CHQ; ++d) q[d] = Elt<T>::ld(p.QKV, row * p.ld + h * ch + dq * CHQ + d) * p.scale;
      }
    }
    float sc[32];
    float mx = -1e30f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      float a = 0.f;
      if (j < Tn) {
#pragma unroll
        for (int d = 0; d < CHQ; ++d) a += q[d] * sK[j * ch + dq * CHQ + d];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        mx = fmaxf(mx, a);
      }
      sc[j] = a;
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const float e = j < Tn ? __expf(sc[j] - mx) : 0.f;
      sc[j] = e;
      sum += e;
    }
    float o[CHQ];
#pragma unroll
    for (int d = 0; d < CHQ; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if (j < Tn) {
#pragma unroll
        for (int d = 0; d < CHQ; ++d) o[d] += sc[j] * sV[j * ch + dq * CHQ + d];
      }
    }
    if (ok) {
      const float inv = 1.f / sum;
      const int64_t row = base + (int64_t)qi * p.tstride;
      if (CHQ % EPV == 0 && p.ldo % EPV == 0 && ((uintptr_t)p.O) % 16 == 0) {
#pragma unroll
        for (int d = 0; d < CHQ; d += EPV) {
          float f[EPV];
#pragma unroll
          for (int e = 0; e < EPV; ++e) f[e] = o[(d + e) < CHQ ? (d + e) : 0] * inv;
          *(u32x4*)(p.O + (row * p.ldo + h * ch + dq * CHQ + d) * ES) = Elt<T>::pack(f);
        }
      } else {
#pragma unroll
        for (int d = 0; d < CHQ; ++d) Elt<T>::st(p.O, row * p.ldo + h * ch + dq * CHQ + d, o[d] * inv);
      }
    }
  }
}

// attn_small_mfma_kernel (bf16, Tn <= 16, ch in {32,64,96,128}): the same (slice, head) item per wave, on
// v_mfma_f32_16x16x32_bf16 with NO LDS staging - the VALU kernel above re-reads its K / V quarter from LDS for every query
// (128 KB of LDS reads per 6 KB item: LDS-return bound at ~1.4 TB/s of HBM traffic).
//   S^T = K Q^T : A = K rows, B = Q rows, both loaded straight into fragment layout (lane&15 = token, lane>>4 = 8-channel
//                 k-slot group); D gives the lane of query t = lane&15 the scores of keys 4g .. 4g+3 (g = lane>>4).
//   softmax     : over the lane's 4 keys and the 4 lanes sharing lane&15 (xor 16, 32), fp32, exp2 with the scale folded in.
//   O^T = V^T P^T: k-slot (g, e<4) <-> key 4g+e (slots e >= 4 repeat the keys for the low half of P), so P^T is the lane's own
//                 4 probabilities; V^T rows
//                 are 2-byte gathers (4 keys of one channel).  Row r of tile tt carries channel (CH/4)*(r>>2) + 4*tt + (r&3),
//                 which leaves lane (t, g) with the CH/4 contiguous channels g*CH/4 .. of query t -> 16-byte stores.
template <int CH>
__global__ __launch_bounds__(256) void attn_small_mfma_kernel(const SmallAttnParams p) {
  constexpr int NT = CH / 16, KK = CH / 32, OWN = CH / 4;
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (int64_t)p.S * p.heads) return;
  const int s = (int)(item / p.heads), h = (int)(item % p.heads);
  const int64_t base = (int64_t)(s / p.inner) * p.outer_stride + (int64_t)(s % p.inner) * p.inner_stride;
  const int t = lane & 15, g = lane >> 4;
  const bool tok = t < p.Tn;
  const char* qrow = p.QKV + ((base + (int64_t)(tok ? t : 0) * p.tstride) * p.ld + h * CH) * 2;
  u32x4 qf[KK], kf[KK];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const u32x4 q = *(const u32x4*)(qrow + (kk * 32 + g * 8) * 2);
    const u32x4 k = *(const u32x4*)(qrow + (p.C + kk * 32 + g * 8) * 2);
    qf[kk] = tok ? q : zero4;
    kf[kk] = tok ? k : zero4;
  }
  // V^T fragments: channel of this lane's A row per tile, keys 4g .. 4g+3 (clamped to a valid row: their P is zero)
  uint32_t vlo[NT], vhi[NT];
  {
    const char* vcol = p.QKV + (2 * p.C + h * CH + OWN * (t >> 2) + (t & 3)) * 2;
    const char* vr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = 4 * g + e;
      vr[e] = vcol + (base + (int64_t)(key < p.Tn ? key : 0) * p.tstride) * p.ld * 2;
    }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
      const uint32_t v0 = *(const uint16_t*)(vr[0] + tt * 8), v1 = *(const uint16_t*)(vr[1] + tt * 8);
      const uint32_t v2 = *(const uint16_t*)(vr[2] + tt * 8), v3 = *(const uint16_t*)(vr[3] + tt * 8);
      vlo[tt] = v0 | (v1 << 16);
      vhi[tt] = v2 | (v3 << 16);
    }
  }
  f32x4 sT = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
    sT = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kf[kk]), __builtin_bit_cast(bf16x8, qf[kk]), sT, 0, 0, 0);
  const float sc = p.scale * 1.44269504088896f;
  float e4[4], mx = -3e38f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    e4[i] = 4 * g + i < p.Tn ? sT[i] * sc : -3e38f;
    mx = fmaxf(mx, e4[i]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    e4[i] = 4 * g + i < p.Tn ? __builtin_amdgcn_exp2f(e4[i] - mx) : 0.f;
    sum += e4[i];
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  // probabilities as a bf16 hi + lo pair in k-slots 0-3 / 4-7 (V duplicated likewise): the product keeps ~16 mantissa bits
  // of P, i.e. the fp32-softmax numerics of the VALU kernel, for the price of the otherwise empty half of the MFMA
  float lo4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) lo4[i] = e4[i] - bf16_bits_to_f32(f32_to_bf16_bits(e4[i]));
  const u32x4 pf = {pack_bf16x2(e4[0], e4[1]), pack_bf16x2(e4[2], e4[3]), pack_bf16x2(lo4[0], lo4[1]), pack_bf16x2(lo4[2], lo4[3])};
  float o[NT * 4];
#pragma unroll
  for (int tt = 0; tt < NT; ++tt) {
    const u32x4 vf = {vlo[tt], vhi[tt], vlo[tt], vhi[tt]};
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf), z, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[tt * 4 + i] = r[i];
  }
  if (tok) {
    const float inv = 1.f / sum;
    char* op = p.O + ((base + (int64_t)t * p.tstride) * p.ldo + h * CH + OWN * g) * 2;
#pragma unroll
    for (int c = 0; c < OWN; c += 8) {
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = o[c + j] * inv;
      *(u32x4*)(op + c * 2) = Elt<__bf16>::pack(f);
    }
  }
}

// ============================================================================= C-ABI
template <int D>
static int launch_mfma(const AttnParams& p, int qmax, hipStream_t st) {
  constexpr int SK = D * 2 + 16;
  constexpr int DT = (D + 31) / 32;
  const size_t lds = 64 * SK + DT * 32 * SVT_STRIDE;
  dim3 grid(cdiv(qmax, 128), p.heads, p.nb * p.G);
  hipLaunchKernelGGL(attn_mfma_kernel<D>, grid, dim3(256), lds, st, p);
  return mmd_check_launch("attn_mfma");
}

template <int D>
static int launch_dma(const AttnParams& p, int qmax, hipStream_t st) {
  const size_t lds = 4 * 64 * 128;
  dim3 grid(cdiv(qmax, 128), p.heads, p.nb * p.G);
  hipLaunchKernelGGL(attn_dma_kernel<D>, grid, dim3(256), lds, st, p);
  return mmd_check_launch("attn_dma");
}

template <int D>
static int launch_pipe(const AttnParams& p, int qmax, hipStream_t st) {
  // tuning switch (read once; tools only): MMD_ATTN_PIPE_LDSPAD = extra bytes of LDS per block, e.g. 65536 = one block per CU
  static const int pad = [] { const char* e = getenv("MMD_ATTN_PIPE_LDSPAD"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 120 * 1024 ? v : 0; }();
  const size_t lds = 4 * 64 * 128 + (size_t)pad;
  if (pad) {
    static bool attr_done[MMD_MAX_DEVICES] = {};
    bool& attr_set = attr_done[mmd_device_slot()];
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)attn_pipe_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return mmd_set_error(MMD_ERR_LAUNCH, "attn_pipe: set LDS attr");
      attr_set = true;
    }
  }
  dim3 grid(cdiv(qmax, 128), p.heads, p.nb * p.G);
  hipLaunchKernelGGL(attn_pipe_kernel<D>, grid, dim3(256), lds, st, p);
  return mmd_check_launch("attn_pipe");
}

template <int D>
static int launch_stage(const AttnParams& p, int qmax, hipStream_t st) {
  const size_t lds = (size_t)ATS_KEYS * (D * 2 + 16) + (size_t)D * ATS_VT_STRIDE;
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_stage_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "attn_stage: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  dim3 grid(cdiv(qmax, 256), p.heads, p.nb * p.G);
  hipLaunchKernelGGL((attn_stage_kernel<D>), grid, dim3(512), lds, st, p);
  return mmd_check_launch("attn_stage");
}

template <typename T, int KT, int OB>
static int launch_generic_as(const AttnParams& p, int qmax, hipStream_t st) {
  const size_t lds = (size_t)(64 * (p.ch + 1) + KT * (p.ch + 1) + KT * p.ch + 64 * (KT + 1) + 192) * sizeof(float);
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_generic_kernel<T, KT, OB>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "attn_generic: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  dim3 grid(cdiv(qmax, 64), p.heads, p.nb * p.G);
  hipLaunchKernelGGL((attn_generic_kernel<T, KT, OB>), grid, dim3(256), lds, st, p);
  return mmd_check_launch("attn_generic");
}

template <typename T>
static int launch_generic(const AttnParams& p, int qmax, hipStream_t st) {
  return p.ch <= 128 ? launch_generic_as<T, 64, 8>(p, qmax, st) : launch_generic_as<T, 32, 12>(p, qmax, st);
}

static int attn_fwd_impl(int dtype, const void* Q, int64_t ldq, int q_off, const void* KV, int64_t ldkv, int k_off,
                         int v_off, void* O, int64_t ldo, int heads, int ch, int nb, int G, int64_t q_rows_per_batch,
                         int q_per_group, int64_t k_rows_per_batch, int k_per_group, int win, const int* shift_dev,
                         int impl, float* lse2_out, void* stream) {
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "attn_fwd: bad dtype %d", dtype);
  MMD_REQUIRE(Q && KV && O, "attn_fwd: null pointer");
  MMD_REQUIRE(heads > 0 && ch > 0 && ch <= 192 && nb > 0 && G > 0, "attn_fwd: bad heads/ch/nb/G (%d,%d,%d,%d)", heads, ch, nb, G);
  MMD_REQUIRE(q_per_group > 0 && (int64_t)(G - 1) * q_per_group < q_rows_per_batch, "attn_fwd: bad query grouping");
  MMD_REQUIRE(k_per_group > 0 && win > 0 && (int64_t)win * k_per_group <= k_rows_per_batch, "attn_fwd: key window exceeds the key rows");
  AttnParams p;
  p.Q = (const char*)Q; p.ldq = ldq; p.KV = (const char*)KV; p.ldkv = ldkv; p.O = (char*)O; p.ldo = ldo;
  p.q_off = q_off; p.k_off = k_off; p.v_off = v_off; p.heads = heads; p.ch = ch; p.nb = nb; p.G = G;
  p.q_rows_per_batch = q_rows_per_batch; p.q_per_group = q_per_group; p.k_rows_per_batch = k_rows_per_batch;
  p.k_per_group = k_per_group; p.win = win; p.shift_ptr = shift_dev;
  p.scale = 1.0f / sqrtf((float)ch);
  p.lse2 = lse2_out;
  const int qmax = (int)(q_rows_per_batch - (int64_t)(G - 1) * q_per_group);   // last group is the largest
  hipStream_t st = (hipStream_t)stream;
  const bool aligned = ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 4 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 &&
                       ((uintptr_t)Q | (uintptr_t)KV) % 16 == 0 && (uintptr_t)O % 8 == 0;
  // long windows at head width 64 (the ds-2 level): staged-window kernel, K/V staged once per 256 / 512 queries
  const int k_count = win * k_per_group;
  // the staged kernel stores O as whole 16-byte vectors (the per-128-query kernel: 8-byte pieces): stricter output alignment
  const bool stage_ok = dtype == MMD_BF16 && aligned && ch == 64 && ldo % 8 == 0 && (uintptr_t)O % 16 == 0;
  // DMA-staged kernel (impl 4; the default at head width 64 unless MMD_ATTN_DMA=0): 32-bit byte offsets into the K/V rows
  static const bool dma_on = [] { const char* e = getenv("MMD_ATTN_DMA"); return !(e && e[0] == '0'); }();
  // Its descriptor covers the K/V rows of the batches of ONE launch: a batch range beyond 2 GB is cut into several launches of
  // the same kernel (the kernel family of a layer - and with it the last bits of its output - must not depend on the batch size)
  const int64_t kv_batch_bytes = k_rows_per_batch * ldkv * 2;
  const bool dma_ok = stage_ok && kv_batch_bytes < 0x7fffffffLL;
  // hand-placed pipelined kernel (impl 5; round 6): same staging, same arithmetic, bitwise the same output
  static const bool pipe_on = [] { const char* e = getenv("MMD_ATTN_PIPE"); return e && e[0] == '1'; }();
  const bool use_pipe = impl == 5 || (impl == 0 && pipe_on);
  if (impl == 5 && !dma_ok) return mmd_set_error(MMD_ERR_UNSUPPORTED, "attn_fwd impl 5 (pipelined): needs bf16, head width 64, aligned rows, one batch of K/V below 2 GB");
  if (impl == 4 && !dma_ok) return mmd_set_error(MMD_ERR_UNSUPPORTED, "attn_fwd impl 4 (DMA-staged): needs bf16, head width 64, aligned rows, one batch of K/V below 2 GB");
  if (dma_ok && (impl == 4 || impl == 5 || (impl == 0 && dma_on))) {
    const int per = (int)(0x7fffffffLL / kv_batch_bytes) < nb ? (int)(0x7fffffffLL / kv_batch_bytes) : nb;
    for (int n0 = 0; n0 < nb; n0 += per) {
      AttnParams c = p;
      c.nb = nb - n0 < per ? nb - n0 : per;
      c.Q = p.Q + (int64_t)n0 * q_rows_per_batch * ldq * 2;
      c.KV = p.KV + (int64_t)n0 * kv_batch_bytes;
      c.O = p.O + (int64_t)n0 * q_rows_per_batch * ldo * 2;
      if (p.lse2) c.lse2 = p.lse2 + (int64_t)n0 * q_rows_per_batch * heads;
      const int rc = use_pipe ? launch_pipe<64>(c, qmax, st) : launch_dma<64>(c, qmax, st);
      if (rc != MMD_OK) return rc;
    }
    return MMD_OK;
  }
  if (impl == 3 && !stage_ok) return mmd_set_error(MMD_ERR_UNSUPPORTED, "attn_fwd impl 3 (staged window): needs bf16, head width 64, aligned rows");
  // auto rule from tools/attn_bench.py on MI355X (batch 4): the staged kernel wins where a group has FEW queries per staged key
  // (audio <- video at ds2: 400 queries x 1024 keys, 63 us vs 74 us) and loses a few percent where the per-128-query kernel's three
  // co-resident blocks per CU hide its staging (spatial 1024 x 1024: 142 vs 129 us; video <- audio 1024 x 400: 69 vs 67 us)
  if (stage_ok && (impl == 3 || (impl == 0 && k_count >= 768 && qmax >= 128 && qmax <= 512)))
    return launch_stage<64>(p, qmax, st);
  if (dtype == MMD_BF16 && (impl == 0 || impl == 2 || impl == 3) && aligned) {
    switch (ch) {
      case 16: return launch_mfma<16>(p, qmax, st);
      case 32: return launch_mfma<32>(p, qmax, st);
      case 48: return launch_mfma<48>(p, qmax, st);
      case 64: return launch_mfma<64>(p, qmax, st);
      case 96: return launch_mfma<96>(p, qmax, st);
      case 128: return launch_mfma<128>(p, qmax, st);
      case 192: return launch_mfma<192>(p, qmax, st);      // SR U-Net: 768 channels / 4 heads
      default: break;
    }
  }
  if (lse2_out) return mmd_set_error(MMD_ERR_UNSUPPORTED, "attn_fwd_lse: needs the bf16 MFMA path (ch in {16,32,48,64,96,128}, aligned rows)");
  if (dtype == MMD_BF16) return launch_generic<__bf16>(p, qmax, st);
  return launch_generic<float>(p, qmax, st);
}

// impl: 0 = auto (MFMA when dtype is bf16 and ch is supported; staged-window kernel for long windows at head width 64),
// 1 = force generic VALU kernel, 2 = force the per-128-query MFMA kernel, 3 = force the staged-window kernel
extern "C" int mmd_attn_fwd(int dtype, const void* Q, int64_t ldq, int q_off, const void* KV, int64_t ldkv, int k_off,
                            int v_off, void* O, int64_t ldo, int heads, int ch, int nb, int G, int64_t q_rows_per_batch,
                            int q_per_group, int64_t k_rows_per_batch, int k_per_group, int win, const int* shift_dev,
                            int impl, void* stream) {
  return attn_fwd_impl(dtype, Q, ldq, q_off, KV, ldkv, k_off, v_off, O, ldo, heads, ch, nb, G, q_rows_per_batch, q_per_group,
                       k_rows_per_batch, k_per_group, win, shift_dev, impl, nullptr, stream);
}

// Training forward (bf16 MFMA path only): as mmd_attn_fwd, plus lse2_out [q rows, heads] = log2-domain log-sum-exp of
// (scale * log2 e * q.k) per query row, consumed by mmd_attn_bwd_mfma.
extern "C" int mmd_attn_fwd_lse(int dtype, const void* Q, int64_t ldq, int q_off, const void* KV, int64_t ldkv, int k_off,
                                int v_off, void* O, int64_t ldo, int heads, int ch, int nb, int G, int64_t q_rows_per_batch,
                                int q_per_group, int64_t k_rows_per_batch, int k_per_group, int win, const int* shift_dev,
                                float* lse2_out, void* stream) {
  MMD_REQUIRE(lse2_out, "attn_fwd_lse: null lse buffer");
  return attn_fwd_impl(dtype, Q, ldq, q_off, KV, ldkv, k_off, v_off, O, ldo, heads, ch, nb, G, q_rows_per_batch, q_per_group,
                       k_rows_per_batch, k_per_group, win, shift_dev, 0, lse2_out, stream);
}

template <typename T, int CHQ>
static int launch_small(const SmallAttnParams& p, hipStream_t st) {
  const size_t lds = (size_t)4 * 2 * 32 * (CHQ * 4) * sizeof(float);
  static bool attr_done[MMD_MAX_DEVICES] = {};
  bool& attr_set = attr_done[mmd_device_slot()];
  if (!attr_set && lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_small_kernel<T, CHQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return mmd_set_error(MMD_ERR_LAUNCH, "attn_small: set LDS attr: %s", hipGetErrorString(e));
    attr_set = true;
  }
  const int64_t items = (int64_t)p.S * p.heads;
  hipLaunchKernelGGL((attn_small_kernel<T, CHQ>), dim3((unsigned)((items + 3) / 4)), dim3(256), lds, st, p);
  return mmd_check_launch("attn_small");
}

template <typename T>
static int dispatch_small(const SmallAttnParams& p, hipStream_t st) {
  switch (p.ch) {
    case 16: return launch_small<T, 4>(p, st);
    case 32: return launch_small<T, 8>(p, st);
    case 48: return launch_small<T, 12>(p, st);
    case 64: return launch_small<T, 16>(p, st);
    case 96: return launch_small<T, 24>(p, st);
    case 128: return launch_small<T, 32>(p, st);
    default: return mmd_set_error(MMD_ERR_UNSUPPORTED, "attn_small: head width %d not in {16,32,48,64,96,128}", p.ch);
  }
}

// Self-attention over short slices (temporal attention: slices = pixels, Tn = frames <= 32).
// qkv rows hold [q(C) | k(C) | v(C)], head h = channels h*ch .. (h+1)*ch of each third.
extern "C" int mmd_attn_small_fwd(int dtype, const void* QKV, int64_t ld, void* O, int64_t ldo, int C, int heads, int S,
                                  int Tn, int inner, int64_t outer_stride, int64_t inner_stride, int64_t tstride,
                                  void* stream) {
  MMD_REQUIRE(dtype == MMD_BF16 || dtype == MMD_F32, "attn_small: bad dtype %d", dtype);
  MMD_REQUIRE(QKV && O && C > 0 && heads > 0 && C % heads == 0, "attn_small: bad argument");
  MMD_REQUIRE(Tn >= 1 && Tn <= 32, "attn_small: sequence length %d not in [1,32]", Tn);
  SmallAttnParams p;
  p.QKV = (const char*)QKV; p.ld = ld; p.O = (char*)O; p.ldo = ldo; p.C = C; p.heads = heads; p.ch = C / heads;
  p.S = S; p.Tn = Tn; p.inner = inner; p.outer_stride = outer_stride; p.inner_stride = inner_stride; p.tstride = tstride;
  p.scale = 1.0f / sqrtf((float)p.ch);
  hipStream_t st = (hipStream_t)stream;
  const bool vec = ld % 8 == 0 && ldo % 8 == 0 && C % 8 == 0 && ((uintptr_t)QKV) % 16 == 0 && ((uintptr_t)O) % 16 == 0;
  if (dtype == MMD_BF16 && Tn <= 16 && vec && (p.ch == 32 || p.ch == 64 || p.ch == 96 || p.ch == 128)) {
    const dim3 grid((unsigned)(((int64_t)S * heads + 3) / 4));
    switch (p.ch) {
      case 32: hipLaunchKernelGGL(attn_small_mfma_kernel<32>, grid, dim3(256), 0, st, p); break;
      case 64: hipLaunchKernelGGL(attn_small_mfma_kernel<64>, grid, dim3(256), 0, st, p); break;
      case 96: hipLaunchKernelGGL(attn_small_mfma_kernel<96>, grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL(attn_small_mfma_kernel<128>, grid, dim3(256), 0, st, p); break;
    }
    return mmd_check_launch("attn_small_mfma");
  }
  return dtype == MMD_BF16 ? dispatch_small<__bf16>(p, st) : dispatch_small<float>(p, st);
}
