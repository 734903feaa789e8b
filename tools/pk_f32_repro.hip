// Standalone reproducer for the round-3 packed-fp32 miscompute (no Python, no engine, no torch): gn_small_kernel - the one-launch
// GroupNorm of short slices - came out ~1e-2 wrong in lanes 48-63 (the HIGH register of its v_pk_add_f32 / v_pk_fma_f32 accumulator
// pairs) in 1 of ~130 graph replays whenever its waves shared a SIMD with the other stream's 128-row GEMM whose loader applies
// GroupNorm + SiLU (v_exp / v_rcp heavy).  This file compiles the SAME kernels (it includes the library's sources) into a plain
// executable, twice:
//     tools/pk_f32_repro.sh   ->  pk_repro_packed (default code generation: packed fp32 allowed)
//                                  pk_repro_nopk   (-Xclang -target-feature -Xclang -packed-fp32-ops: the product's build)
// Each run captures ONE hipGraph with two branches - REPS launches of the victim (mmd_gn_small) on the origin stream, REPS launches of
// the aggressor (mmd_gn_conv1x1, tile 128) on a forked stream - replays it, and compares every victim output with the output of
// the victim run alone.  It prints the number of replays (and launches) that differ.  Exit code 0 = no difference.
#include "../mm-diffusion_amd/csrc/mmd_core.hip"
#include "../mm-diffusion_amd/csrc/mmd_norm.hip"
#include "../mm-diffusion_amd/csrc/mmd_gemm.hip"
#include <vector>
#include <stdlib.h>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)
#define MM(x)                                                                        \
  do {                                                                               \
    if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, mmd_last_error()); exit(2); }    \
  } while (0)

static uint16_t bf16_of(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float frand(uint32_t& s) {          // uniform (-1, 1), LCG
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) * (1.0f / 8388608.0f)) - 1.0f;
}

int main(int argc, char** argv) {
  const int replays = argc > 1 ? atoi(argv[1]) : 300, REPS = argc > 2 ? atoi(argv[2]) : 6;
  // victim: the temporal-attention norm of the mid-size model - slices = the 16 frames of a pixel (row stride HW), C channels
  const int N = 2, F = 16, HW = 256, C = argc > 3 ? atoi(argv[3]) : 128;
  const int rows = N * F * HW, S = N * HW;
  // aggressor: GroupNorm + SiLU + 1x1 conv on the 128-row register-staged tile (audio ResBlock out conv of that model)
  const int M = 12800, Ca = 128, Sa = 2;
  uint32_t seed = 12345u;
  std::vector<uint16_t> hx((size_t)rows * C), ha((size_t)M * Ca), hw((size_t)Ca * Ca);
  for (auto& v : hx) v = bf16_of(frand(seed) * 2.f);
  for (auto& v : ha) v = bf16_of(frand(seed) * 2.f);
  for (auto& v : hw) v = bf16_of(frand(seed) * 0.1f);
  std::vector<float> hg(C), hb(C), haa((size_t)Sa * Ca), hab((size_t)Sa * Ca), hbias(Ca);
  for (auto& v : hg) v = 1.f + 0.3f * frand(seed);
  for (auto& v : hb) v = 0.3f * frand(seed);
  for (auto& v : haa) v = 1.f + 0.3f * frand(seed);
  for (auto& v : hab) v = 0.3f * frand(seed);
  for (auto& v : hbias) v = 0.1f * frand(seed);
  uint16_t *dx, *dy, *dref, *da, *dw, *dya;
  float *dg, *db, *daa, *dab, *dbias;
  CK(hipMalloc(&dx, hx.size() * 2));
  CK(hipMalloc(&dy, hx.size() * 2 * REPS));
  CK(hipMalloc(&dref, hx.size() * 2));
  CK(hipMalloc(&da, ha.size() * 2));
  CK(hipMalloc(&dya, ha.size() * 2));
  CK(hipMalloc(&dw, hw.size() * 2));
  CK(hipMalloc(&dg, C * 4)); CK(hipMalloc(&db, C * 4));
  CK(hipMalloc(&daa, haa.size() * 4)); CK(hipMalloc(&dab, hab.size() * 4)); CK(hipMalloc(&dbias, Ca * 4));
  CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dg, hg.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(daa, haa.data(), haa.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dab, hab.data(), hab.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, hbias.data(), Ca * 4, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  auto victim = [&](uint16_t* out, hipStream_t st) {
    // temporal geometry: slice (n, pixel) = rows n * F * HW + pixel + j * HW, j < F
    MM(mmd_gn_small(MMD_BF16, dx, C, out, C, C, S, F, HW, (int64_t)F * HW, 1, HW, dg, db, 1e-5f, 0, st));
  };
  auto aggressor = [&](hipStream_t st) {
    MM(mmd_gn_conv1x1(MMD_BF16, da, Ca, daa, dab, 1, Sa, M / Sa, dw, dbias, nullptr, 0, dya, Ca, M, Ca, Ca, 128, st));
  };
  // reference: the victim alone (twice: it must at least agree with itself)
  victim(dref, sa);
  victim(dy, sa);
  aggressor(sb);                              // (warm: function attributes are set outside the capture)
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> ref(hx.size()), got(hx.size());
  CK(hipMemcpy(ref.data(), dref, ref.size() * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(got.data(), dy, got.size() * 2, hipMemcpyDeviceToHost));
  if (memcmp(ref.data(), got.data(), ref.size() * 2) != 0) { printf("victim alone is not repeatable\n"); return 3; }
  // one graph: victim chain on the origin stream || aggressor chain on the forked stream
  hipEvent_t fork, join;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  hipGraph_t g;
  hipGraphExec_t ex;
  CK(hipStreamBeginCapture(sa, hipStreamCaptureModeThreadLocal));
  CK(hipEventRecord(fork, sa));
  CK(hipStreamWaitEvent(sb, fork, 0));
  for (int r = 0; r < REPS; ++r) {
    victim(dy + (size_t)r * hx.size(), sa);
    aggressor(sb);
    aggressor(sb);
  }
  CK(hipEventRecord(join, sb));
  CK(hipStreamWaitEvent(sa, join, 0));
  CK(hipStreamEndCapture(sa, &g));
  CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  int bad_replays = 0, bad_launches = 0, shown = 0;
  for (int i = 0; i < replays; ++i) {
    CK(hipGraphLaunch(ex, sa));
    CK(hipStreamSynchronize(sa));
    bool any = false;
    for (int r = 0; r < REPS; ++r) {
      CK(hipMemcpy(got.data(), dy + (size_t)r * hx.size(), got.size() * 2, hipMemcpyDeviceToHost));
      if (memcmp(ref.data(), got.data(), ref.size() * 2) != 0) {
        any = true;
        ++bad_launches;
        if (shown < 3) {
          ++shown;
          size_t nd = 0, first = 0;
          for (size_t k = 0; k < ref.size(); ++k)
            if (ref[k] != got[k]) { if (!nd) first = k; ++nd; }
          const size_t row = first / C, col = first % C, sl = (row / ((size_t)F * HW)) * HW + row % HW;   // slice (n, pixel)
          const size_t cv = C / 8, spb = 256 / cv, t = (sl % spb) * cv + col / 8;                            // gn_small_kernel: tid = pl * CV + cv
          printf("  replay %d launch %d: %zu of %zu elements differ; first at row %zu column %zu = thread %zu (lane %zu) of its block, channel quad %zu of its vector\n",
                 i, r, nd, ref.size(), row, col, t, t % 64, (col % 8) / 4);
        }
      }
    }
    bad_replays += any ? 1 : 0;
  }
  printf("pk_f32_repro (%s): C=%d, %d victim launches per replay beside %d aggressor launches: %d of %d replays differ (%d of %d launches)\n",
#ifdef PK_REPRO_NOPK
         "no packed fp32",
#else
         "packed fp32 allowed",
#endif
         C, REPS, 2 * REPS, bad_replays, replays, bad_launches, replays * REPS);
  return bad_replays ? 1 : 0;
}
