"""Lane-level model of the head GEMM (mm-diffusion_amd/csrc/mmd_misc.hip: head_gemm_kernel<8>) + its host-side weight image
(ops.head_gemm_pack): checks, on the CPU, that

  * the image [hl][ob][cg][half][l31][8] read as `sW + (((hl * 3 + ob) * KS + cg) * 64 + lane) * 16` hands lane (l31, half) the A
    operand W[32 ob + l31][16 cg + 8 half + e] of v_mfma_f32_32x32x16_bf16,
  * the activation fragment a lane loads (row m = 128 g + 32 wave + l31, channels 16 cg + 8 half + e; GroupNorm affine from the slice
    table at the same channel index, SiLU) is the B operand B[k = 8 half + e][n = l31],
  * the accumulator register 4 q + j of lane (l31, half) is output o = 32 ob + 8 q + 4 half + j of row m, stored at P[o][m],
  * running the hi and the lo image against the same activations sums to (hi + lo) . act - the fp32 weight to 2^-16,

for several GroupNorm slices per launch, several row groups per block and NO < 96 (S x gn_rows == M with gn_rows % 128 == 0, so a
launch never has a ragged row group; the kernel's clamp of rows past M is modelled but idle).  bf16 x bf16 products and their sums
are exact in float64, so the comparison is exact.
Model of the instruction as in tools/tattn_model.py.  `check()` is run by tests/test_host_cpu.py."""
import numpy as np

KS, C, L = 8, 128, 64


def bf16_round(x):
    """fp32 -> bf16 (round to nearest even) -> fp32, numpy."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def mfma(A, B, acc):
    """A, B: [64 lanes][8]; acc [64][16]: A lane (m = l % 32, h) holds A[m][8 h + e], B lane holds B[8 h + e][n = l % 32],
    D lane holds D[8 (i // 4) + 4 h + i % 4][n]."""
    Am, Bm = np.zeros((32, 16)), np.zeros((16, 32))
    for l in range(L):
        Am[l % 32, 8 * (l // 32): 8 * (l // 32) + 8] = A[l]
        Bm[8 * (l // 32): 8 * (l // 32) + 8, l % 32] = B[l]
    Dm = Am @ Bm
    out = acc.copy()
    for l in range(L):
        for i in range(16):
            out[l, i] += Dm[8 * (i // 4) + 4 * (l // 32) + i % 4, l % 32]
    return out


def pack(Wm):
    """ops.head_gemm_pack on the [ntaps * Co, Cin] matrix: zero-padded to 96 outputs, split into (hi, lo) bf16, laid out
    [hl][ob][cg][half][l31][8] and flattened (the kernel addresses it in units of 8 elements = 16 bytes)."""
    full = np.zeros((96, C), np.float32)
    full[:Wm.shape[0]] = Wm
    hi = bf16_round(full)
    lo = bf16_round(full - hi)
    img = np.stack([hi, lo]).reshape(2, 3, 32, C // 16, 2, 8).transpose(0, 1, 3, 4, 2, 5)
    return np.ascontiguousarray(img).reshape(-1, 8), hi, lo


def silu(y):
    return y / (1.0 + np.exp(-y))


def run(M, NO, S, seed=0, per_block=2, act=True):
    rng = np.random.default_rng(seed)
    assert M % S == 0 and (M // S) % 128 == 0
    gn_rows = M // S
    x = bf16_round(rng.standard_normal((M, C)).astype(np.float32))
    ga = (1 + 0.2 * rng.standard_normal((S, C))).astype(np.float32)
    gb = (0.3 * rng.standard_normal((S, C))).astype(np.float32)
    Wm = (rng.standard_normal((NO, C)) / np.sqrt(C)).astype(np.float32)
    sW, hi, lo = pack(Wm)
    P = np.full((NO, M), np.nan)
    ngroups = (M + 127) // 128
    nblocks = (ngroups + per_block - 1) // per_block
    for bid in range(nblocks):
        for g in range(bid * per_block, min((bid + 1) * per_block, ngroups)):
            sl = (g * 128) // gn_rows
            for wave in range(4):
                rows = [g * 128 + wave * 32 + (l % 32) for l in range(L)]
                ok = [m < M for m in rows]
                xa = []
                for cg in range(KS):
                    frag = np.zeros((L, 8))
                    for l in range(L):
                        mc, half = (rows[l] if ok[l] else M - 1), l // 32
                        ch = 16 * cg + 8 * half + np.arange(8)
                        y = x[mc, ch] * ga[sl, ch] + gb[sl, ch]            # sGN[cg * 16 + half * 8 + e] and + C
                        frag[l] = bf16_round((silu(y) if act else y).astype(np.float32))
                    xa.append(frag)
                for ob in range(3):
                    if ob * 32 >= NO:
                        break
                    acc = np.zeros((L, 16))
                    for hl in range(2):
                        for cg in range(KS):
                            fw = np.stack([sW[((hl * 3 + ob) * KS + cg) * 64 + l] for l in range(L)])
                            acc = mfma(fw, xa[cg], acc)
                    for l in range(L):
                        if not ok[l]:
                            continue
                        for q in range(4):
                            for j in range(4):
                                o = ob * 32 + 8 * q + 4 * (l // 32) + j
                                if o < NO:
                                    assert np.isnan(P[o, rows[l]]), "an element of P written twice"
                                    P[o, rows[l]] = acc[l, 4 * q + j]
    assert not np.isnan(P).any(), "an element of P was never written"
    sl_of = (np.arange(M) // gn_rows)
    y = x * ga[sl_of] + gb[sl_of]
    a = bf16_round((silu(y) if act else y).astype(np.float32)).astype(np.float64)
    ref = (hi[:NO].astype(np.float64) + lo[:NO].astype(np.float64)) @ a.T
    err = np.abs(P - ref).max() / np.abs(ref).max()
    werr = np.abs(hi[:NO].astype(np.float64) + lo[:NO] - Wm).max() / np.abs(Wm).max()
    return err, werr


def check(seed=0, verbose=False):
    worst = 0.0
    for M, NO, S, pb in ((512, 81, 2, 2), (384, 96, 3, 1), (256, 27, 1, 3), (640, 40, 5, 2)):
        err, werr = run(M, NO, S, seed=seed, per_block=pb)
        if verbose:
            print(f"M={M} NO={NO} slices={S} per_block={pb}: planes vs (hi + lo) . act {err:.2e}; hi + lo vs fp32 weight {werr:.2e}")
        assert werr < 2.0 ** -15
        worst = max(worst, err)
    return worst


if __name__ == "__main__":
    print("worst", check(verbose=True))
