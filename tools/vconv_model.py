"""Lane-level numpy model of the index maps of vconv2d1d_kernel (mm-diffusion_amd/csrc/mmd_vconv.hip): the weight image of
mmd_vconv2d1d_pack, the halo DMA pieces (LDS row hh * 96 + f * 6 + ww, chunk swizzle hh & 3), the in-place GroupNorm slots, the
fragment addresses of both phases, the MFMA operand / accumulator layouts, the T image and the epilogue's row / record addressing -
restated with the SAME expressions as the kernel and checked against a direct two-conv computation.  Pure CPU (tests/test_host_cpu.py
runs it): an index error costs O(1) here instead of a GPU call.  Values are small integers so every sum is exact in fp32 and the
check is equality; the bf16 rounding points of the kernel are therefore not modelled (tests/test_vconv_gpu.py covers them)."""
import numpy as np

STAGE_B, WSLOT_B, WT_B, TPLANE_B = 36864, 24576, 16384, 36864


def pack_weights(Ws, Wt, Cin):
    """Ws [128, 9 * Cin], Wt [128, 384] (element arrays) -> the flat element image of mmd_vconv2d1d_pack (2-byte elements)."""
    nsp = (Cin // 32) * 3 * (WSLOT_B // 2)
    total = nsp + 6 * (WT_B // 2)
    out = np.zeros(total, dtype=Ws.dtype)
    i = np.arange(total)
    sp = i < nsp
    s1, r = i[sp] // (WSLOT_B // 2), i[sp] % (WSLOT_B // 2)
    c, dhi = s1 // 3, s1 % 3
    dwi, r2 = r // 4096, r % 4096
    co, pc, e = r2 >> 5, (r2 >> 3) & 3, r2 & 7
    lc = pc ^ ((co >> 2) & 3)
    out[sp] = Ws[co, (dhi * 3 + dwi) * Cin + c * 32 + lc * 8 + e]
    k = i[~sp] - nsp
    s2, r = k // (WT_B // 2), k % (WT_B // 2)
    tap, plane = s2 >> 1, s2 & 1
    co, pc, e = r >> 6, (r >> 3) & 7, r & 7
    lc = pc ^ ((co >> 1) & 7)
    out[~sp] = Wt[co, tap * 128 + plane * 64 + lc * 8 + e]
    return out


def mfma_32x32x16(Afrag, Bfrag, acc):
    """Afrag/Bfrag [64 lanes, 8]: lane (l31, half) holds row l31, k = 8 * half + 0..7.  acc [64, 16]: lane (j = l31, half), register
    4 q + jj = D[i = 8 q + 4 half + jj][j]."""
    A = np.zeros((32, 16), dtype=np.float64)
    B = np.zeros((16, 32), dtype=np.float64)
    for lane in range(64):
        l31, half = lane & 31, lane >> 5
        A[l31, 8 * half: 8 * half + 8] = Afrag[lane]
        B[8 * half: 8 * half + 8, l31] = Bfrag[lane]
    D = A @ B
    for lane in range(64):
        l31, half = lane & 31, lane >> 5
        for q in range(4):
            for jj in range(4):
                acc[lane, 4 * q + jj] += D[8 * q + 4 * half + jj, l31]


def run_block(X, ldx, Wf, bias_s, bias_t, gn, n, patch, H, W, Cin, Y, stats):
    """One block of the kernel.  X flat element array of rows [N * 16 * H * W, ldx]; gn = None or (a [S, Cin], b [S, Cin], act fn);
    Y [M, 128] output; stats dict {(record, quad): (sum, sq)}."""
    HW = H * W
    nchunk = Cin // 32
    PWc = W >> 2
    h0, w0 = (patch // PWc) * 4, (patch % PWc) * 4
    xbase = n * 16 * HW * ldx
    lds = np.zeros(2 * STAGE_B // 2 + 3 * WSLOT_B // 2, dtype=np.float64)      # element-addressed (2-byte elements)
    sA, sW = 0, 2 * STAGE_B // 2
    lanes = np.arange(64)
    tid_all = np.arange(512)

    def dma_halo(stage, c):
        for wave in range(8):
            for j in range(5):
                g = wave + 8 * j if j < 3 else (24 + wave if j == 3 else 32 + wave)
                if not (j < 4 or wave < 4):
                    continue
                row, pc = 16 * g + (lanes >> 2), lanes & 3
                hh, rem = row // 96, row % 96
                f, ww = rem // 6, rem % 6
                y, x = h0 - 1 + hh, w0 - 1 + ww
                ok = (g < 36) & (y >= 0) & (y < H) & (x >= 0) & (x < W)
                logical = pc ^ (hh & 3)
                for L in range(64):
                    dst = sA + (stage * STAGE_B + g * 1024 + L * 16) // 2
                    if ok[L]:
                        src = xbase + ((f[L] * H + y[L]) * W + x[L]) * ldx + logical[L] * 8 + c * 32
                        lds[dst: dst + 8] = X[src: src + 8]
                    else:
                        lds[dst: dst + 8] = 0.0

    def transform(stage, c, slots):
        if gn is None:
            return
        a, b, act = gn
        for i in slots:
            for tid in tid_all:
                if i == 4 and tid >= 256:
                    continue
                s = tid + 512 * i
                row, pc = s >> 2, s & 3
                hh, rem = row // 96, row % 96
                ww = rem % 6
                ok = 0 <= h0 - 1 + hh < H and 0 <= w0 - 1 + ww < W
                lc = pc ^ (hh & 3)
                q = sA + (stage * STAGE_B + tid * 16 + i * 8192) // 2
                if ok:
                    ch = c * 32 + lc * 8 + np.arange(8)
                    lds[q: q + 8] = act(lds[q: q + 8] * a[n, ch] + b[n, ch])

    def dma_ws(slot, s1):
        lds[sW + slot * WSLOT_B // 2: sW + (slot + 1) * WSLOT_B // 2] = Wf[s1 * WSLOT_B // 2: (s1 + 1) * WSLOT_B // 2]

    def dma_wt(slot, s2):
        base = nchunk * 3 * WSLOT_B // 2 + s2 * WT_B // 2
        lds[sW + slot * WSLOT_B // 2: sW + slot * WSLOT_B // 2 + WT_B // 2] = Wf[base: base + WT_B // 2]

    half, l31 = lanes >> 5, lanes & 31
    ph, pw = (l31 >> 2) & 3, l31 & 3
    acc = np.zeros((8, 2, 2, 64, 16))

    def rd(addr_bytes):                                        # [64] byte addresses -> [64, 8] elements
        return np.stack([lds[a // 2: a // 2 + 8] for a in addr_bytes])

    dma_halo(0, 0)
    transform(0, 0, range(5))
    for c in range(nchunk):
        if c + 1 < nchunk:
            dma_halo((c + 1) & 1, c + 1)
            transform((c + 1) & 1, c + 1, range(5))
        for J in range(3):
            dma_ws(J, c * 3 + J)
            for wave in range(8):
                wc, wr = wave & 1, wave >> 1
                kw1 = (l31 >> 2) & 3
                wl1 = (wc * 64 + l31) * 64
                key = (ph + J) & 3
                for k in range(6):
                    dwi, k2 = k >> 1, k & 1
                    fw = [rd(2 * sW + J * WSLOT_B + wl1 + dwi * 8192 + a * 2048 + ((2 * k2 + half) ^ kw1) * 16) for a in range(2)]
                    fa = []
                    for b in range(2):
                        rbB = ((ph + 1) * 96 + (wr * 4 + b * 2 + (l31 >> 4)) * 6 + pw + 1) * 64
                        st = 2 * sA + (c & 1) * STAGE_B + (J - 1) * (96 * 64) - 64
                        fa.append(rd(st + rbB + dwi * 64 + ((2 * k2 + half) ^ key) * 16))
                    for a in range(2):
                        for b in range(2):
                            mfma_32x32x16(fw[a], fa[b], acc[wave, a, b])
    # transition
    T = np.zeros(2 * TPLANE_B // 2)
    for wave in range(8):
        wc, wr = wave & 1, wave >> 1
        for a in range(2):
            for b in range(2):
                trow = wr * 64 + b * 32 + l31 + 16
                for j2 in range(2):
                    for L in range(64):
                        # permlane32_swap pairs: lane holds channels 16 j2 + 8 half .. + 8 of its row
                        own, other = acc[wave, a, b, L], acc[wave, a, b, L ^ 32]
                        if half[L] == 0:
                            v = np.concatenate([own[8 * j2: 8 * j2 + 4], other[8 * j2: 8 * j2 + 4]])
                        else:
                            v = np.concatenate([other[8 * j2 + 4: 8 * j2 + 8], own[8 * j2 + 4: 8 * j2 + 8]])
                        cl = a * 32 + 16 * j2 + 8 * half[L]
                        v = v + bias_s[wc * 64 + cl: wc * 64 + cl + 8]
                        dst = (wc * TPLANE_B + trow[L] * 128 + (((cl >> 3)) ^ ((trow[L] >> 1) & 7)) * 16) // 2
                        T[dst: dst + 8] = v
    lds[sA: sA + 2 * TPLANE_B // 2] = T
    acc[:] = 0.0
    for S2 in range(6):
        dma_wt(S2 % 3, S2)
        tap, plane = S2 >> 1, S2 & 1
        for wave in range(8):
            wc, wr = wave & 1, wave >> 1
            kw2 = (l31 >> 1) & 7
            wl2 = (wc * 64 + l31) * 128
            tl2 = (wr * 64 + l31 + 16) * 128
            for k in range(4):
                fw = [rd(2 * sW + (S2 % 3) * WSLOT_B + wl2 + a * 4096 + ((2 * k + half) ^ kw2) * 16) for a in range(2)]
                fa = [rd(2 * sA + plane * TPLANE_B + tl2 + (tap - 1) * 2048 + b * 4096 + ((2 * k + half) ^ kw2) * 16) for b in range(2)]
                for a in range(2):
                    for b in range(2):
                        mfma_32x32x16(fw[a], fa[b], acc[wave, a, b])
    # epilogue
    for wave in range(8):
        wc, wr = wave & 1, wave >> 1
        rec = (n * (HW >> 4) + patch) * 4 + wr
        for a in range(2):
            for j2 in range(2):
                u = np.zeros((64, 4))
                for b in range(2):
                    for L in range(64):
                        own, other = acc[wave, a, b, L], acc[wave, a, b, L ^ 32]
                        if half[L] == 0:
                            v = np.concatenate([own[8 * j2: 8 * j2 + 4], other[8 * j2: 8 * j2 + 4]])
                        else:
                            v = np.concatenate([other[8 * j2 + 4: 8 * j2 + 8], own[8 * j2 + 4: 8 * j2 + 8]])
                        col = wc * 64 + a * 32 + 16 * j2 + 8 * half[L]
                        v = v + bias_t[col: col + 8]
                        m = n * 16 * HW + (h0 + ph[L]) * W + w0 + pw[L] + (wr * 4 + b * 2 + (l31[L] >> 4)) * HW
                        Y[m, col: col + 8] = v
                        u[L] += [v[:4].sum(), v[4:].sum(), (v[:4] ** 2).sum(), (v[4:] ** 2).sum()]
                for hf in range(2):                            # half-wave totals land in lanes 16 / 17 (+ 32 hf): quad 0 / 1
                    tot = u[32 * hf: 32 * hf + 32].sum(axis=0)
                    col = wc * 64 + a * 32 + 16 * j2 + 8 * hf
                    stats[(rec, (col >> 2))] = (tot[0], tot[2])
                    stats[(rec, (col >> 2) + 1)] = (tot[1], tot[3])


def reference(X4, Ws, Wt, bias_s, bias_t, gn, Cin):
    """X4 [N, 16, H, W, Cin] -> [N, 16, H, W, 128] by two direct convolutions (zero padding of the normalised activation)."""
    N, F, H, W, _ = X4.shape
    x = X4.astype(np.float64)
    if gn is not None:
        a, b, act = gn
        x = act(x * a[:, None, None, None, :] + b[:, None, None, None, :])
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1), (0, 0)))
    t = np.zeros((N, F, H, W, 128))
    for dh in range(3):
        for dw in range(3):
            w = Ws[:, (dh * 3 + dw) * Cin: (dh * 3 + dw + 1) * Cin]
            t += xp[:, :, dh: dh + H, dw: dw + W, :] @ w.T
    t += bias_s
    tp = np.pad(t, ((0, 0), (1, 1), (0, 0), (0, 0), (0, 0)))
    y = np.zeros((N, F, H, W, 128))
    for df in range(3):
        y += tp[:, df: df + F] @ Wt[:, df * 128: (df + 1) * 128].T
    return y + bias_t


def check(N=1, H=8, W=8, Cin=64, with_gn=True, seed=0):
    rng = np.random.default_rng(seed)
    ldx = Cin + 8
    M = N * 16 * H * W
    X = np.zeros(M * ldx)
    X4 = rng.integers(-2, 3, size=(N, 16, H, W, Cin)).astype(np.float64)
    X.reshape(M, ldx)[:, :Cin] = X4.reshape(M, Cin)
    X.reshape(M, ldx)[:, Cin:] = 99.0                          # the pad columns must never be read
    Ws = rng.integers(-1, 2, size=(128, 9 * Cin)).astype(np.float64)
    Wt = rng.integers(-1, 2, size=(128, 384)).astype(np.float64)
    bias_s = rng.integers(-2, 3, size=128).astype(np.float64)
    bias_t = rng.integers(-2, 3, size=128).astype(np.float64)
    gn = None
    if with_gn:
        a = rng.integers(1, 3, size=(N, Cin)).astype(np.float64)
        b = rng.integers(-1, 2, size=(N, Cin)).astype(np.float64)
        gn = (a, b, lambda v: np.where(v > 0, v, 2 * v))       # any elementwise map with act(0) != 0 would expose transformed padding: act(b) != 0
    Wf = pack_weights(Ws, Wt, Cin)
    Y = np.full((M, 128), np.nan)
    stats = {}
    for n in range(N):
        for patch in range((H // 4) * (W // 4)):
            run_block(X, ldx, Wf, bias_s, bias_t, gn, n, patch, H, W, Cin, Y, stats)
    ref = reference(X4, Ws, Wt, bias_s, bias_t, gn, Cin).reshape(M, 128)
    assert not np.isnan(Y).any(), "rows never written"
    assert np.array_equal(Y, ref), f"max abs diff {np.abs(Y - ref).max()}"
    # records: every sample's records sum to the sample's totals per quad
    nrec = M // 64
    S = np.zeros((nrec, 32, 2))
    assert len(stats) == nrec * 32, (len(stats), nrec * 32)
    for (r, q), (s, sq) in stats.items():
        S[r, q] = (s, sq)
    per = 16 * H * W // 64
    for n in range(N):
        yn = ref[n * 16 * H * W: (n + 1) * 16 * H * W].reshape(-1, 32, 4)
        assert np.array_equal(S[n * per: (n + 1) * per, :, 0].sum(0), yn.sum(axis=(0, 2)))
        assert np.array_equal(S[n * per: (n + 1) * per, :, 1].sum(0), (yn ** 2).sum(axis=(0, 2)))
    return True


if __name__ == "__main__":
    for kw in (dict(with_gn=False), dict(with_gn=True), dict(N=2, H=4, W=8, Cin=32, with_gn=True, seed=3)):
        check(**kw)
        print("vconv index model ok:", kw)
