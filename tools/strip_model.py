#!/usr/bin/env python3
"""Lane-level model of conv1x1_strip_kernel (mmd_gemm.hip, tile 131): the index maps of the kernel restated in numpy and run on
small problems, so that the DMA image / fragment / epilogue / statistics addressing can be checked on a machine without a GPU
(tests/test_host_cpu.py).  The MFMA and v_permlane32_swap semantics are the ones the tiled kernels and the attention kernels are
verified with on the GPU:

  v_mfma_f32_32x32x16_bf16  D[i][j] += sum_k A[i][k] B[k][j];  lane (l31, half) supplies A[i = l31][8 half .. +8] and
                            B[8 half .. +8][j = l31] and holds D[8 q + 4 half + jj][l31] in acc[4 q + jj]
  v_permlane32_swap vdst, src   lanes 32-63 of vdst <-> lanes 0-31 of src
"""
import numpy as np


def xcd_remap(bid, nwg):
    q, r, xcd = nwg >> 3, nwg & 7, bid & 7
    return (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + (bid >> 3)


def pick_nsplit(M, Cout, BR, CC):
    rowblocks, nch = -(-M // BR), Cout // CC
    nsplit = 1
    for d in range(1, min(nch, 16) + 1):
        if nch % d == 0:
            nsplit = d
            if rowblocks * d >= 448:
                break
    return nsplit


def halfwave_sum16(u):
    """u [64 lanes, 16] -> [64] : lane ends with the half-wave total of u[:, l31 >> 1] (recursive halving, kernel order)."""
    lanes = np.arange(64)
    l31 = lanes & 31

    def step(a, bit, xor):
        n = a.shape[1] // 2
        hi = (l31 & bit) != 0
        keep = np.where(hi[:, None], a[:, n:], a[:, :n])
        send = np.where(hi[:, None], a[:, :n], a[:, n:])
        return (keep + send[lanes ^ xor]).astype(np.float32)
    a8 = step(u.astype(np.float32), 16, 16)
    a4 = step(a8, 8, 8)
    a2 = step(a4, 4, 4)
    a1 = step(a2, 2, 2)[:, 0]
    return (a1 + a1[lanes ^ 1]).astype(np.float32)


def bf16_round(x):
    """fp32 -> bf16 (RNE) -> fp32, as the kernel's v_cvt_pk_bf16_f32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def run(M, K, Cout, bias=True, residual=False, gn=None, want_stats=False, seed=0, taps=None, dims=(1, 1, 1)):
    """Whole launch on random bf16-representable data.  K = ntaps * Cin; taps = [(o0, o1, o2)] over dims (D0, D1, D2), zero padded.
    Returns (Y model, Y reference, stats model, stats reference, touched, nsplit)."""
    rng = np.random.default_rng(seed)
    KS = K // 64
    RF, CC = (2, 64) if K <= 256 else (1, 32)
    BR = 128 * RF
    taps = taps or [(0, 0, 0)]
    Cin = K // len(taps)
    A = bf16_round(rng.standard_normal((M, Cin)).astype(np.float32))
    W = bf16_round((rng.standard_normal((Cout, K)) / np.sqrt(K)).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32) if bias else None
    R = bf16_round(rng.standard_normal((M, Cout)).astype(np.float32)) if residual else None
    g = None
    X = A
    if len(taps) > 1:                                            # im2row reference: X[m] = [A[src(m, tap)] or 0 for tap in taps]
        D0, D1, D2 = dims
        m = np.arange(M)
        q2, q1, q0 = m % D2, (m // D2) % D1, (m // (D1 * D2)) % D0
        cols = []
        for (o0, o1, o2) in taps:
            ok = (q0 + o0 >= 0) & (q0 + o0 < D0) & (q1 + o1 >= 0) & (q1 + o1 < D1) & (q2 + o2 >= 0) & (q2 + o2 < D2)
            src = np.clip(m + o0 * D1 * D2 + o1 * D2 + o2, 0, M - 1)
            cols.append(np.where(ok[:, None], A[src], 0.0))
        X = np.concatenate(cols, axis=1).astype(np.float32)
    if gn is not None:
        rows, act = gn
        S = M // rows
        ga = (1.0 + 0.1 * rng.standard_normal((S, K))).astype(np.float32)
        gb = (0.1 * rng.standard_normal((S, K))).astype(np.float32)
        g = (ga, gb, rows, act)
        sl = np.arange(M) // rows
        y = A * ga[sl] + gb[sl]
        if act:
            y = y / (1.0 + np.exp(-y))
        X = bf16_round(y.astype(np.float32))
    nsplit = pick_nsplit(M, Cout, BR, CC)
    nwg = -(-M // BR) * nsplit
    Y = np.full((M, Cout), np.nan, np.float32)
    touched = np.zeros((M, Cout), np.int32)
    stats = np.full((M // 64 if M % 64 == 0 else 1, Cout, 2), np.nan, np.float32)
    for bid in range(nwg):
        _run_block_chunks(bid, nwg, nsplit, KS, RF, CC, A, W, b, R, M, Cout, g, want_stats, Y, stats, touched, taps, dims)
    ref = X.astype(np.float64) @ W.astype(np.float64).T
    if b is not None:
        ref = ref + b
    if R is not None:
        ref = ref + R
    sref = None
    if want_stats:
        yq = Y.reshape(M // 64, 64, Cout).astype(np.float64)
        sref = np.stack([yq.sum(1), (yq * yq).sum(1)], axis=-1)
    return Y, ref, stats, sref, touched, nsplit


def _run_block_chunks(bid, nwg, nsplit, KS, RF, CC, A, W, bias, R, M, Cout, gn, want_stats, Y, stats, touched, taps=((0, 0, 0),), dims=(1, 1, 1)):
    """One workgroup.  A [M, K], W [Cout, K] (bf16-representable values), gn = (a [S, K], b [S, K], rows per slice, act).  The
    two-stage ring is modelled faithfully: chunk ci is computed from stage ci & 1 after chunk ci + 1 has been issued into the other
    stage (an LDS image that was never written reads back NaN)."""
    K, NCG, BR, NA, GP = 64 * KS, 4 * KS, 128 * RF, CC // 32, CC // 32
    PLANE_B, STAGE_B = CC * 128, KS * CC * 128
    wgid = xcd_remap(bid, nwg)
    sp, mt = wgid % nsplit, wgid // nsplit
    Cs = Cout // nsplit
    cbase, nchunk, m0 = sp * Cs, Cs // CC, mt * BR
    lanes = np.arange(64)
    half, l31, lrow, pc = lanes >> 5, lanes & 31, lanes >> 3, lanes & 7
    Wb = W.astype(np.float32)
    lds = np.full(2 * STAGE_B // 2, np.nan, dtype=np.float32)

    def issue(stage, ci):
        for wave in range(4):
            for pl in range(KS):
                for ih in range(GP):
                    row = 8 * (ih * 4 + wave) + lrow
                    logical = pc ^ ((row >> 1) & 7)
                    dst = (stage * STAGE_B + pl * PLANE_B + (ih * 4 + wave) * 1024) // 2 + lanes * 8
                    for ln in range(64):
                        lds[dst[ln]: dst[ln] + 8] = Wb[cbase + ci * CC + row[ln], pl * 64 + logical[ln] * 8: pl * 64 + logical[ln] * 8 + 8]

    sBias = np.zeros(Cs, np.float32) if bias is None else bias[cbase: cbase + Cs].astype(np.float32)
    waves = []
    for wave in range(4):
        rows = [m0 + wave * 32 * RF + f * 32 + l31 for f in range(RF)]
        rok = [r < M for r in rows]
        rowc = [np.where(ok, r, M - 1) for r, ok in zip(rows, rok)]
        if len(taps) == 1:
            xa = [[A[rowc[f][:, None], (16 * cg + 8 * half)[:, None] + np.arange(8)[None, :]].astype(np.float32) for cg in range(NCG)]
                  for f in range(RF)]
        else:                                                    # a 64-channel plane lies inside one tap: shifted row or the zero row
            Cin, (D0, D1, D2) = A.shape[1], dims
            xa = [[None] * NCG for _ in range(RF)]
            for f in range(RF):
                q2, q1, q0 = rowc[f] % D2, (rowc[f] // D2) % D1, (rowc[f] // (D1 * D2)) % D0
                for pl in range(KS):
                    tap = (64 * pl) // Cin
                    ci0 = 64 * pl - tap * Cin
                    o0, o1, o2 = taps[tap]
                    ok = (q0 + o0 >= 0) & (q0 + o0 < D0) & (q1 + o1 >= 0) & (q1 + o1 < D1) & (q2 + o2 >= 0) & (q2 + o2 < D2)
                    src = np.where(ok, rowc[f] + o0 * D1 * D2 + o1 * D2 + o2, 0)
                    for c in range(4):
                        ch = (ci0 + 16 * c + 8 * half)[:, None] + np.arange(8)[None, :]
                        xa[f][4 * pl + c] = np.where(ok[:, None], A[src[:, None], ch], 0.0).astype(np.float32)
        if gn is not None:
            ga, gb, gnr, act = gn
            s0, S = m0 // gnr, ga.shape[0]
            table = np.zeros((2, 2, K), np.float32)
            for sl in range(2):
                sidx = min(s0 + sl, S - 1)
                table[sl, 0], table[sl, 1] = ga[sidx], gb[sidx]
            for f in range(RF):
                gs = np.minimum(rowc[f] // gnr - s0, 1)
                for cg in range(NCG):
                    ch = (16 * cg + 8 * half)[:, None] + np.arange(8)[None, :]
                    y = xa[f][cg] * table[gs[:, None], 0, ch] + table[gs[:, None], 1, ch]
                    if act:
                        y = y / (1.0 + np.exp(-y))
                    xa[f][cg] = bf16_round(y.astype(np.float32))
        waves.append((rows, rok, rowc, xa))
    issue(0, 0)
    xsw = (l31 >> 1) & 7
    sRec = np.full((2, 4, NA * 2, 32), np.nan, np.float32)     # RF = 1: half-record partials, double-buffered by chunk parity
    for ci in range(nchunk):
        st = ci & 1
        if ci + 1 < nchunk:
            issue(st ^ 1, ci + 1)
        own = {}
        for wave in range(4):
            rows, rok, rowc, xa = waves[wave]
            wave_ok = m0 + wave * 32 * RF < M
            rec = (m0 + wave * 32 * RF) // 64
            for a in range(NA):
                cb = ci * CC + a * 32
                acc = [np.zeros((64, 16), np.float32) for _ in range(RF)]
                for pl in range(KS):
                    for c in range(4):
                        off = (st * STAGE_B + pl * PLANE_B + (a * 32 + l31) * 128 + ((2 * c + half) ^ xsw) * 16) // 2
                        fw = np.stack([lds[o: o + 8] for o in off])
                        Amat = np.zeros((32, 16), np.float32)
                        for h in range(2):
                            Amat[l31[half == h], 8 * h: 8 * h + 8] = fw[half == h]
                        for f in range(RF):
                            fb = xa[f][4 * pl + c]
                            Bmat = np.zeros((16, 32), np.float32)
                            for h in range(2):
                                Bmat[8 * h: 8 * h + 8, l31[half == h]] = fb[half == h].T
                            D = Amat @ Bmat
                            for q in range(4):
                                for h in range(2):
                                    acc[f][half == h, 4 * q: 4 * q + 4] += D[8 * q + 4 * h: 8 * q + 4 * h + 4, :].T
                for j2 in range(2):
                    col = cbase + cb + 16 * j2 + 8 * half
                    u = np.zeros((64, 16), np.float32)
                    for f in range(RF):
                        vd = acc[f][:, 8 * j2: 8 * j2 + 4].copy()
                        vs = acc[f][:, 8 * j2 + 4: 8 * j2 + 8].copy()
                        nd, ns = vd.copy(), vs.copy()
                        nd[32:] = vs[:32]
                        ns[:32] = vd[32:]
                        v = np.concatenate([nd, ns], axis=1)
                        v = v + sBias[(cb + 16 * j2 + 8 * half)[:, None] + np.arange(8)[None, :]]
                        if R is not None:
                            v = v + R[rowc[f][:, None], col[:, None] + np.arange(8)[None, :]].astype(np.float32)
                        pk = bf16_round(v.astype(np.float32))
                        for ln in range(64):
                            if rok[f][ln]:
                                Y[rows[f][ln], col[ln]: col[ln] + 8] = pk[ln]
                                touched[rows[f][ln], col[ln]: col[ln] + 8] += 1
                        u[:, 0::2] += pk
                        u[:, 1::2] += pk * pk
                    if want_stats:
                        tot = halfwave_sum16(u)
                        if RF == 1:
                            for ln in range(64):
                                if (l31[ln] & 1) == 0:
                                    sRec[ci & 1, wave, a * 2 + j2, half[ln] * 16 + (l31[ln] >> 1)] = tot[ln]
                            own[(wave, j2)] = (tot, col, rec, wave_ok)
                        else:
                            for ln in range(64):
                                if wave_ok and (l31[ln] & 1) == 0:
                                    stats.reshape(-1)[(rec * Cout + col[ln]) * 2 + (l31[ln] >> 1)] = tot[ln]
        # after the chunk's barrier: the even wave of a pair adds its partner's half-record and stores
        if want_stats and RF == 1:
            for (wave, j2), (tot, col, rec, wave_ok) in own.items():
                if wave_ok and wave % 2 == 0:
                    for ln in range(64):
                        if (l31[ln] & 1) == 0:
                            other = sRec[ci & 1, wave + 1, (NA - 1) * 2 + j2, half[ln] * 16 + (l31[ln] >> 1)]
                            stats.reshape(-1)[(rec * Cout + col[ln]) * 2 + (l31[ln] >> 1)] = np.float32(tot[ln] + other)


if __name__ == "__main__":
    for (M, K, Cout, res, gn, st, taps, dims) in [(512, 128, 128, True, (256, True), True, None, (1, 1, 1)),
                                                  (320, 256, 192, False, None, False, None, (1, 1, 1)),
                                                  (384, 384, 96, False, (128, False), True, None, (1, 1, 1)),
                                                  (512, 384, 64, True, None, True, [(-1, 0, 0), (0, 0, 0), (1, 0, 0)], (4, 128, 1)),
                                                  (256, 512, 96, False, None, True, None, (1, 1, 1))]:
        Y, ref, stats, sref, touched, nsplit = run(M, K, Cout, residual=res, gn=gn, want_stats=st, taps=taps, dims=dims)
        err = np.abs(Y - ref).max() / np.abs(ref).max()
        print(f"M={M} K={K} N={Cout} nsplit={nsplit}: every element written once: {bool((touched == 1).all())}, max rel err {err:.2e}"
              + (f", stats max rel err {np.abs(stats - sref).max() / np.abs(sref).max():.2e}" if st else ""))
