"""Lane-level model of the fused temporal-attention block kernel (mm-diffusion_amd/csrc/mmd_tattn.hip): checks, on the CPU, the register
algebra the kernel relies on - that the accumulators of one v_mfma_f32_32x32x16_bf16 can be packed straight into the operands of the
next one, with no shuffle and no LDS transpose, for the whole chain

    x (B operand) -> q, k (A = W, B = x)  -> S^T = k q^T (A = k, B = q) -> softmax in-lane -> P (B operand)
                  -> v^T (A = x, B = W: the SAME two fragments, swapped)  -> O^T = v^T P (A = v^T, B = P)
                  -> out = Wp' O (A = Wp with its K columns permuted at pack time, B = O)

Model of the instruction (lane l: n = l % 32, h = l // 32):
    A operand: lane holds A[m = l % 32][k = 8 h + e], e < 8       B operand: lane holds B[k = 8 h + e][n = l % 32]
    D / C    : lane holds D[m = 8 (i // 4) + 4 h + i % 4][n], i < 16
A wave fragment = 32 rows = 2 pixels x 16 frames (row r: pixel r // 16, frame r % 16); one head of 64 channels, C = 256 here.
`check()` is run by tests/test_host_cpu.py."""
import numpy as np

L = 64


def mfma(A, B, Cacc):
    """A, B: [64 lanes][8]; Cacc: [64][16] -> D (same layout)."""
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(L):
        for e in range(8):
            Am[l % 32, 8 * (l // 32) + e] = A[l, e]
            Bm[8 * (l // 32) + e, l % 32] = B[l, e]
    Dm = Am @ Bm
    D = Cacc.copy()
    for l in range(L):
        for i in range(16):
            D[l, i] += Dm[8 * (i // 4) + 4 * (l // 32) + i % 4, l % 32]
    return D


def acc_to_operands(acc):
    """16 accumulators of a lane -> two 8-element operand registers: step s holds i = 8 s .. 8 s + 7.  The element (s, e) of lane half h
    carries the C-row index m = 16 s + 8 (e // 4) + 4 h + e % 4 - the SAME map for every tensor packed this way, which is all a
    contraction over that index needs."""
    return [acc[:, 8 * s: 8 * s + 8].copy() for s in range(2)]


def crow(s, h, e):
    return 16 * s + 8 * (e // 4) + 4 * h + e % 4


def std_k(h, e):
    return 8 * h + e


def check(seed=0, verbose=False, C=256):
    rng = np.random.default_rng(seed)
    HEADS = 4
    CH = C // HEADS                                        # 64 / 96 / 128: 2 / 3 / 4 sub-tiles of 32 channels per head
    x = rng.standard_normal((32, C))                       # normalised rows of the fragment (the GroupNorm is elementwise bookkeeping)
    Wqkv = rng.standard_normal((3 * C, C)) / np.sqrt(C)
    bqkv = rng.standard_normal(3 * C) / 4
    Wp = rng.standard_normal((C, C)) / np.sqrt(C)
    scale = CH ** -0.5

    # ---- reference
    qkv = x @ Wqkv.T + bqkv
    O_ref = np.zeros((32, C))
    for p in range(2):
        rows = slice(16 * p, 16 * p + 16)
        for hd in range(HEADS):
            q = qkv[rows, hd * CH:(hd + 1) * CH]
            k = qkv[rows, C + hd * CH: C + (hd + 1) * CH]
            v = qkv[rows, 2 * C + hd * CH: 2 * C + (hd + 1) * CH]
            s = (q @ k.T) * scale
            pr = np.exp(s - s.max(axis=1, keepdims=True))
            pr /= pr.sum(axis=1, keepdims=True)
            O_ref[rows, hd * CH:(hd + 1) * CH] = pr @ v
    out_ref = O_ref @ Wp.T

    # ---- register model.  x fragments: k-step cg (16 channels), lane (row r = l % 32, half h): channels 16 cg + 8 h + e
    xf = [np.array([[x[l % 32, 16 * cg + std_k(l // 32, e)] for e in range(8)] for l in range(L)]) for cg in range(C // 16)]

    def wfrag(W, row0, cg):
        """weight fragment from the LDS image: lane (row l % 32 of the 32-row sub-tile, half h) holds K columns 16 cg + 8 h + e"""
        return np.array([[W[row0 + l % 32, 16 * cg + std_k(l // 32, e)] for e in range(8)] for l in range(L)])

    o_ops = {}                                             # (head, sub-tile a, step s) -> B operand of the projection
    for hd in range(HEADS):
        qo, ko, vo = {}, {}, {}
        for a in range(CH // 32):
            for name, third, store, swap in (("q", 0, qo, False), ("k", 1, ko, False), ("v", 2, vo, True)):
                row0 = third * C + hd * CH + 32 * a
                acc = np.zeros((L, 16))
                for cg in range(C // 16):
                    w = wfrag(Wqkv, row0, cg)
                    acc = mfma(xf[cg], w, acc) if swap else mfma(w, xf[cg], acc)
                # bias: q / k lanes hold C-rows = channels (a vector over i), v^T lanes hold C-column = channel l % 32 (a lane scalar)
                for l in range(L):
                    for i in range(16):
                        ch = l % 32 if swap else 8 * (i // 4) + 4 * (l // 32) + i % 4
                        acc[l, i] += bqkv[row0 + ch]
                store[a] = acc_to_operands(acc)
        # S^T[key][query]: A = k (lane = key row), B = q (lane = query row); both carry channel crow(s, h, e) of sub-tile a
        sT = np.zeros((L, 16))
        for a in range(CH // 32):
            for s in range(2):
                sT = mfma(ko[a][s], qo[a][s], sT)
        # lane (query r = l % 32 of pixel p = r // 16) holds keys m = 8 (i // 4) + 4 h + i % 4; its own pixel's keys are i in [8 p, 8 p + 8)
        P = np.zeros((L, 16))
        for l in range(L):
            p = (l % 32) // 16
            own = sT[l, 8 * p: 8 * p + 8] * scale
            other = sT[l ^ 32, 8 * p: 8 * p + 8] * scale          # the partner half-wave holds the other 8 keys of the pixel
            mx = max(own.max(), other.max())
            den = np.exp(own - mx).sum() + np.exp(other - mx).sum()
            P[l, 8 * p: 8 * p + 8] = np.exp(own - mx) / den
        po = acc_to_operands(P)                            # step s = keys of pixel s; zero for the other pixel's queries
        for a in range(CH // 32):
            oT = np.zeros((L, 16))
            for s in range(2):
                oT = mfma(vo[a][s], po[s], oT)                   # A = v^T (lane = channel, k = key crow), B = P (lane = query, k = key crow)
            ops_ = acc_to_operands(oT)                           # lane = query row, elements = channel crow(s, h, e) of sub-tile a
            for s in range(2):
                o_ops[(hd, a, s)] = ops_[s]
    # projection: K step (hd, a, s) covers channels CH hd + 32 a + crow(s, h, e) = a 16-channel block [CH hd + 32 a + 16 s, + 16) in the
    # order pi(h, e) = 8 (e // 4) + 4 h + e % 4; the packed Wp stores, at standard position 8 h + e, the column of that channel
    perm = np.zeros(C, dtype=np.int64)
    for blk in range(C // 16):
        for h in range(2):
            for e in range(8):
                perm[16 * blk + std_k(h, e)] = 16 * blk + 8 * (e // 4) + 4 * h + e % 4
    Wp_packed = Wp[:, perm]
    out = np.zeros((32, C))
    for mt in range(C // 32):
        acc = np.zeros((L, 16))
        for hd in range(HEADS):
            for a in range(CH // 32):
                for s in range(2):
                    cg = (CH * hd + 32 * a + 16 * s) // 16
                    acc = mfma(wfrag(Wp_packed, 32 * mt, cg), o_ops[(hd, a, s)], acc)
        for l in range(L):
            for i in range(16):
                out[l % 32, 32 * mt + 8 * (i // 4) + 4 * (l // 32) + i % 4] = acc[l, i]
    err = np.abs(out - out_ref).max() / np.abs(out_ref).max()
    if verbose:
        print("max relative error of the register model against the reference:", err)
    assert err < 1e-12, err
    return err


if __name__ == "__main__":
    for C_ in (256, 384, 512):
        check(verbose=True, C=C_)
