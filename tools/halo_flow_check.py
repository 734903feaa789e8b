"""Sequential emulation of conv_gemm_halo_kernel's whole block flow (buffers, chunk/tap loop, halo issue schedule, W offsets,
epilogue row mapping) against a direct convolution.  DMA is modelled as landing at the s_waitcnt that follows its issue."""
import numpy as np
rng = np.random.default_rng(1)
PH, PW, HWD, HR, HG, HJ = 8, 16, 18, 180, 23, 6

def conv_ref(A, W, taps, D0, D1, D2, Cin, Cout):
    M = D0 * D1 * D2
    Y = np.zeros((M, Cout), np.float64)
    for m in range(M):
        d0, h, w = m // (D1 * D2), (m // D2) % D1, m % D2
        for t, (o0, o1, o2) in enumerate(taps):
            hh, ww = h + o1, w + o2
            if 0 <= hh < D1 and 0 <= ww < D2:
                Y[m] += W[:, t * Cin:(t + 1) * Cin] @ A[(d0 * D1 + hh) * D2 + ww]
    return Y

def block(A, W, taps, D0, D1, D2, Cin, Cout, mt, nt):
    ntaps = len(taps)
    TW = D2 // PW; tpf = TW * (D1 // PH)
    d0, trem = divmod(mt, tpf)
    h0, w0 = (trem // TW) * PH, (trem % TW) * PW
    n0 = nt * 128
    nchunk = Cin // 64
    nit = nchunk * ntaps
    sA = np.full((2, HG * 8, 64), np.nan)      # logical (unswizzled) view: the swizzle is checked by halo_index_check.py
    sW = np.full((2, 128, 64), np.nan)
    pending = []                                # DMA writes that land at the next wait
    def issue_w(buf, t, c):
        def land():
            for row in range(128):
                co = n0 + row
                sW[buf, row] = W[co, t * Cin + c * 64: t * Cin + c * 64 + 64] if co < Cout else 0.0
        pending.append(land)
    def issue_h(buf, c, j):
        def land():
            for wave in range(4):
                g = wave + 4 * j
                if g >= HG: continue
                for lrow in range(8):
                    r = 8 * g + lrow
                    hr, hc = divmod(r, HWD)
                    hh, ww = h0 - 1 + hr, w0 - 1 + hc
                    ok = r < HR and 0 <= hh < D1 and 0 <= ww < D2
                    sA[buf, r] = A[(d0 * D1 + hh) * D2 + ww, c * 64:c * 64 + 64] if ok else 0.0
        pending.append(land)
    def wait():
        for f in pending: f()
        pending.clear()
    acc = np.zeros((128, 128))                  # [co_local][pixel]
    def compute(bufw, bufa, t):
        o1, o2 = taps[t][1], taps[t][2]
        toff = o1 * HWD + o2
        for px in range(128):
            r = ((px >> 4) + 1) * HWD + (px & 15) + 1 + toff
            a = sA[bufa, r]
            assert not np.isnan(a).any(), ("A not landed", bufa, r)
            assert not np.isnan(sW[bufw]).any()
            acc[:, px] += sW[bufw] @ a
    for j in range(HJ): issue_h(0, 0, j)
    issue_w(0, 0, 0)
    wait()
    c = t = 0
    for it in range(nit - 1):
        tn, cn = t + 1, c
        if tn == ntaps: tn, cn = 0, c + 1
        # poison the buffers about to be overwritten only at landing time (they may still be read by this step's compute? no:
        # W buf (it+1)&1 was read in step it-1; A buf (c+1)&1 was read during chunk c-1)
        issue_w((it + 1) & 1, tn, cn)
        if c + 1 < nchunk:
            for j in range(HJ):
                if j % ntaps == t: issue_h((c + 1) & 1, c + 1, j)
        compute(it & 1, c & 1, t)
        wait()
        t, c = tn, cn
    compute((nit - 1) & 1, c & 1, t)
    out = {}
    for ml in range(128):
        m = (d0 * D1 + h0 + (ml >> 4)) * D2 + w0 + (ml & 15)
        out[m] = acc[:, ml]
    return n0, out

def check(D0, D1, D2, Cin, Cout, taps):
    M = D0 * D1 * D2
    A = rng.standard_normal((M, Cin)); W = rng.standard_normal((Cout, Cin * len(taps)))
    ref = conv_ref(A, W, taps, D0, D1, D2, Cin, Cout)
    Y = np.full((M, Cout), np.nan)
    Nt = (Cout + 127) // 128
    for mt in range(M // 128):
        for nt in range(Nt):
            n0, out = block(A, W, taps, D0, D1, D2, Cin, Cout, mt, nt)
            for m, v in out.items():
                k = min(128, Cout - n0)
                Y[m, n0:n0 + k] = v[:k]
    assert not np.isnan(Y).any()
    return float(np.abs(Y - ref).max())

sp = [(0, dh, dw) for dh in (-1, 0, 1) for dw in (-1, 0, 1)]
if __name__ == "__main__":
    print(check(2, 8, 32, 128, 136, sp), check(1, 16, 16, 192, 64, sp), check(2, 16, 48, 64, 128, [(0, -1, 0), (0, 0, 0), (0, 1, 0)]), check(1, 8, 16, 64, 8, [(0, 0, 0)]))
