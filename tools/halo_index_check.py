"""Index-math emulation of conv_gemm_halo_kernel (tile 130): DMA lane mapping + swizzle -> LDS image -> fragment reads."""
import numpy as np
rng = np.random.default_rng(0)
PH, PW, HWD = 8, 16, 18
HR, HG = 180, 23
def run(D0, D1, D2, Cin, taps, d0, h0, w0):
    EPV = 8
    A = rng.standard_normal((D0 * D1 * D2, Cin)).astype(np.float32)
    nchunk = Cin // 64
    ok_all = True
    for c in range(nchunk):
        lds = np.full((HG * 8, 8, 8), np.nan, np.float32)      # [row][phys chunk][8 elems]
        for wave in range(4):
            for j in range(6):
                g = wave + 4 * j
                if g >= HG: continue
                for lane in range(64):
                    lrow, pc = lane >> 3, lane & 7
                    r = 8 * g + lrow
                    hr, hc = divmod(r, HWD)
                    hh, ww = h0 - 1 + hr, w0 - 1 + hc
                    ok = r < HR and 0 <= hh < D1 and 0 <= ww < D2
                    logical = pc ^ ((r >> 1) & 7)
                    if ok:
                        m = (d0 * D1 + hh) * D2 + ww
                        val = A[m, c * 64 + logical * 8: c * 64 + logical * 8 + 8]
                    else:
                        val = np.zeros(8, np.float32)
                    # dest lane-linear: base g*1024 + lane*16 -> row 8g + lane//8, phys chunk lane%8
                    lds[8 * g + lane // 8, lane % 8] = val
        # fragment reads
        for (dh, dw) in taps:
            toff = dh * HWD + dw
            for wr in range(2):
                for b in range(2):
                    banks_seen = {}
                    for l31 in range(32):
                        px = wr * 64 + b * 32 + l31
                        rb = ((px >> 4) + 1) * HWD + (px & 15) + 1
                        r = rb + toff
                        key = (r >> 1) & 7
                        for half in range(2):
                            for cc in range(4):
                                logical = 2 * cc + half
                                phys = logical ^ key
                                got = lds[r, phys]
                                ph, pw = px >> 4, px & 15
                                hh, ww = h0 + ph + dh, w0 + pw + dw
                                if 0 <= hh < D1 and 0 <= ww < D2:
                                    exp = A[(d0 * D1 + hh) * D2 + ww, c * 64 + logical * 8: c * 64 + logical * 8 + 8]
                                else:
                                    exp = np.zeros(8, np.float32)
                                if not np.array_equal(got, exp):
                                    ok_all = False
                    # bank conflicts: 16-lane groups of one ds_read_b128 (fixed cc, half)
                    for cc in range(4):
                        for half in range(2):
                            for grp in range(2):
                                slots = set()
                                for l31 in range(grp * 16, grp * 16 + 16):
                                    px = wr * 64 + b * 32 + l31
                                    r = ((px >> 4) + 1) * HWD + (px & 15) + 1 + toff
                                    phys = (2 * cc + half) ^ ((r >> 1) & 7)
                                    slots.add(((r * 128 + phys * 16) // 16) % 16)     # 16-byte slot within the 256-byte bank row
                                if len(slots) != 16:
                                    print("bank conflict", dh, dw, wr, b, cc, half, grp, len(slots))
                                    ok_all = False
    return ok_all
taps = [(dh, dw) for dh in (-1, 0, 1) for dw in (-1, 0, 1)]
if __name__ == "__main__":
    print(run(2, 16, 32, 128, taps, 1, 0, 0), run(2, 16, 32, 128, taps, 0, 8, 16), run(1, 8, 16, 64, taps, 0, 0, 0), run(3, 24, 48, 64, taps, 2, 16, 32))
