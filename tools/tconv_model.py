"""Lane-level model of mmd_tconv (mm-diffusion_amd/csrc/mmd_tconv.hip): the temporal k = 3 conv with the activations stationary in
registers.  A wave's 32 rows are 2 pixels x 16 frames (row r: pixel r // 16, frame r % 16), so the 16 frames of a pixel are the 16 lanes
of a DPP row; the operand of tap df is the wave's own x fragment moved by df lanes inside the row with zeros shifted in
(v_mov_b32_dpp row_shr:1 for df = -1, row_shl:1 for df = +1, bound_ctrl) - the conv's zero padding in time.  The model applies exactly
that to the fragment registers, runs the MFMAs (tools/tattn_model.py: mfma) in the kernel's K order (tap-major, then channel) and
compares with a float64 conv1d.  `check()` is run by tests/test_host_cpu.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tattn_model import L, mfma  # noqa: E402


def dpp_row_shift(frag, df):
    """frag [64 lanes][8] -> the registers after row_shr:1 (df = -1: lane i reads lane i - 1 of its 16-lane row) or row_shl:1 (df = +1),
    lanes that would read outside the row get 0."""
    out = np.zeros_like(frag)
    for l in range(L):
        src = (l % 16) + df
        if 0 <= src < 16:
            out[l] = frag[(l // 16) * 16 + src]
    return out


def check(seed=0, Cin=64, Cout=32, verbose=False):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((2, 16, Cin))                  # [pixel][frame][channel]
    W = rng.standard_normal((Cout, 3, Cin)) / np.sqrt(3 * Cin)       # [co][tap df = -1, 0, +1][ci] (the packed K = tap * Cin + ci)
    ref = np.zeros((2, 16, Cout))
    for f in range(16):
        for t, df in enumerate((-1, 0, 1)):
            if 0 <= f + df < 16:
                ref[:, f] += x[:, f + df] @ W[:, t].T
    # fragments: k-step cg, lane (row r = l % 32 = pixel * 16 + frame, half h): channels 16 cg + 8 h + e
    rows = x.reshape(32, Cin)
    xf = [np.array([[rows[l % 32, 16 * cg + 8 * (l // 32) + e] for e in range(8)] for l in range(L)]) for cg in range(Cin // 16)]
    out = np.zeros((32, Cout))
    for mt in range(Cout // 32):
        acc = np.zeros((L, 16))
        for t, df in enumerate((-1, 0, 1)):
            for cg in range(Cin // 16):
                wfrag = np.array([[W[32 * mt + l % 32, t, 16 * cg + 8 * (l // 32) + e] for e in range(8)] for l in range(L)])
                b = xf[cg] if df == 0 else dpp_row_shift(xf[cg], df)
                acc = mfma(wfrag, b, acc)
        for l in range(L):
            for i in range(16):
                out[l % 32, 32 * mt + 8 * (i // 4) + 4 * (l // 32) + i % 4] = acc[l, i]
    err = np.abs(out - ref.reshape(32, Cout)).max() / np.abs(ref).max()
    if verbose:
        print("max relative error of the register model against conv1d:", err)
    assert err < 1e-12, err
    return err


if __name__ == "__main__":
    check(verbose=True)
