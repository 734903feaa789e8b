#!/usr/bin/env python3
"""Lane-level numpy model of the row-major LDS tile + transposing reads that the DMA attention kernel (V operand) and the 9-tap weight
gradient kernel (both operands) share (mmd_attn.hip: read_v, mmd_bwd.hip: wgrad_tr_bf16_kernel::frag).

Tile: one plane = [64 rows][128 B] (64 bf16 channels per row), the 16-byte chunk c of row r stored at chunk c ^ (((r >> 1) & 1) << 2).
The DMA writes lane-linear: lane L of wave-instruction g lands at byte g * 1024 + L * 16, i.e. row 8 g + L / 8, physical chunk L % 8, and
FETCHES the logical chunk (L % 8) ^ swz(row) - the swizzle sits on the source address.

ds_read_b64_tr_b16 as the kernels rely on it (verified on the MI355X by the parity tests of both kernels): within a group of 16 lanes,
lane i supplies the address of 8 bytes (4 channels) at (row base + i / 4, channels c0 + 4 (i % 4)); the 16 lanes together cover a
4-row x 16-channel block and lane i RECEIVES the four rows of channel c0 + i.  A fragment of the 32x32x16 MFMA (8 k-slots per lane) is
two such reads (rows +0 and +8)."""
import numpy as np

ROWS, CH, ROWB = 64, 64, 128


def swz(row):
    return ((row >> 1) & 1) << 2


def stage_plane(src):
    """src [64 rows][64 channels] (any dtype of 2 bytes per element, here int16 codes) -> the LDS image as the DMA leaves it."""
    lds = np.zeros(ROWS * ROWB // 2, dtype=src.dtype)              # in 2-byte elements
    for g in range(8):                                             # wave-instructions of one plane
        for lane in range(64):
            row, pc = 8 * g + lane // 8, lane % 8
            lc = pc ^ swz(row)                                     # logical chunk this lane fetches
            dst = (g * 1024 + lane * 16) // 2
            lds[dst:dst + 8] = src[row, lc * 8:lc * 8 + 8]
    return lds


def tr_read(lds, addr_bytes):
    """One ds_read_b64_tr_b16 of a wave: addr_bytes[64] -> out[64][4]."""
    out = np.zeros((64, 4), dtype=lds.dtype)
    for grp in range(4):
        blk = np.stack([lds[addr_bytes[16 * grp + i] // 2: addr_bytes[16 * grp + i] // 2 + 4] for i in range(16)])   # [16 lanes][4 ch]
        blk = blk.reshape(4, 16)                                   # lanes 4 r .. 4 r + 3 = the 16 channels of row r of the block
        for i in range(16):
            out[16 * grp + i] = blk[:, i]                          # lane i: the four rows of channel i
    return out


def fragment(lds, rbase, ct):
    """The kernels' address arithmetic for the fragment of rows rbase .. rbase + 15, channel tile ct (32 channels): [64 lanes][8 k-slots]."""
    lo_addr, hi_addr = np.zeros(64, dtype=np.int64), np.zeros(64, dtype=np.int64)
    for lane in range(64):
        half = lane >> 5
        vrow0 = 4 * half + ((lane & 15) >> 2)
        vcolb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2
        for dst, extra in ((lo_addr, 0), (hi_addr, 8)):
            row = rbase + extra + vrow0
            ch = ((vcolb >> 4) + 4 * ct) ^ swz(row)
            dst[lane] = row * ROWB + ch * 16 + (vcolb & 15)
    return np.concatenate([tr_read(lds, lo_addr), tr_read(lds, hi_addr)], axis=1)


def check():
    rng = np.random.default_rng(0)
    src = rng.integers(-30000, 30000, size=(ROWS, CH)).astype(np.int16)
    lds = stage_plane(src)
    for rbase in (0, 16, 32, 48):
        for ct in (0, 1):
            fr = fragment(lds, rbase, ct)
            for lane in range(64):
                half, l31 = lane >> 5, lane & 31
                rows = [rbase + 4 * half + j for j in range(4)] + [rbase + 8 + 4 * half + j for j in range(4)]
                want = src[rows, 32 * ct + l31]
                assert np.array_equal(fr[lane], want), (rbase, ct, lane)
    # the 16 rows of a k-step are covered exactly once by the two half-waves' 8 k-slots: the reduction over k sees every row once
    seen = sorted(r for half in (0, 1) for r in [4 * half + j for j in range(4)] + [8 + 4 * half + j for j in range(4)])
    assert seen == list(range(16))
    return True


if __name__ == "__main__":
    print("ok" if check() else "mismatch")
