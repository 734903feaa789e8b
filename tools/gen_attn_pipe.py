#!/usr/bin/env python3
"""Generator of the hand-written instruction stream of attn_pipe_kernel<64> (mm-diffusion_amd/csrc/mmd_attn_pipe_body.inc).

One pipelined iteration of the flash-attention loop at head width 64 works on TWO key tiles at once: the online softmax of tile t
(scores in sa0 / sa1, 32 per lane) is pure VALU, and the matrix work issued between its instructions is the S^T = K Q^T of tile
t + 1 (8 MFMAs, no dependence on this iteration's VALU) and the O^T += V^T P^T of tile t (8 MFMAs, each pair behind the exp / cvt of
its 8 P values).  hipcc places a block of 8 MFMAs, then ~190 VALU instructions, then 8 MFMAs (the wave stalls in-order for
256 cycles per block and the matrix pipe idles under the VALU block).  Here the whole iteration is ONE asm statement with literal
registers: the order below is the issue order, nothing is left to the compiler (a first version with one `asm volatile` per
instruction and compiler-allocated registers got an `s_nop 0` from hipcc between every dependent pair of statements - its hazard
recognizer assumes any asm result may be a dst-sel forward - 56 pads per iteration).

Interface (physical-register constraints in mmd_attn.hip; variant A = even tiles, B = odd tiles with the two score sets swapped):
  v[0:31]  scores set 0   v[32:63] scores set 1   v[64:95] O^T accumulators   v[96:111] Q fragments
  v112-115 K fragment addresses (per k step)   v116-117 V^T fragment addresses (per d tile)   v118 scale*log2e   v119 m   v120 l
  v121 ... temporaries (clobbers), allocated below by a linear scan over the final order.

The order comes from a list scheduler over the dependency graph of the iteration (longest path first) under the machine's rules:
  * v_exp_f32 result -> any use: 1 wait state (distance 2);  VALU write of an MFMA operand (v_cvt_pk, the O rescale) -> the MFMA:
    2 wait states (distance 3);  VALU write -> v_permlane32_swap: `s_nop 1` in front of the swap;
  * MFMAs in a fixed order, at least MFMA_GAP instructions apart (an MFMA holds the pipe 32 cycles = ~7 issue slots);
  * LDS reads in consumption order, each at least LDS_AHEAD instructions in front of the counted `s_waitcnt lgkmcnt(N)` that guards
    its MFMA (N from the final order: LDS operations return in order);
  * a register is not handed out again within REUSE_GAP instructions of its last read by an MFMA;
  * MFMA result -> VALU read never occurs inside an iteration: the scores of tile t + 1 are read in the next asm statement, the O
    accumulators are rescaled >= 40 instructions after the last P V MFMA of the previous one (the top of the loop lies between).

The arithmetic (instruction for instruction) is that of attn_mfma_kernel / attn_dma_kernel: same products, same summation order,
same rounding - the output is bitwise equal (tests/test_round6_gpu.py).

usage: python tools/gen_attn_pipe.py [--check] [--stats]     (--check: fail if the committed .inc differs)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "mm-diffusion_amd", "csrc", "mmd_attn_pipe_body.inc")

MFMA_GAP = 9        # instructions between consecutive MFMAs (minimum)
LDS_AHEAD = 10      # instructions between an LDS read and the wait of its consumer (minimum)
REUSE_GAP = 4       # instructions before a register read by an MFMA may be redefined
VALU_DIST = 2       # instructions between a VALU result and its first use: a dependent VALU instruction issued right behind its
                    # producer waits for it; measured (profiles/r06_attn_pipe_variants.txt): distances 1 .. 4 run the same 99 - 101 us - the
                    # stream is not bound by it; 2 keeps the temporaries at 25 registers
K_UPFRONT = 4       # K fragment reads issued at the top of the iteration (the rest follow the S^T MFMAs one by one)
ALIGN_PAD = 0       # s_nop 0 instructions behind the `.p2align 3` that opens the stream (4-byte phase of the instruction stream)
SCALE_BONUS = (30, 12)   # priority of the O rescale (d tile 0, 1) over the exp stream: the first P V MFMA waits for both
TEMP0 = 121

PINNED = {"qf0": 96, "qf1": 100, "qf2": 104, "qf3": 108, "ka0": 112, "ka1": 113, "ka2": 114, "ka3": 115, "va0": 116, "va1": 117,
          "sc": 118, "m_run": 119, "l_run": 120, "o0": 64, "o1": 80}
TILE_B = 8192


class Op:
    def __init__(self, name, text, kind):
        self.name, self.text, self.kind = name, text, kind
        self.preds, self.succs = [], []
        self.slot, self.height, self.bonus = None, 0, 0
        self.res_dist = 1


class Sched:
    """ops with symbolic registers: {name} scalar temp / pinned scalar, {name:N} tuple of N, {name.i} element i of a tuple."""

    def __init__(self):
        self.ops = []
        self.last_def = {}
        self.width = {}        # temp name -> registers

    def add(self, name, text, kind, defs=(), uses=(), extra=(), dist=None):
        op = Op(name, text, kind)
        if dist is None:
            dist = VALU_DIST if kind in ("valu", "trans") else 1
        op.res_dist = max(dist, VALU_DIST) if kind in ("valu", "trans") else dist     # distance this op's results need before a use
        for u in uses:
            if u in self.last_def:
                p = self.last_def[u]
                op.preds.append((p, p.res_dist))
        for e, d in extra:
            op.preds.append((e, d))
        for d in defs:
            self.last_def[d] = op
        self.ops.append(op)
        return op


def build():
    S = Sched()
    add = S.add
    # ---------------- row max: four chains of 8 scores, combined, pair exchange, new running max, rescale factor
    quads = [(0, 0), (0, 8), (1, 0), (1, 8)]
    for c, (kt, b) in enumerate(quads):
        add(f"mx{c}_0", f"v_max3_f32 {{mx{c}}}, {{sa{kt}.{b}}}, {{sa{kt}.{b + 1}}}, {{sa{kt}.{b + 2}}}", "valu", [f"mx{c}"])
    for i in range(2):
        for c, (kt, b) in enumerate(quads):
            add(f"mx{c}_{i + 1}", f"v_max3_f32 {{mx{c}}}, {{mx{c}}}, {{sa{kt}.{b + 3 + 2 * i}}}, {{sa{kt}.{b + 4 + 2 * i}}}", "valu", [f"mx{c}"], [f"mx{c}"])
    add("mz", "v_max_f32 {mz}, {sa0.15}, {sa1.15}", "valu", ["mz"])
    add("mxa", "v_max3_f32 {mxa}, {mx0}, {mx1}, {sa0.7}", "valu", ["mxa"], ["mx0", "mx1"])
    add("mxb", "v_max3_f32 {mxb}, {mx2}, {mx3}, {sa1.7}", "valu", ["mxb"], ["mx2", "mx3"])
    add("mxw", "v_max3_f32 {mxw}, {mxa}, {mxb}, {mz}", "valu", ["mxw"], ["mxa", "mxb", "mz"])
    add("mxs", "v_mov_b32 {mxs}, {mxw}", "valu", ["mxs"], ["mxw"])
    add("mswap", "s_nop 1\nv_permlane32_swap_b32 {mxw}, {mxs}", "valu", ["mxw", "mxs"], ["mxw", "mxs"])
    add("mxp", "v_max_f32 {mxp}, {mxw}, {mxs}", "valu", ["mxp"], ["mxw", "mxs"])
    add("mt", "v_mul_f32 {mt}, {sc}, {mxp}", "valu", ["mt"], ["mxp"])
    add("m_new", "v_max_f32 {m_new}, {m_run}, {mt}", "valu", ["m_new"], ["mt"])
    add("dlt", "v_sub_f32 {dlt}, {m_run}, {m_new}", "valu", ["dlt"], ["m_new"])
    add("alpha", "v_exp_f32 {alpha}, {dlt}", "trans", ["alpha"], ["dlt"], dist=2)
    for op in S.ops:                # the serial chain up to alpha first: everything else of the iteration hangs on m_new / alpha
        op.bonus = 1000
    # ---------------- O *= alpha (alpha == 1.0 exactly for a lane whose running max did not move: x * 1.0 == x bitwise, so the
    #                  unconditional form equals the `if (__any(m_new != m_run))` form of the other kernels)
    for dt in range(2):
        for r in range(16):
            add(f"sc{dt}_{r}", f"v_mul_f32 {{o{dt}.{r}}}, {{alpha}}, {{o{dt}.{r}}}", "valu", [f"o{dt}_{r}"], ["alpha"], dist=3).bonus = SCALE_BONUS[dt]
    # ---------------- p = exp2(s * sc - m_new), running sum in the order of the other kernels, P packed to bf16 pairs
    fma_ops = []
    for i in range(32):
        kt, r = divmod(i, 16)
        g = i // 8
        fma_ops.append(add(f"f{i}", f"v_fma_f32 {{f{i}}}, {{sa{kt}.{r}}}, {{sc}}, -{{m_new}}", "valu", [f"f{i}"], ["m_new"]))
        add(f"e{i}", f"v_exp_f32 {{e{i}}}, {{f{i}}}", "trans", [f"e{i}"], [f"f{i}"], dist=2)
        if i >= 1:
            prev = "e0" if i == 1 else f"ps{i - 1}"
            add(f"ps{i}", f"v_add_f32 {{ps{i}}}, {{e{i}}}, {{{prev}}}", "valu", [f"ps{i}"], [f"e{i}", prev])
        if i % 2 == 1:
            k = (i % 8) // 2
            add(f"p{i // 2}", f"v_cvt_pk_bf16_f32 {{pf{g}.{k}}}, {{e{i - 1}}}, {{e{i}}}", "valu", [f"p{i // 2}"], [f"e{i - 1}", f"e{i}"], dist=3)
    add("pss", "v_mov_b32 {pss}, {ps31}", "valu", ["pss"], ["ps31"])
    add("pswap", "s_nop 1\nv_permlane32_swap_b32 {ps31}, {pss}", "valu", ["ps31", "pss"], ["ps31", "pss"])
    add("psum", "v_add_f32 {psum}, {ps31}, {pss}", "valu", ["psum"], ["ps31", "pss"])
    add("l_upd", "v_fma_f32 {l_run}, {l_run}, {alpha}, {psum}", "valu", ["l_run_next"], ["alpha", "psum"])
    # m <- m_new once nobody reads the old m (dlt) and nobody reads m_new any more (the last fma): m_new's register is the temp
    add("m_upd", "v_mov_b32 {m_run}, {m_new}", "valu", ["m_run_next"], ["m_new", "dlt"], [(f, 1) for f in fma_ops])
    for g in range(4):
        S.width[f"pf{g}"] = 4

    # ---------------- LDS reads (issue order = consumption order), waits, MFMAs
    lds_prev = [None]

    def lds(name, text, defs, after=None):
        extra = []
        if lds_prev[0] is not None:
            extra.append((lds_prev[0], 1))
        if after is not None:
            extra.append((after, 1))
        op = add(name, text, "lds", defs, (), extra)
        lds_prev[0] = op
        return op

    def kread(n, after=None):
        st, kt = divmod(n, 2)
        S.width[f"kf{n}"] = 4
        return lds(f"k{n}", f"ds_read_b128 {{kf{n}:4}}, {{ka{st}}} offset:@K+{kt * 4096}", [f"kf{n}_raw"], after)

    def vread(g, dt, u, after=None):
        kt, st = divmod(g, 2)
        off = (32 * kt + 16 * st + 8 * u) * 128
        S.width[f"vf{g}{dt}"] = 4
        return lds(f"v{g}{dt}{u}", f"ds_read_b64_tr_b16 {{vf{g}{dt}.{2 * u}:2}}, {{va{dt}}} offset:@V+{off}", [f"vf{g}{dt}_raw{u}"], after)

    reads = {}
    for n in range(K_UPFRONT):
        reads[f"k{n}"] = kread(n)
        reads[f"k{n}"].bonus = 2000                      # the first K fragments go out before anything else
    mf_prev = None
    vq = [(g, dt) for g in range(4) for dt in range(2)]
    for n in range(8):                                   # S^T MFMAs of tile t + 1: (st, kt) = divmod(n, 2)
        st, kt = divmod(n, 2)
        rd = reads[f"k{n}"]
        add(f"wk{n}", f"@WAIT k{n}", "wait", [f"kf{n}"], (), [(rd, LDS_AHEAD)])
        c = f"{{sb{kt}:16}}" if st else "0"
        extra = [(mf_prev, MFMA_GAP)] if mf_prev else []
        m = add(f"S{n}", f"v_mfma_f32_32x32x16_bf16 {{sb{kt}:16}}, {{kf{n}:4}}, {{qf{st}:4}}, {c}", "mfma", [f"sb{kt}"],
                [f"kf{n}"] + ([f"sb{kt}"] if st else []), extra)
        mf_prev = m
        if n + K_UPFRONT < 8:                            # the other K fragments one by one behind the first S^T MFMAs (registers)
            reads[f"k{n + K_UPFRONT}"] = kread(n + K_UPFRONT, after=m)
        if n >= 2:                                       # V^T fragment pairs 0 .. 5 go out behind S2 .. S7
            g, dt = vq[n - 2]
            reads[f"v{g}{dt}0"] = vread(g, dt, 0, after=m)
            reads[f"v{g}{dt}1"] = vread(g, dt, 1)
    for n in range(8):                                   # P V MFMAs of tile t: group g (16 keys), d tile dt
        g, dt = divmod(n, 2)
        r0, r1 = reads[f"v{g}{dt}0"], reads[f"v{g}{dt}1"]
        add(f"wv{g}{dt}", f"@WAIT v{g}{dt}1", "wait", [f"vf{g}{dt}"], (), [(r0, LDS_AHEAD), (r1, LDS_AHEAD)])
        m = add(f"P{g}{dt}", f"v_mfma_f32_32x32x16_bf16 {{o{dt}:16}}, {{vf{g}{dt}:4}}, {{pf{g}:4}}, {{o{dt}:16}}", "mfma", [f"o{dt}"],
                [f"vf{g}{dt}"] + [f"p{4 * g + k}" for k in range(4)] + [f"o{dt}_{r}" for r in range(16)] + ([f"o{dt}"] if g else []),
                [(mf_prev, MFMA_GAP)])
        mf_prev = m
        if n < 2:                                        # the last two V^T fragment pairs behind the first two P V MFMAs
            g2, dt2 = vq[n + 6]
            reads[f"v{g2}{dt2}0"] = vread(g2, dt2, 0, after=m)
            reads[f"v{g2}{dt2}1"] = vread(g2, dt2, 1)
    return S


def schedule(S):
    ops = S.ops
    index = {op: i for i, op in enumerate(ops)}
    for op in ops:
        for p, d in op.preds:
            p.succs.append((op, d))
    done = set()

    def height(op):
        if op in done:
            return op.height
        op.height = max([d + height(s) for s, d in op.succs], default=0)
        done.add(op)
        return op.height

    sys.setrecursionlimit(10000)
    for op in ops:
        height(op)
    order, pending, slot, nops = [], list(ops), 0, 0
    while pending:
        best = None
        for op in pending:
            if all(p.slot is not None and slot - p.slot >= d for p, d in op.preds):
                key = (op.height + op.bonus, -index[op])
                if best is None or key > best[0]:
                    best = (key, op)
        if best is None:
            order.append(Op(f"nop{nops}", "s_nop 0", "nop"))
            nops += 1
        else:
            best[1].slot = slot
            order.append(best[1])
            pending.remove(best[1])
        slot += 1
    return order, nops


TOK = re.compile(r"\{([a-z_0-9]+)(?:\.(\d+))?(?::(\d+))?\}")
FIXED = ("sa0", "sa1", "sb0", "sb1")


def allocate(S, order):
    """linear scan: tuple temps get even-aligned contiguous blocks; returns {temp: base register}, highest register used"""
    first, last, last_mfma = {}, {}, {}
    for i, op in enumerate(order):
        for m in TOK.finditer(op.text):
            n = m.group(1)
            if n in PINNED or n in FIXED:
                continue
            first.setdefault(n, i)
            last[n] = i
            if op.kind == "mfma":
                last_mfma[n] = i
    free_at = {}                  # register -> instruction index from which it is free
    base = {}
    top = TEMP0
    for n in sorted(first, key=lambda n: first[n]):
        w = S.width.get(n, 1)
        al = 2 if w > 1 else 1
        r = TEMP0
        while not (r % al == 0 and all(free_at.get(r + k, -1) <= first[n] for k in range(w))):
            r += 1
        base[n] = r
        rel = last[n] + (REUSE_GAP if last_mfma.get(n) == last[n] else 1)
        for k in range(w):
            free_at[r + k] = rel
        top = max(top, r + w - 1)
    return base, top


def verify(S, order, base):
    """Replay the final order on register NAMES: every register an instruction reads must still hold the temporary it names (the
    allocator handed nothing out while it was live), the hazard distances are re-checked on final instruction indices, and no MFMA
    reads an LDS destination in front of the wait that covers it."""
    holds = {}                    # register -> (temp name, element)
    pending = []                  # LDS reads not yet covered by a wait: (op name, temp)
    written_at = {}               # temp -> (index, distance its value needs) of the last VALU writer
    for i, op in enumerate(order):
        if op.kind == "nop":
            continue
        if op.kind == "wait":
            names = [n for n, _ in pending]
            if op.text.split()[1] in names:          # (else: an earlier wait for a later read has covered it - LDS returns in order)
                pending = pending[names.index(op.text.split()[1]) + 1:]
            continue
        toks = [(m.group(1), int(m.group(2) or 0), int(m.group(3) or 1)) for m in TOK.finditer(op.text)]
        swap = "permlane" in op.text
        uses = toks if swap else toks[1:]
        defs = toks if swap else toks[:1]
        for n, el, w in uses:
            if n in PINNED or n in FIXED:
                continue
            for k in range(w):
                assert holds.get(base[n] + el + k) == (n, el + k), ("clobbered", i, op.name, n, base[n] + el + k, holds.get(base[n] + el + k))
            if n in written_at:
                wi, d = written_at[n]
                assert i - wi >= d, ("hazard distance", op.name, n, i - wi, d)
            assert not any(t == n for _, t in pending), ("read of an LDS destination in front of its wait", op.name, n)
        for n, el, w in defs:
            if n in PINNED or n in FIXED:
                continue
            for k in range(w):
                holds[base[n] + el + k] = (n, el + k)
            if op.kind == "lds":
                pending.append((op.name, n))
            else:
                written_at[n] = (i, op.res_dist)
    assert not pending
    return True


def render(S, order, base, variant):
    sa, sb = (0, 32) if variant == "A" else (32, 0)
    koff = TILE_B if variant == "A" else 0                       # K stage of tile t + 1
    voff = 2 * TILE_B + (0 if variant == "A" else TILE_B)        # V stage of tile t
    tup = {"sa0": sa, "sa1": sa + 16, "sb0": sb, "sb1": sb + 16}

    def reg(m):
        n, el, w = m.group(1), m.group(2), m.group(3)
        b = tup[n] if n in tup else PINNED[n] if n in PINNED else base[n]
        if el is not None:
            b += int(el)
        if w is not None and int(w) > 1:
            return f"v[{b}:{b + int(w) - 1}]"
        return f"v{b}"

    issued, lines = [], []
    for op in order:
        if op.kind == "lds":
            issued.append(op.name)
        if op.kind == "wait":
            cnt = len(issued) - 1 - issued.index(op.text.split()[1])
            assert 0 <= cnt <= 15
            lines.append(f"s_waitcnt lgkmcnt({cnt})")
            continue
        t = TOK.sub(reg, op.text)
        t = re.sub(r"offset:@K\+(\d+)", lambda m: f"offset:{koff + int(m.group(1))}", t)
        t = re.sub(r"offset:@V\+(\d+)", lambda m: f"offset:{voff + int(m.group(1))}", t)
        lines += t.split("\n")
    return lines


def ablate(lines, kind):
    """TIMING-ONLY variants of the stream (tools/attn_pipe_ablate.sh): the results are wrong, the durations say which resource binds."""
    if kind == "nomfma":
        return [ln for ln in lines if not ln.startswith("v_mfma")]
    if kind == "novalu":
        return [ln for ln in lines if ln.startswith(("v_mfma", "ds_read", "s_waitcnt"))]
    if kind == "nolds":
        return [ln for ln in lines if not ln.startswith(("ds_read", "s_waitcnt"))]
    if kind == "noexp":
        return [ln.replace("v_exp_f32", "v_mov_b32") for ln in lines]
    if kind == "mfmaonly":
        return [ln for ln in lines if ln.startswith("v_mfma")]
    raise SystemExit("unknown ablation " + kind)


def generate(abl=None):
    S = build()
    order, nops = schedule(S)
    assert sum(1 for o in order if o.kind == "mfma") == 16 and sum(1 for o in order if o.kind == "lds") == 24
    base, top = allocate(S, order)
    assert top <= 255, top
    verify(S, order, base)
    mf = [i for i, op in enumerate(order) if op.kind == "mfma"]
    n_valu = sum(1 for op in order if op.kind in ("valu", "trans"))
    out = [
        "// GENERATED by tools/gen_attn_pipe.py - do not edit; `python tools/gen_attn_pipe.py --check` is a CPU test.",
        f"// one pipelined iteration: {len(order)} instructions = 16 MFMAs + {n_valu} VALU + 24 LDS reads + 16 waits + {nops} pads;",
        f"// MFMAs at instructions {mf}; temporaries v{TEMP0} .. v{top}.",
    ]
    for variant in "AB":
        lines = render(S, order, base, variant)
        if abl:
            lines = ablate(lines, abl)
        lines = [".p2align 3"] + ["s_nop 0"] * ALIGN_PAD + lines
        out.append(f"#define ATTN_PIPE_ASM_{variant} \\")
        out += [f'  "{ln}\\n\\t" \\' for ln in lines[:-1]] + [f'  "{lines[-1]}"']
    out.append("#define ATTN_PIPE_CLOBBERS " + ", ".join(f'"v{r}"' for r in range(TEMP0, top + 1)))
    return "\n".join(out) + "\n", order, nops, mf, top


def main():
    abl = sys.argv[sys.argv.index("--ablate") + 1] if "--ablate" in sys.argv else None
    for kv in sys.argv[1:]:                      # schedule parameters for experiments: MFMA_GAP=12 LDS_AHEAD=20 ...
        if "=" in kv and kv.split("=")[0] in ("MFMA_GAP", "LDS_AHEAD", "REUSE_GAP", "VALU_DIST", "ALIGN_PAD", "K_UPFRONT"):
            globals()[kv.split("=")[0]] = int(kv.split("=")[1])
    text, order, nops, mf, top = generate(abl)
    if "--out" in sys.argv:
        with open(sys.argv[sys.argv.index("--out") + 1], "w") as f:
            f.write(text)
        print("wrote variant", abl, [b - a for a, b in zip(mf, mf[1:])])
        return
    if "--stats" in sys.argv:
        print(f"{len(order)} instructions, {nops} pads, top v{top}, MFMAs at {mf}, gaps {[b - a for a, b in zip(mf, mf[1:])]}")
        for i, op in enumerate(order):
            print(i, op.kind, op.name)
        return
    if "--check" in sys.argv:
        if open(OUT).read() != text:
            print("mmd_attn_pipe_body.inc is stale: run python tools/gen_attn_pipe.py")
            sys.exit(1)
        print("ok")
        return
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, ";", len(order), "instructions,", nops, "pads; top register v%d;" % top, "MFMA gaps", [b - a for a, b in zip(mf, mf[1:])])


if __name__ == "__main__":
    main()
