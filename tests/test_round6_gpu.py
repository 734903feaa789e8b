"""Round-6 kernels (through the C ABI).

attn_pipe_kernel<64> (mmd_attn_fwd impl 5): the software-pipelined flash attention at head width 64 whose loop iteration is one
hand-written asm statement (tools/gen_attn_pipe.py).  Same arithmetic in the same order as attn_mfma_kernel (impl 2, the kernel
tests/test_ops_gpu.py pins against the oracle's attention, unet:221-240, 507-564): the outputs must be BITWISE equal - on circular
windows with a shift, key counts that are not multiples of 64 (ragged last tile, zero-filled DMA rows), 1 .. 17 key tiles (prologue /
odd and even pipelined iterations / drain), ragged query tiles, waves without queries, the last group's remainder, and on inputs that
force the online-softmax rescale."""
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from mm_diffusion import ops as o
    return o


CASES = [
    ("spatial 1024", 2, 4, 4 * 1024, 1024, 4 * 1024, 1024, 1, 4), ("v<-a", 2, 16, 16 * 256, 256, 1600, 100, 1, 4),
    ("a<-v window 4", 1, 16, 1600, 100, 16 * 256, 256, 4, 6), ("ragged keys / queries", 2, 8, 8 * 77, 77, 8 * 50, 50, 3, 2),
    ("last group takes the remainder", 1, 16, 1610, 100, 16 * 64, 64, 8, 2), ("one short tile", 1, 16, 16 * 64, 64, 16 * 25, 25, 1, 8),
    ("one full tile", 1, 4, 4 * 200, 200, 4 * 64, 64, 1, 2), ("two tiles", 1, 4, 4 * 130, 130, 4 * 128, 128, 1, 2),
    ("three tiles, ragged", 1, 4, 4 * 130, 130, 4 * 150, 150, 1, 2), ("four tiles", 2, 2, 2 * 300, 300, 2 * 256, 256, 1, 1),
    ("v<-a ds2 full size", 1, 16, 16 * 1024, 1024, 6400, 400, 1, 4), ("a<-v ds2 full size (16-query tail block)", 1, 16, 6400, 400, 16 * 1024, 1024, 1, 4),
    ("600 queries, 7 tiles", 2, 4, 4 * 600, 600, 4 * 440, 440, 1, 2), ("17 tiles", 1, 2, 2 * 160, 160, 2 * 1050, 1050, 1, 2),
    ("window wraps around the end of the key rows", 1, 8, 8 * 128, 128, 8 * 96, 96, 5, 2)]


@pytest.mark.parametrize("name,N,F,qr,qg,kr,kg,win,heads", CASES)
def test_attn_pipe_kernel_is_bitwise_the_mfma_kernel(ops, name, N, F, qr, qg, kr, kg, win, heads):
    ch = 64
    C = heads * ch
    g = torch.Generator(device="cuda").manual_seed(qr + kr)
    q = torch.randn(N * qr, 3 * C, device="cuda", generator=g).to(BF)
    kv = torch.randn(N * kr, 3 * C, device="cuda", generator=g).to(BF)
    for shift in (0, 5):
        sh = torch.tensor([shift], dtype=torch.int32, device="cuda")
        o2 = torch.zeros(N * qr, C, device="cuda", dtype=BF)
        o5 = torch.full((N * qr, C), 7.0, device="cuda", dtype=BF)
        ops.attn(q, kv, o2, heads, ch, N, F, qr, qg, kr, kg, win, shift_dev=sh, impl=2)
        ops.attn(q, kv, o5, heads, ch, N, F, qr, qg, kr, kg, win, shift_dev=sh, impl=5)
        torch.cuda.synchronize()
        assert torch.equal(o2, o5), f"{name} shift {shift}: rel-L2 {rel_l2(o5.float().cpu(), o2.float().cpu().numpy()):.3e}"


@pytest.mark.parametrize("spike_key,scale", [(250, 40.0), (70, 40.0), (3, 40.0), (250, -40.0), (135, 25.0)])
def test_attn_pipe_kernel_softmax_spike(ops, spike_key, scale):
    """The running maximum jumps in the first tile, in a pipelined iteration, in the last (ragged) tile; a key far below everything."""
    T, heads, ch = 300, 1, 64
    g = torch.Generator().manual_seed(28)
    qkv = (torch.randn(T, 3 * 64, generator=g) * 0.3).to(BF).float()
    qkv[spike_key, 64:128] = qkv[7, :64] * scale           # key aligned (or anti-aligned) with query 7
    qkv = qkv.to(BF).cuda()
    outs = {}
    for impl in (2, 5):
        out = torch.zeros(T, 64, dtype=BF, device="cuda")
        ops.attn(qkv, qkv, out, heads, ch, 1, 1, T, T, T, T, 1, impl=impl)
        outs[impl] = out
    torch.cuda.synchronize()
    assert torch.isfinite(outs[5].float()).all()
    assert torch.equal(outs[2], outs[5])


def test_attn_pipe_kernel_scores_far_below_zero(ops):
    """Every score around -60 (log2 domain): the first tile's rescale from the -1e30 start must not underflow the sums."""
    T, heads, ch = 200, 1, 64
    g = torch.Generator().manual_seed(5)
    q = torch.full((T, 64), 1.0) + torch.randn(T, 64, generator=g) * 0.05
    k = -q * 5.0 + torch.randn(T, 64, generator=g) * 0.05
    v = torch.randn(T, 64, generator=g)
    qkv = torch.cat([q, k, v], dim=1).to(BF).cuda()
    outs = {}
    for impl in (2, 5):
        out = torch.zeros(T, 64, dtype=BF, device="cuda")
        ops.attn(qkv, qkv, out, heads, ch, 1, 1, T, T, T, T, 1, impl=impl)
        outs[impl] = out
    torch.cuda.synchronize()
    assert torch.isfinite(outs[5].float()).all()
    assert torch.equal(outs[2], outs[5])


def test_attn_pipe_kernel_repeatable_under_load(ops):
    """200 launches of the ds2 cross-attention shape on two streams at once (the kernel shares its SIMDs with another instance of
    itself in a different phase): every output bitwise the first - the hand-counted waits hold under contention."""
    N, F, qr, qg, kr, kg, win, heads, ch = 2, 16, 16 * 1024, 1024, 6400, 400, 1, 4, 64
    C = heads * ch
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn(N * qr, 3 * C, device="cuda", generator=g).to(BF)
    kv = torch.randn(N * kr, 3 * C, device="cuda", generator=g).to(BF)
    ref = torch.zeros(N * qr, C, device="cuda", dtype=BF)
    ops.attn(q, kv, ref, heads, ch, N, F, qr, qg, kr, kg, win, impl=2)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = [torch.zeros_like(ref) for _ in range(4)]
    bad = 0
    for rep in range(50):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            if i % 2:
                with torch.cuda.stream(side):
                    ops.attn(q, kv, o, heads, ch, N, F, qr, qg, kr, kg, win, impl=5)
            else:
                ops.attn(q, kv, o, heads, ch, N, F, qr, qg, kr, kg, win, impl=5)
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    assert bad == 0, f"{bad} of 200 launches differ"
