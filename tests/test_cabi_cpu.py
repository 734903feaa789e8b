"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every declared symbol."""
import os
import re

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from mm_diffusion import _hip
    if not os.path.exists(_hip.LIB_PATH):
        import importlib.util
        spec = importlib.util.spec_from_file_location("mmd_build", os.path.join(ROOT, "mm-diffusion_amd", "build.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        m.build(verbose=False)
    lib = _hip.lib()                       # binds argtypes for every symbol; AttributeError if one is missing
    assert lib.mmd_version() >= 100
    hdr = open(os.path.join(ROOT, "include", "mmd.h")).read()
    declared = set(re.findall(r"\b(mmd_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_hip.EXPORTS), declared ^ set(_hip.EXPORTS)
    for name in declared:
        assert hasattr(lib, name)


def test_argument_errors_are_reported_not_fatal():
    from mm_diffusion import _hip
    lib = _hip.lib()
    # bad dtype / null pointers must come back as an error code with a message (no GPU needed: checked before launch)
    rc = lib.mmd_conv_gemm(7, None, 0, None, None, None, 0, None, 0, 1, 8, 8, 1, None, 1, 1, 1, 0, None)
    assert rc < 0 and b"conv_gemm" in lib.mmd_last_error()
    rc = lib.mmd_attn_small_fwd(1, None, 0, None, 0, 64, 4, 1, 64, 1, 1, 1, 1, None)
    assert rc < 0


def test_host_only_size_queries():
    """Entry points that never touch the GPU: the workgroup count of one conv weight in the re-pack kernels (tiles of 32 output x
    (216 / taps) input channels, input tile a multiple of 16 where it can be; at most 27 taps), argument errors of the new round-6 entries."""
    from mm_diffusion import _hip
    lib = _hip.lib()
    ct = lambda nt: (216 // nt) & ~15 if 216 // nt >= 16 else 216 // nt       # noqa: E731
    for Cout, Cin, nt in ((128, 3, 27), (3, 128, 27), (512, 512, 1), (256, 384, 9), (40, 24, 3), (33, 17, 9), (1, 1, 1)):
        want = -(-Cout // 32) * -(-Cin // ct(nt))
        assert lib.mmd_pack_blocks(Cout, Cin, nt) == want, (Cout, Cin, nt)
    assert lib.mmd_pack_blocks(8, 8, 28) == -1 and lib.mmd_pack_blocks(0, 8, 1) == -1
    rc = lib.mmd_gn_group(1, None, 0, None, 0, 512, 2, 400, 1, 400, 400, 1, None, None, None, 0, 1e-5, 1, None, None, None, None)
    assert rc < 0 and b"gn_group" in lib.mmd_last_error()
    rc = lib.mmd_gn_bwd_ws0(1, None, 0, None, 0, None, 0, 0, 256, 1, 64, 1, 64, 64, 1, None, None, None, None, None, None, 0, 1, None, None, None, 0,
                            None, None)
    assert rc < 0 and b"gn_bwd" in lib.mmd_last_error()
