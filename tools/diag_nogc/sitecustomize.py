# diagnostic only (tools/gpu_diag.sh): run a process with the cyclic GC off to test the "GC-driven __del__ during capture" hypothesis
import os
if os.environ.get("MMD_DIAG_NOGC") == "1":
    import gc
    gc.disable()
