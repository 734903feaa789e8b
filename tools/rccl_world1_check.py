#!/usr/bin/env python3
"""RCCL beside libmmd on ONE MI355X: a single-rank `nccl` process group (WORLD_SIZE=1; RCCL refuses two ranks on one device, so this is the
most a 1-GPU box can run of the real backend) around the exact collective sequence of `bench.py`'s sampling leg - all-reduce of ones,
barrier fences around the timed graph replays, MAX all-reduce of the fp64 elapsed time, the terminal all-gather of both sample tensors -
and a check that the replayed trajectory is bitwise the one the same stepper produces with no process group initialised, whatever the
engine's buffers held before (NaN- / zero-filled allocator blocks) and whether or not the plan was built + autotuned after seeding.
usage (GPU box): python tools/rccl_world1_check.py [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402
import bench                                    # noqa: E402  (puts mm-diffusion_amd on sys.path)


def trajectory(steps, with_group, prefill=None):
    """One seeded run of `steps` + 1 graph-replayed p_sample steps at batch 4.  prefill (a byte value): park a 24 GB block filled with
    that byte in torch's caching allocator first, so every large buffer of the new engine starts from that content (0xFF = bf16 / fp32
    NaNs, 0x00 = what a fresh process gets) - a read of memory nobody wrote would show as a difference between the two."""
    import gc
    import random
    from mm_diffusion.sampler import GraphStepper
    device = torch.device("cuda", 0)
    gc.collect()
    if prefill is not None:
        torch.cuda.empty_cache()
        blk = torch.empty(24 << 30, dtype=torch.uint8, device=device)
        blk.fill_(prefill)
        torch.cuda.synchronize()
        del blk
    fl, model, diff = bench.build("bf16", "250", 4, device)
    random.seed(1234)
    torch.manual_seed(1234)
    st = GraphStepper(diff, model, 4, device, clip_denoised=True)
    st.load(torch.randn(4, *fl["video_size"]).to(device), torch.randn(4, *fl["audio_size"]).to(device))
    T, rep = diff.num_timesteps, {}

    def fence():
        torch.cuda.synchronize()
        if with_group:
            dist.barrier()
        torch.cuda.synchronize()

    st.step(T - 1)
    fence()
    first = {k: v.clone() for k, v in st.current().items()}
    t0 = time.perf_counter()
    for i in range(steps):
        st.step(T - 2 - i)
    fence()
    el = time.perf_counter() - t0
    cur = {k: v.clone() for k, v in st.current().items()}
    if with_group:
        tt = torch.tensor([el], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        rep["max_elapsed_matches"] = float(tt.item()) == el
        for k, v in cur.items():
            out = [torch.empty_like(v)]
            dist.all_gather(out, v.contiguous())
            rep[f"all_gather_{k}_bitwise"] = bool(torch.equal(out[0], v))
        fence()
    rep["ms_per_step"] = 1e3 * el / steps
    rep["finite"] = all(bool(torch.isfinite(v).all()) for v in cur.values())
    return {"first": first, "last": cur}, rep


def diff_of(a, b):
    out = {}
    for when in ("first", "last"):
        for k in a[when]:
            x, y = a[when][k].double(), b[when][k].double()
            out[f"{when}_{k}"] = {"bitwise": bool(torch.equal(a[when][k], b[when][k])), "n_diff": int((x != y).sum().item()),
                                  "rel_l2": float(((x - y).norm() / y.norm().clamp_min(1e-30)).item())}
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    A, rA = trajectory(steps, False)                    # fresh process
    B, rB = trajectory(steps, False, prefill=0xFF)      # every large buffer starts as NaNs
    C, rC = trajectory(steps, False, prefill=0x00)      # ... as zeros
    os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    from mm_diffusion import dist_util
    dist_util.setup_dist(backend="nccl")
    ones = torch.ones(1, device="cuda")
    dist.all_reduce(ones)
    D, rD = trajectory(steps, True, prefill=0x00)       # RCCL process group up, collectives of bench.py's sampling leg around the loop
    rep = {"backend": dist.get_backend(), "world": dist.get_world_size(), "ranks_seen": int(ones.item()), "steps": steps,
           "fresh_vs_zero_prefill": diff_of(A, C), "nan_prefill_vs_zero_prefill": diff_of(B, C), "rccl_group_vs_none_zero_prefill": diff_of(D, C),
           "runs": {"fresh": rA, "nan_prefill": rB, "zero_prefill": rC, "rccl_group": rD}}
    dist.destroy_process_group()
    print(json.dumps(rep))
    ok = rD["finite"] and rD["max_elapsed_matches"] and rep["ranks_seen"] == 1 and all(v for k, v in rD.items() if k.startswith("all_gather_")) \
        and all(v["bitwise"] for v in rep["rccl_group_vs_none_zero_prefill"].values()) \
        and all(v["bitwise"] for v in rep["nan_prefill_vs_zero_prefill"].values()) \
        and all(v["bitwise"] for v in rep["fresh_vs_zero_prefill"].values())      # the first run builds + autotunes the plan after seeding
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
