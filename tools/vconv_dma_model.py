"""The DMA queue of one wave of vconv2d1d_kernel (mmd_vconv.hip), replayed in issue order against the counted s_waitcnt vmcnt(N) at
the top of every step: vmcnt retires in order, so "at most N outstanding" = "everything but the newest N instructions has landed".
Checks, for every step, that what the step READS has landed for the issuing wave (the barrier behind the wait extends that to all
waves): its weights, the halo rows its taps touch, the slots / affine rows its norm pieces transform - and that nothing is
overwritten while a step may still read it.  Restates the kernel's schedule; run by tests/test_host_cpu.py."""


def simulate(nchunk, gn, npatch=2):
    q = []            # issued instructions, in order: (tag, payload)
    landed = set()

    def issue(tag):
        q.append(tag)

    def wait(n):
        for t in q[:len(q) - n] if n else q:
            landed.add(t)

    need_errors = []

    def need(step, tag):
        if tag not in landed:
            need_errors.append((step, tag))

    def first_halo(P):
        for j in range(5):
            issue(("H", P, 0, j))
        if gn:
            issue(("G", P, 0))
            if nchunk > 1:
                issue(("G", P, 1))

    # prologue of the first patch
    first_halo(0)
    for i in range(3):
        issue(("W", 0, 0, i))
    for i in range(3):
        issue(("W", 0, 1, i))
    for P in range(npatch):
        has_next = P + 1 < npatch
        wait(0)
        if gn:                                   # chunk 0 is normalised before the first step
            for j in range(5):
                need((P, "pro"), ("H", P, 0, j))
            need((P, "pro"), ("G", P, 0))
        for c in range(nchunk):
            more = c + 1 < nchunk
            for J in range(3):
                s = 3 * c + J
                n = (4 if gn else 3) if J == 0 else (3 if J == 1 else (5 if more else 2))
                wait(n)
                for i in range(3):
                    need((P, c, J), ("W", P, s, i))
                hh_needed = {0: (0, 1, 2), 1: (0, 1, 2, 3, 4), 2: (0, 1, 2, 3, 4)}[J]   # pieces: 0..2 = rows hh 0..3, 3..4 = rows hh 4..5
                for j in hh_needed:
                    need((P, c, J), ("H", P, c, j))
                if gn:
                    if J == 0 and c > 0:
                        need((P, c, J), ("H", P, c, 3)); need((P, c, J), ("H", P, c, 4)); need((P, c, J), ("G", P, c))
                    if J == 1 and more:
                        for j in range(3):
                            need((P, c, J), ("H", P, c + 1, j))
                        need((P, c, J), ("G", P, c + 1))
                    if J == 2 and more:
                        need((P, c, J), ("G", P, c + 1))
                for k in range(6):
                    if J == 0 and more:
                        issue(("H", P, c + 1, k)) if k < 3 else issue(("W", P, s + 2, k - 3))
                    elif J == 0:
                        if k < 3:
                            issue(("W", P, s + 2, k))
                    elif J == 1 and more:
                        if k < 3:
                            issue(("W", P, s + 2, k))
                        elif k < 5:
                            issue(("H", P, c + 1, k))
                    elif J == 1:
                        if k < 2:
                            issue(("T", P, 0, k))
                    elif more:
                        if k < 3:
                            issue(("W", P, s + 2, k))
                        elif gn and k == 3:
                            issue(("G", P, c + 2) if c + 2 < nchunk else ("dummy", P, s))
                    else:
                        if k < 2:
                            issue(("T", P, 1, k))
        for S2 in range(6):
            wait(2 if S2 < 5 else 3)
            for i in range(2):
                need((P, "t", S2), ("T", P, S2, i))
            for k in range(3):
                if S2 + 2 < 6:
                    if k < 2:
                        issue(("T", P, S2 + 2, k))
                else:                            # steps 4 / 5: the next patch's spatial slabs 0 / 1 (or dummies)
                    issue(("W", P + 1, S2 - 4, k) if has_next else ("dummy", P, S2, k))
        if has_next:
            first_halo(P + 1)
    return need_errors, len(q)


def check():
    for nchunk in (1, 2, 3, 4, 8, 12):
        for gn in (False, True):
            errs, n = simulate(nchunk, gn)
            assert not errs, (nchunk, gn, errs[:5])
    return True


if __name__ == "__main__":
    check()
    print("vconv DMA schedule ok")
