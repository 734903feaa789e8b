"""ORACLE (test infrastructure - NOT part of the product path).

CPU restatement of the reference's multimodal DPM-Solver / DPM-Solver++ driver
(/root/reference/mm_diffusion/multimodal_dpm_solver_plus.py): discrete VP noise schedule with piecewise-linear
log alpha (dpm:11-181,1306-1347), the continuous-time model wrapper (dpm:285-332), dynamic thresholding (dpm:419-440),
first / single-step second / single-step third / multistep second updates (dpm:532-968), the adaptive embedded pair
(dpm:1088-1149) and sample() (dpm:1151-1300) - on {"video", "audio"} dicts with [B]-shaped coefficient tensors, plain
torch fp32 ops.  Includes the reference's quirk of moving the AUDIO stream with x0-form coefficients in the
noise-prediction first-order update (dpm:576-584).

Parity status: PINNED against fixtures captured from the imported reference (tests/golden/tiny_dpm*.npz).
"""
import torch


def interp(x, xp, yp):
    """y(x) through keypoints (xp, yp) [K], linear extrapolation outside; x [N]."""
    K = xp.shape[0]
    i = (torch.searchsorted(xp, x.contiguous(), right=False) - 1).clamp(0, K - 2)
    return yp[i] + (x - xp[i]) * (yp[i + 1] - yp[i]) / (xp[i + 1] - xp[i])


class Schedule:
    def __init__(self, alphas_cumprod):
        self.log_alpha = 0.5 * torch.log(alphas_cumprod.float())
        self.N = len(self.log_alpha)
        self.t = torch.linspace(0., 1., self.N + 1)[1:]

    def log_mean(self, t):
        return interp(t.reshape(-1), self.t, self.log_alpha)

    def alpha(self, t):
        return torch.exp(self.log_mean(t))

    def std(self, t):
        return torch.sqrt(1. - torch.exp(2. * self.log_mean(t)))

    def lam(self, t):
        lm = self.log_mean(t)
        return lm - 0.5 * torch.log(1. - torch.exp(2. * lm))

    def inv_lam(self, lamb):
        la = -0.5 * torch.logaddexp(torch.zeros(1), -2. * lamb.reshape(-1))
        return interp(la, torch.flip(self.log_alpha, [0]), torch.flip(self.t, [0]))


def _e(v, x):
    return v.reshape(-1, *([1] * (x.dim() - 1)))


class Solver:
    def __init__(self, model, alphas_cumprod, predict_x0=False, thresholding=False, max_val=1., single=False):
        """single=True: tensor-valued solver of the SR stage (dpm_solver_plus.py): state {"x": tensor}, model(x, t) -> tensor."""
        self.model, self.ns = model, Schedule(alphas_cumprod)
        self.predict_x0, self.thresholding, self.max_val, self.single = predict_x0, thresholding, max_val, single

    def noise(self, x, t):
        B = next(iter(x.values())).shape[0]
        t = t.reshape(-1)
        t = t.expand(B) if t.shape[0] == 1 else t
        ti = ((t - 1. / self.ns.N) * self.ns.N).to(torch.int)
        if self.single:
            return {"x": self.model(x["x"], ti)[:, :3].float()}
        v, a = self.model(x["video"], x["audio"], ti)
        return {"video": v[:, :, :3].float(), "audio": a[:, :1].float()}

    def fn(self, x, t):
        n = self.noise(x, t)
        if not self.predict_x0:
            return n
        al, sg = self.ns.alpha(t), self.ns.std(t)
        out = {}
        for k in x:
            x0 = (x[k] - _e(sg, x[k]) * n[k]) / _e(al, x[k])
            if self.thresholding:
                s = torch.quantile(x0.abs().reshape(x0.shape[0], -1), 0.995, dim=1)
                s = _e(torch.maximum(s, torch.ones_like(s)), x0)
                x0 = torch.clamp(x0, -s, s) / (s / self.max_val)
            out[k] = x0
        return out

    def _coef(self, s, t):
        ns = self.ns
        return ns.lam(t) - ns.lam(s), ns.log_mean(s), ns.log_mean(t), ns.std(s), ns.std(t)

    def first(self, x, s, t, ms=None, inter=False):
        h, las, lat, sgs, sgt = self._coef(s, t)
        at = torch.exp(lat)
        ms = self.fn(x, s) if ms is None else ms
        if self.predict_x0:
            p1 = torch.expm1(-h)
            xt = {k: _e(sgt / sgs, x[k]) * x[k] - _e(at * p1, x[k]) * ms[k] for k in x}
        else:
            p1 = torch.expm1(h)
            xt = {k: _e(torch.exp(lat - las), x[k]) * x[k] - _e(sgt * p1, x[k]) * ms[k] for k in x}
            if "audio" in x:                                                                                # reference quirk
                xt["audio"] = _e(sgt / sgs, x["audio"]) * x["audio"] - _e(at * p1, x["audio"]) * ms["audio"]
        return (xt, {"ms": ms}) if inter else xt

    def second(self, x, s, t, r1=0.5, ms=None, inter=False):
        ns = self.ns
        r1 = 0.5 if r1 is None else r1
        h, las, lat, sgs, sgt = self._coef(s, t)
        s1 = ns.inv_lam(ns.lam(s) + r1 * h)
        la1, sg1 = ns.log_mean(s1), ns.std(s1)
        a1, at = torch.exp(la1), torch.exp(lat)
        ms = self.fn(x, s) if ms is None else ms
        if self.predict_x0:
            p11, p1 = torch.expm1(-r1 * h), torch.expm1(-h)
            x1 = {k: _e(sg1 / sgs, x[k]) * x[k] - _e(a1 * p11, x[k]) * ms[k] for k in x}
            m1 = self.fn(x1, s1)
            xt = {k: _e(sgt / sgs, x[k]) * x[k] - _e(at * p1, x[k]) * ms[k] - _e((0.5 / r1) * at * p1, x[k]) * (m1[k] - ms[k]) for k in x}
        else:
            p11, p1 = torch.expm1(r1 * h), torch.expm1(h)
            x1 = {k: _e(torch.exp(la1 - las), x[k]) * x[k] - _e(sg1 * p11, x[k]) * ms[k] for k in x}
            m1 = self.fn(x1, s1)
            xt = {k: _e(torch.exp(lat - las), x[k]) * x[k] - _e(sgt * p1, x[k]) * ms[k] - _e((0.5 / r1) * sgt * p1, x[k]) * (m1[k] - ms[k])
                  for k in x}
        return (xt, {"ms": ms, "m1": m1}) if inter else xt

    def third(self, x, s, t, r1=1. / 3., r2=2. / 3., ms=None, m1=None):
        ns = self.ns
        r1 = 1. / 3. if r1 is None else r1
        r2 = 2. / 3. if r2 is None else r2
        h, las, lat, sgs, sgt = self._coef(s, t)
        s1, s2 = ns.inv_lam(ns.lam(s) + r1 * h), ns.inv_lam(ns.lam(s) + r2 * h)
        la1, la2, sg1, sg2 = ns.log_mean(s1), ns.log_mean(s2), ns.std(s1), ns.std(s2)
        a1, a2, at = torch.exp(la1), torch.exp(la2), torch.exp(lat)
        ms = self.fn(x, s) if ms is None else ms
        if self.predict_x0:
            p11, p12, p1 = torch.expm1(-r1 * h), torch.expm1(-r2 * h), torch.expm1(-h)
            p22, p2 = torch.expm1(-r2 * h) / (r2 * h) + 1., p1 / h + 1.
            if m1 is None:
                m1 = self.fn({k: _e(sg1 / sgs, x[k]) * x[k] - _e(a1 * p11, x[k]) * ms[k] for k in x}, s1)
            x2 = {k: _e(sg2 / sgs, x[k]) * x[k] - _e(a2 * p12, x[k]) * ms[k] + _e(r2 / r1 * a2 * p22, x[k]) * (m1[k] - ms[k]) for k in x}
            m2 = self.fn(x2, s2)
            return {k: _e(sgt / sgs, x[k]) * x[k] - _e(at * p1, x[k]) * ms[k] + _e((1. / r2) * at * p2, x[k]) * (m2[k] - ms[k]) for k in x}
        p11, p12, p1 = torch.expm1(r1 * h), torch.expm1(r2 * h), torch.expm1(h)
        p22, p2 = torch.expm1(r2 * h) / (r2 * h) - 1., p1 / h - 1.
        if m1 is None:
            m1 = self.fn({k: _e(torch.exp(la1 - las), x[k]) * x[k] - _e(sg1 * p11, x[k]) * ms[k] for k in x}, s1)
        x2 = {k: _e(torch.exp(la2 - las), x[k]) * x[k] - _e(sg2 * p12, x[k]) * ms[k] - _e(r2 / r1 * sg2 * p22, x[k]) * (m1[k] - ms[k]) for k in x}
        m2 = self.fn(x2, s2)
        return {k: _e(torch.exp(lat - las), x[k]) * x[k] - _e(sgt * p1, x[k]) * ms[k] - _e((1. / r2) * sgt * p2, x[k]) * (m2[k] - ms[k]) for k in x}

    def multi2(self, x, mlist, tlist, t):
        ns = self.ns
        (m1, m0), (t1, t0) = mlist, tlist
        h0, h = ns.lam(t0) - ns.lam(t1), ns.lam(t) - ns.lam(t0)
        la0, lat, sg0, sgt = ns.log_mean(t0), ns.log_mean(t), ns.std(t0), ns.std(t)
        at, r0 = torch.exp(lat), h0 / h
        D = {k: _e(1. / r0, x[k]) * (m0[k] - m1[k]) for k in x}
        if self.predict_x0:
            c = at * (torch.exp(-h) - 1.)
            return {k: _e(sgt / sg0, x[k]) * x[k] - _e(c, x[k]) * m0[k] - 0.5 * _e(c, x[k]) * D[k] for k in x}
        c = sgt * (torch.exp(h) - 1.)
        return {k: _e(torch.exp(lat - la0), x[k]) * x[k] - _e(c, x[k]) * m0[k] - 0.5 * _e(c, x[k]) * D[k] for k in x}

    def adaptive(self, x, order, t_T, t_0, h_init=0.05, atol=0.0078, rtol=0.05, theta=0.9, t_err=1e-5):
        ns = self.ns
        B = next(iter(x.values())).shape[0]
        s = t_T * torch.ones(B)
        lam_s, lam_0 = ns.lam(s), ns.lam(t_0 * torch.ones(B))
        h = h_init * torch.ones(B)
        xp = x
        while torch.abs(s - t_0).mean() > t_err:
            t = ns.inv_lam(lam_s + h)
            if order == 2:
                lo, kw = self.first(x, s, t, inter=True)
                hi = self.second(x, s, t, r1=0.5, ms=kw["ms"])
            else:
                lo, kw = self.second(x, s, t, r1=1. / 3., inter=True)
                hi = self.third(x, s, t, ms=kw["ms"], m1=kw["m1"])
            nf = lambda v: torch.sqrt(torch.square(v.reshape(v.shape[0], -1)).mean(dim=-1, keepdim=True))   # noqa: E731
            E = torch.cat([nf((hi[k] - lo[k]) / torch.max(torch.ones_like(x[k]) * atol, rtol * torch.max(lo[k].abs(), xp[k].abs())))
                           for k in x]).max()
            if torch.all(E <= 1.):
                x, s, xp = hi, t, lo
                lam_s = ns.lam(s)
            h = torch.min(theta * h * torch.float_power(E, -1. / order).float(), lam_0 - lam_s)
        return x

    def steps(self, skip, t_T, t_0, N):
        if skip == "logSNR":
            lT, l0 = self.ns.lam(torch.tensor(t_T)), self.ns.lam(torch.tensor(t_0))
            return self.ns.inv_lam(torch.linspace(lT.item(), l0.item(), N + 1))
        if skip == "time_uniform":
            return torch.linspace(t_T, t_0, N + 1)
        return torch.linspace(t_T ** 0.5, t_0 ** 0.5, N + 1).pow(2)

    @torch.no_grad()
    def sample(self, x, steps=20, order=3, skip_type="time_uniform", method="singlestep", denoise=False, atol=0.0078, rtol=0.05):
        t_0, t_T = 1. / self.ns.N, 1.
        B = next(iter(x.values())).shape[0]
        if method == "adaptive":
            x = self.adaptive(x, order, t_T, t_0, atol=atol, rtol=rtol)
        elif method == "multistep":
            ts = self.steps(skip_type, t_T, t_0, steps)
            vt = ts[0].expand(B)
            ml, tl = [self.fn(x, vt)], [vt]
            for io in range(1, order):
                vt = ts[io].expand(B)
                x = self.first(x, tl[-1], vt, ms=ml[-1]) if io == 1 else self.multi2(x, ml, tl, vt)
                ml.append(self.fn(x, vt))
                tl.append(vt)
            for step in range(order, steps + 1):
                vt = ts[step].expand(B)
                x = self.first(x, tl[-1], vt, ms=ml[-1]) if order == 1 else self.multi2(x, ml, tl, vt)
                ml, tl = ml[1:] + [None], tl[1:] + [vt]
                if step < steps:
                    ml[-1] = self.fn(x, vt)
        else:
            K = steps // 3 + 1
            orders = {3: [3] * (K - 2) + [2, 1] if steps % 3 == 0 else ([3] * (K - 1) + [1] if steps % 3 == 1 else [3] * (K - 1) + [2]),
                      2: [2] * (steps // 2) + ([1] if steps % 2 else []), 1: [1] * steps}[order]
            ts = self.steps(skip_type, t_T, t_0, steps)
            i = 0
            for o in orders:
                vs, vt = ts[i].expand(B), ts[i + o].expand(B)
                h = self.ns.lam(ts[i + o]) - self.ns.lam(ts[i])
                r1 = None if o <= 1 else (self.ns.lam(ts[i + 1]) - self.ns.lam(ts[i])) / h
                r2 = None if o <= 2 else (self.ns.lam(ts[i + 2]) - self.ns.lam(ts[i])) / h
                x = self.first(x, vs, vt) if o == 1 else (self.second(x, vs, vt, r1=r1) if o == 2 else self.third(x, vs, vt, r1=r1, r2=r2))
                i += o
        if denoise:
            self_px0, self.predict_x0 = self.predict_x0, True
            x = self.fn(x, torch.ones(B) * t_0)
            self.predict_x0 = self_px0
        return x
