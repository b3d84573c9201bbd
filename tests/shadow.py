"""One-step shadowing: the parity statement that survives a chaotic iteration.

Some hard instances crawl for 50-100 iterations along a path on which every rollout amplifies a perturbation (the cost moves
by up to three orders of magnitude more than the iterate did).  Two correct implementations that differ only in rounding -- the
GPU kernel and the C oracle, or two builds of the C oracle itself (-ffp-contract=off / fast) -- then drift apart smoothly, with
the SAME step lengths, from 1e-16 to 1e-3 relative, and only then take different step lengths and iteration counts.  Comparing
such solves end to end says nothing about either.  What can be checked, and is checked here for EVERY accepted step of the engine
under test: take its iterate k (trajectory and the carried state: merit weight, regularisation, second-order switch, open /
closed gaps), let the oracle do ONE iteration from it, and compare with the engine's iterate k + 1 -- same step length, same
cost, same trajectory.  One step amplifies rounding by at most the one-step factor, so the tolerance stays tight and a real
defect of the kernel (a wrong gain, a wrong candidate, a wrong acceptance test) cannot hide behind "chaos".
"""
import numpy as np

from oracle import cport, ddp as oddp


PROBE_REL = 1e-12      # relative size (per entry) of the perturbations of the iterate
SENS_DRAWS = 3         # perturbed repeats of every step: the oracle's own sensitivity there
PROBE_DRAWS = 16       # more of them where engine and oracle choose different step lengths
SENS_FACTOR = 30.0     # a step may deviate by this multiple of the oracle's own response to a PROBE_REL perturbation ...
COND_ILL_POSED = 1e12  # cond(Quu) from which a sweep's gains count as determined by rounding (eps * cond = 1e-4)
BAND_FACTOR = 2.0      # a step-length decision is "inside the oracle's noise" if its margin is within this multiple of the band
ONE_STEP_FLOOR = 1e-9  # ... or by this much (relative cost), whichever is larger
ONE_STEP_MEDIAN_RTOL = 1e-7    # and the median step of an instance by no more than this


def _one_step(cst, o1, x0, P, x, u, resume, model, variant=None):
    xo, uo, st, tr = cport.solve_trace(cst, o1, x0, P, x, u, model=model, variant=variant, resume=resume)
    return xo, uo, st, tr, (float(st[3]) if int(st[1]) == 1 else 0.0)      # step length taken (0.0: line search exhausted)


def _max_cond_quu(cst, model, x0, P, xs, us, mu):
    """largest cond(Quu + mu I) over the knots of the (Gauss-Newton) sweep from the iterate (xs, us): numpy oracle"""
    from oracle import models as omodels
    m = omodels.make_model(model, cst)
    xs = np.array(xs, dtype=float); xs[0] = x0
    N = us.shape[0]
    d = oddp.defects(m, xs, us, P)
    _, Vx, _, Vxx, _, _ = m.cost_derivs(xs[N], None, P[N], N)
    worst = 0.0
    for k in range(N - 1, -1, -1):
        fx, fu = m.f_jac(xs[k], us[k], P[k])
        _, lx, lu, lxx, lux, luu = m.cost_derivs(xs[k], us[k], P[k], k)
        vp = Vx + Vxx @ d[k]
        Quu = luu + fu.T @ Vxx @ fu + mu * np.eye(m.nu)
        Qux = lux + fu.T @ Vxx @ fx
        if not np.all(np.isfinite(Quu)):
            return float("inf")
        worst = max(worst, float(np.linalg.cond(Quu)))
        Kk = -np.linalg.lstsq(Quu, Qux, rcond=None)[0]; kv = -np.linalg.lstsq(Quu, lu + fu.T @ vp, rcond=None)[0]
        Vx = lx + fx.T @ vp + Qux.T @ kv
        Vxx = lxx + fx.T @ Vxx @ fx + Qux.T @ Kk
        Vxx = 0.5 * (Vxx + Vxx.T)
    return worst


def shadow_one_instance(cst, opts: dict, x0, P, states, model="srbd13", variant=None, seed=12345):
    """states[k] = dict(x, u, cost, alpha, gap, mu, rho, iters, status, converged) of the engine under test cut at max_iters = k
    (k = 0: the warm start).  For every accepted step k -> k + 1 of the engine: the oracle's ONE iteration from the engine's
    iterate k, and the same from SENS_DRAWS copies of that iterate perturbed by PROBE_REL -- the oracle's own sensitivity at that
    step, which is what the engine's deviation is measured against.  -> list of per-step records."""
    a0 = opts.get("alpha_0", 1.0)
    fac = opts.get("line_search_decrease_factor", 0.5)
    so = opts.get("second_order", 1)
    rng = np.random.default_rng(seed)
    o1 = oddp.DdpOptions(**dict(opts, max_iters=1))
    out = []
    for k in range(len(states) - 1):
        s, t = states[k], states[k + 1]
        if t["iters"] != k + 1:                # the engine stopped before an accepted step k + 1
            break
        resume = dict(rho=s["rho"], theta=1.0 if (k > 0 and so and s["alpha"] == a0) else 0.0, closed=(k > 0 and s["gap"] == 0.0),
                      mu=s["mu"])
        xo, uo, st, tr, a_o = _one_step(cst, o1, x0, P, s["x"], s["u"], resume, model, variant)
        J = max(abs(t["cost"]), 1e-300)
        rec = dict(k=k + 1, alpha_engine=float(t["alpha"]), alpha_oracle=a_o, rel_cost=abs(st[0] - t["cost"]) / J,
                   rel_x=float(np.max(np.abs(xo - t["x"])) / max(np.max(np.abs(t["x"])), 1e-300)),
                   rel_u=float(np.max(np.abs(uo - t["u"])) / max(np.max(np.abs(t["u"])), 1e-300)))
        # the oracle against itself, from perturbed copies of the same iterate
        alphas, sens = {a_o}, 0.0

        def perturbed():
            xp = s["x"] * (1.0 + PROBE_REL * rng.standard_normal(s["x"].shape))
            up = s["u"] * (1.0 + PROBE_REL * rng.standard_normal(s["u"].shape))
            xp[0] = s["x"][0]
            return xp, up
        for _ in range(SENS_DRAWS):
            _, _, sp, _, a_p = _one_step(cst, o1, x0, P, *perturbed(), resume, model, variant)
            alphas.add(a_p)
            if a_p == a_o:
                sens = max(sens, abs(sp[0] - st[0]) / max(abs(st[0]), 1e-300))
        rec["noise_explained"] = False
        if rec["alpha_engine"] != a_o:           # different step lengths from the same iterate: look harder
            alphas.add(_one_step(cst, o1, x0, P, s["x"], s["u"], resume, model, "fast" if variant != "fast" else "off")[4])
            # the decisive candidate: the larger of the two step lengths -- one side accepted it, the other did not.  The oracle's
            # Armijo margin for it, from this iterate and from the perturbed copies: is its distance from zero inside the band over
            # which the oracle's OWN margin moves under 1e-12 perturbations?  (Seen: cond(Quu) 5e13 at one knot, the gains along
            # its near-null direction are rounding, J(alpha) moves by 2 % of J whatever the size of the perturbation.)
            a_dec = max(rec["alpha_engine"], a_o)

            def margin_at(trace):
                if not trace:
                    return None
                m, a, j = trace[-1]["margin"], a0, 0
                while j < len(m) and a > a_dec * (1 + 1e-12):
                    a *= fac; j += 1
                return float(m[j]) if j < len(m) and abs(a - a_dec) <= 1e-12 * a and np.isfinite(m[j]) else None
            ms = [margin_at(tr)]
            for _ in range(PROBE_DRAWS):
                r_p = _one_step(cst, o1, x0, P, *perturbed(), resume, model, variant)
                alphas.add(r_p[4])
                ms.append(margin_at(r_p[3]))
            if ms[0] is not None and all(v is not None for v in ms):
                band = max(ms) - min(ms)
                rec["margin_dec_rel"], rec["margin_band_rel"] = ms[0] / J, band / J
                rec["noise_explained"] = abs(ms[0]) <= BAND_FACTOR * band
            if not rec["noise_explained"] and len(alphas) == 1:
                # third rule (soak run, seed 870703 step 26: the oracle's rollout at the decisive step length overflows from this
                # iterate and from every perturbed copy, the engine's does not): is the SWEEP itself ill-posed here?  cond(Quu) of
                # the oracle's own sweep from this iterate; at 1e12 and beyond the gains along the near-null direction are set by
                # the elimination order (tests/explain_step.py: there the two sets of gains differ by O(1) at cond 2e16, and the
                # oracle's rollout WITH THE ENGINE'S GAINS reproduces the engine's cost to 3e-11)
                rec["cond_quu"] = _max_cond_quu(cst, model, x0, P, s["x"], s["u"], s["mu"])
                rec["noise_explained"] = rec["cond_quu"] >= COND_ILL_POSED
        rec["oracle_alphas"] = sorted(alphas)
        rec["unstable"] = len(alphas) > 1        # the oracle's own choice of step length flips under a 1e-12 perturbation
        rec["sens"] = sens
        out.append(rec)
    return out


def engine_states_from_oracle(cst, opts: dict, x0, P, xs, us, model="srbd13", variant=None, kmax=None):
    """The `states` list of shadow_one_instance produced by a CPU build of the oracle (stand-in for the GPU engine in the CPU test)."""
    states = []
    kmax = opts["max_iters"] if kmax is None else kmax
    for k in range(kmax + 1):
        o = oddp.DdpOptions(**dict(opts, max_iters=k))
        x, u, st = cport.solve_batch(cst, o, x0[None], P[None], xs[None], us[None], model=model, variant=variant)
        states.append(dict(x=x[0], u=u[0], cost=st[0, 0], iters=int(st[0, 1]), converged=int(st[0, 2]), alpha=st[0, 3], gap=st[0, 4],
                           mu=st[0, 5], status=int(st[0, 6]), rho=st[0, 7]))
        if states[-1]["iters"] < k:
            break
    return states


def summarize(records):
    """An instance's steps, sorted into: same step length at a stable step (deviation measured against the oracle's own
    sensitivity), unstable steps (the oracle's own step length flips under a PROBE_REL perturbation: nothing to compare), and
    step-length mismatches (explained iff the step is an unstable one, or the decisive margin sits inside its own noise band)."""
    if not records:
        return dict(steps=0, violations=[], alpha_mismatch=[], alpha_unexplained=[], unstable_steps=0, max_rel_cost=0.0,
                    median_rel_cost=0.0, max_ratio=0.0)
    stable = [r for r in records if r["alpha_engine"] == r["alpha_oracle"] and not r["unstable"]]
    mis = [r for r in records if r["alpha_engine"] != r["alpha_oracle"]]
    viol = [dict(k=r["k"], rel_cost=r["rel_cost"], sens=r["sens"]) for r in stable
            if r["rel_cost"] > max(ONE_STEP_FLOOR, SENS_FACTOR * r["sens"])]
    # explained: the oracle's own choice is not unanimous there (seen: one iterate from which 20 perturbed repeats of the oracle
    # take step lengths from 2^-20 to 2^-5 and the engine 2^-2 -- and the second CPU build 2^-2 as well, one step later)
    # ... or the oracle's Armijo margin for the decisive candidate lies inside the band over which that margin itself moves
    unexplained = [r["k"] for r in mis if not (r["unstable"] or r.get("noise_explained"))]
    return dict(steps=len(records), violations=viol,
                alpha_mismatch=[dict(k=r["k"], engine=r["alpha_engine"], oracle=r["alpha_oracle"], oracle_perturbed=r["oracle_alphas"],
                                     margin_rel=r.get("margin_dec_rel"), margin_band_rel=r.get("margin_band_rel"), cond_quu=r.get("cond_quu"))
                                for r in mis],
                alpha_unexplained=unexplained, unstable_steps=sum(r["unstable"] for r in records),
                max_rel_cost=max([r["rel_cost"] for r in stable] or [0.0]),
                median_rel_cost=float(np.median([r["rel_cost"] for r in records])),
                max_ratio=max([r["rel_cost"] / max(r["sens"], 1e-16) for r in stable if r["rel_cost"] > ONE_STEP_FLOOR] or [0.0]))


# ---------------------------------------------------------------------------------------------------------------------------------
# whole batches: GPU engine vs C oracle, with every instance on another path explained step by step
# ---------------------------------------------------------------------------------------------------------------------------------
def engine_states(model, N, opts, consts, x0, P, xs, us, kmax):
    """states[b][k] of the HIP engine for the D instances given, cut at max_iters = 0..kmax (D instances per launch)."""
    from srbd_horizon_amd.engine import DdpEngine
    D = x0.shape[0]
    eng = DdpEngine(model, N, D, opts=dict(opts, max_iters=0), consts=consts)
    eng.set_initial_state(x0)
    states = [[] for _ in range(D)]
    for k in range(kmax + 1):
        eng.set_options(max_iters=k)
        eng.set_x_warmstart(xs); eng.set_u_warmstart(us)
        x, u = eng.solve(P)
        st = eng.stats
        for b in range(D):
            states[b].append(dict(x=x[b].copy(), u=u[b].copy(), cost=float(st["cost"][b]), iters=int(st["iters"][b]),
                                  converged=int(st["converged"][b]), alpha=float(st["alpha"][b]), gap=float(st["gap"][b]),
                                  mu=float(st["mu"][b]), status=int(st["status"][b]), rho=float(st["rho"][b])))
    eng.close()
    return states


def first_split(states, trace, opts_alpha0=1.0, opts_factor=0.5):
    """First accepted step at which the engine (states, or a second oracle line-search trace) and a full oracle run (its
    line-search trace) take different step lengths, and how far their costs had drifted apart BEFORE that step.
    -> dict or None (same step lengths throughout)."""
    acc = [r for r in trace if r["alpha"] > 0.0]
    if states and "J_new" in states[0]:                                     # a trace: J before step k + 1, alpha of step k + 1
        other = [r for r in states if r["alpha"] > 0.0]
        seq = [(r["J"], r["alpha"]) for r in other]
    else:
        seq = [(states[k]["cost"], states[k + 1]["alpha"]) for k in range(len(states) - 1) if states[k + 1]["iters"] == k + 1]
    for k in range(min(len(seq), len(acc))):
        if seq[k][1] != acc[k]["alpha"]:
            J = max(abs(acc[k]["J"]), 1e-300)
            # the oracle's own Armijo margin (relative to |J|) for the step length the engine took, where the oracle tried it:
            # > 0 = the oracle rejected it.  Not at rounding level: the two sides are no longer at the same iterate (drift_before)
            margins, a, j, m_rel = acc[k]["margin"], opts_alpha0, 0, None
            while j < len(margins) and a > seq[k][1] * (1 + 1e-12):
                a *= opts_factor; j += 1
            if j < len(margins) and abs(a - seq[k][1]) <= 1e-12 * a:
                m_rel = float(margins[j] / J)
            return dict(step=k + 1, alpha_engine=float(seq[k][1]), alpha_oracle=float(acc[k]["alpha"]),
                        drift_before=float(abs(seq[k][0] - acc[k]["J"]) / J), oracle_margin_rel_at_engine_alpha=m_rel,
                        oracle_margin_rel_at_its_own_alpha=float(margins[-1] / J) if len(margins) else None)
    return None


def explain_divergent(model, N, opts, engine_over, consts, cst, batch, idx):
    """For the instances idx of `batch` (those whose GPU iteration count differs from the oracle's): one-step shadowing of the
    whole GPU path, the first split against the full oracle path, and the first split of the second CPU build of the oracle
    (-ffp-contract=fast) against the first.  opts: the algorithm's options (both sides), engine_over: scheduling options of the
    engine.  -> list of records (one per instance)."""
    idx = np.asarray(idx)
    kmax = opts["max_iters"]
    sub = {k: batch[k][idx] for k in ("x0", "params", "xs", "us")}
    gs = engine_states(model, N, dict(opts, **engine_over), consts, sub["x0"], sub["params"], sub["xs"], sub["us"], kmax)
    out = []
    for j, b in enumerate(idx):
        a = (cst, oddp.DdpOptions(**opts), sub["x0"][j], sub["params"][j], sub["xs"][j], sub["us"][j])
        xo, uo, so, tr_off = cport.solve_trace(*a, model=model)
        xf, uf, sf, tr_fast = cport.solve_trace(*a, model=model, variant="fast")
        steps = shadow_one_instance(cst, opts, sub["x0"][j], sub["params"][j], gs[j], model=model)
        end = gs[j][kmax]
        rec = dict(instance=int(b), gpu_iters=end["iters"], oracle_iters=int(so[1]), oracle_fast_iters=int(sf[1]),
                   gpu_status=end["status"], oracle_status=int(so[6]),
                   shadow=summarize(steps),
                   split_gpu=first_split(gs[j], tr_off), split_cpu_fast=first_split(tr_fast, tr_off),
                   end_linf=float(max(np.max(np.abs(end["x"] - xo)), np.max(np.abs(end["u"] - uo)))),
                   end_rel_cost=float(abs(end["cost"] - so[0]) / max(abs(so[0]), 1e-300)))
        out.append(rec)
    return out


def assert_shadowed(rec):
    """Every accepted step of the GPU path is the oracle's step from the same iterate, as far as the oracle itself is
    determined there."""
    s = rec["shadow"]
    assert s["steps"] == rec["gpu_iters"], rec
    assert s["violations"] == [], rec              # stable step, same step length: cost within 10 x the oracle's own sensitivity
    assert s["alpha_unexplained"] == [], rec       # another step length only where the oracle's own choice (or its margin) flips
    # every mismatch above is explained one by one; the caps only guard against a systematic defect hiding behind "unstable" (seen
    # in the soak run: one instance with five unstable steps in a row, step lengths 1e-7 .. 0.5 from the same iterate)
    assert len(s["alpha_mismatch"]) <= max(4, s["steps"] // 3) and s["unstable_steps"] <= max(5, s["steps"] // 3), rec
    assert s["median_rel_cost"] <= ONE_STEP_MEDIAN_RTOL, rec


def check_batch(model, N, batch, opts, engine_over, cst, threads=16):
    """GPU solve of a whole batch through the queue vs the C oracle (both CPU builds).  -> dict(x, u, st, xo, uo, so, same,
    explained: the records of explain_divergent for every instance with another iteration count, n_cpu_pair: instances on which
    the two CPU builds of the oracle take different iteration counts)."""
    from srbd_horizon_amd.engine import DdpEngine
    B = batch["x0"].shape[0]
    eng = DdpEngine(model, N, B, opts=dict(opts, **engine_over), consts=batch["consts"])
    eng.set_initial_state(batch["x0"]); eng.set_x_warmstart(batch["xs"]); eng.set_u_warmstart(batch["us"])
    x, u = eng.solve(batch["params"])
    st = eng.stats.copy()
    qi = eng.queue_info()
    eng.close()
    return compare_batch(model, N, batch, opts, engine_over, cst, x, u, st, threads, queue_info=qi)


def compare_batch(model, N, batch, opts, engine_over, cst, x, u, st, threads=16, queue_info=None):
    """The oracle side of check_batch for results (x, u, st) the engine has already produced from `batch` (x0, params, xs, us,
    consts) -- e.g. one tick of a device-resident receding-horizon loop, whose inputs the caller has rebuilt on the host."""
    o = oddp.DdpOptions(**opts)
    xo, uo, so = cport.solve_batch(cst, o, batch["x0"], batch["params"], batch["xs"], batch["us"], threads=threads, model=model)
    _, _, sf = cport.solve_batch(cst, o, batch["x0"], batch["params"], batch["xs"], batch["us"], threads=threads, model=model,
                                 variant="fast")
    it_o = so[:, 1].astype(int)
    same = st["iters"] == it_o
    idx = np.nonzero(~same)[0]
    explained = explain_divergent(model, N, opts, engine_over, batch["consts"], cst, batch, idx) if idx.size else []
    return dict(x=x, u=u, st=st, xo=xo, uo=uo, so=so, same=same, explained=explained, queue_info=queue_info,
                n_cpu_pair=int((sf[:, 1] != so[:, 1]).sum()), cpu_pair_idx=np.nonzero(sf[:, 1] != so[:, 1])[0])


def parity_record(res):
    """The PARITY-COUNT payload of a check_batch result (tests/conftest.py report_parity)."""
    ex = res["explained"]
    return dict(differ=len(ex), cpu_pair_differ=res["n_cpu_pair"],
                both=len(set(r["instance"] for r in ex) & set(res["cpu_pair_idx"].tolist())),
                instances=[dict(i=r["instance"], it=(r["gpu_iters"], r["oracle_iters"], r["oracle_fast_iters"]),
                                split=(r["split_gpu"] or {}).get("step"), drift=(r["split_gpu"] or {}).get("drift_before"),
                                alphas=((r["split_gpu"] or {}).get("alpha_engine"), (r["split_gpu"] or {}).get("alpha_oracle")),
                                margin=((r["split_gpu"] or {}).get("oracle_margin_rel_at_engine_alpha"),
                                        (r["split_gpu"] or {}).get("oracle_margin_rel_at_its_own_alpha")),
                                cpu_split=(r["split_cpu_fast"] or {}).get("step"), cpu_drift=(r["split_cpu_fast"] or {}).get("drift_before"),
                                step_max=r["shadow"]["max_rel_cost"], step_med=r["shadow"]["median_rel_cost"],
                                ratio=r["shadow"]["max_ratio"], unstable=r["shadow"]["unstable_steps"],
                                amis=r["shadow"]["alpha_mismatch"], end_linf=r["end_linf"],
                                status=(r["gpu_status"], r["oracle_status"])) for r in ex])
