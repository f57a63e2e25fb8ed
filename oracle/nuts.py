"""fp64 restatement of one No-U-Turn transition with counter-based randomness (TEST ORACLE).

Test infrastructure only -- see ``oracle/__init__.py``.  Parity unpinned against TFP: the reference
reaches ``tfp.mcmc.NoUTurnSampler`` through bayes_vi/inference/mcmc/transition_kernels.py:238-246 and
TFP is not installed here, so the algorithm is restated from the publication (Hoffman & Gelman 2014,
with multinomial sampling inside subtrees and biased progressive sampling between them, built
iteratively with checkpointed U-turn checks) -- the same restatement as the batched host kernel
``bayes_vi_amd.inference.mcmc.samplers.NoUTurnSamplerKernel`` -- and is pinned by posterior moments
(tests) rather than by TFP outputs.

The random numbers are the Philox4x32-10 draws of the device kernel (``oracle.philox``): momenta on
stream 0, (direction, subtree-accept uniform) = words 0 and 1 of block ``depth`` on stream 2, the
multinomial uniform of the n-th leapfrog of the transition = word 0 of block n on stream 3; so a
device chain can be followed draw for draw.
"""
from __future__ import annotations

import numpy as np

from . import philox


def _log_add_exp(a, b):
    m = max(a, b)
    if m == -np.inf:
        return -np.inf
    return m + np.log(np.exp(a - m) + np.exp(b - m))


def _turning(aq, ap, bq, bp):
    dq = bq - aq
    return bool(dq @ ap <= 0) or bool(dq @ bp <= 0)


def nuts_step(value_and_grad, z, lp, g, chain, seed, step, step_size, max_tree_depth=5, max_energy_diff=1000.0):
    """One transition of chain number ``chain`` from position z[P] with cached lp, g[P].
    value_and_grad(q[1, P]) -> (lp[1], g[1, P]).  Returns dict(z, lp, g, moved, log_accept_ratio,
    leapfrogs, diverged, margins) where margins collects |a - b| of every comparison a < b that
    decided something, so that a caller can tell whether a float32 implementation may legitimately
    have branched differently."""
    P = z.shape[0]
    n_chains = chain + 1
    p0 = philox.normals(seed, step, n_chains, P)[chain].astype(np.float64)
    dirs = philox.draw_blocks(seed, step, 2, n_chains, max_tree_depth)[chain]
    leaves = philox.draw_blocks(seed, step, 3, n_chains, 2 ** max_tree_depth)[chain]
    h0 = -lp + 0.5 * p0 @ p0
    ql, pl, gl = z.copy(), p0.copy(), g.copy()
    qr, pr, gr = z.copy(), p0.copy(), g.copy()
    zs, lps, gs = z.copy(), lp, g.copy()
    energy = -h0          # tfp's kernel_results.energy: log-density minus kinetic energy at the selected point
    log_w, sum_acc, n_leap, leaf_ctr = 0.0, 0.0, 0, 0
    active, diverged, moved = True, False, False
    margins = []
    ckq = np.zeros((max_tree_depth + 1, P)); ckp = np.zeros((max_tree_depth + 1, P))
    depth = 0
    while depth < max_tree_depth and active:
        u_dir = float(philox.u01(dirs[depth, 0])); margins.append(abs(u_dir - 0.5))
        go_right = u_dir < 0.5
        log_u_sub = np.log(float(philox.u01(dirs[depth, 1])))
        se = step_size if go_right else -step_size
        q, p, gq = (qr.copy(), pr.copy(), gr.copy()) if go_right else (ql.copy(), pl.copy(), gl.copy())
        sub_log_w, sub_q, sub_g, sub_lp, sub_e = -np.inf, zs.copy(), gs.copy(), lps, energy
        sub_ok, turn_sub = True, False
        for leaf in range(2 ** depth):
            p = p + 0.5 * se * gq
            q = q + se * p
            lv, gv = value_and_grad(q[None])
            lpq, gq = float(lv[0]), gv[0]
            p = p + 0.5 * se * gq
            dh = (-lpq + 0.5 * p @ p) - h0
            if not np.isfinite(dh):
                dh = np.inf
            div_now = dh > max_energy_diff
            diverged |= div_now
            ok = not div_now
            n_leap += 1
            sum_acc += np.exp(min(-dh, 0.0))
            lw = -dh
            new_w = _log_add_exp(sub_log_w, lw)
            log_ul = np.log(float(philox.u01(leaves[leaf_ctr, 0]))); leaf_ctr += 1
            if ok:
                margins.append(abs(log_ul - (lw - new_w)))
                if log_ul < lw - new_w:
                    sub_q, sub_g, sub_lp, sub_e = q.copy(), gq.copy(), lpq, lpq - 0.5 * p @ p
                sub_log_w = new_w
            hi = bin(leaf >> 1).count('1')
            if leaf % 2 == 0:
                ckq[hi], ckp[hi] = q, p
            else:
                trailing = 0
                m = leaf
                while m & 1:
                    trailing += 1; m >>= 1
                for k in range(hi - trailing + 1, hi + 1):
                    a, b = ((ckq[k], ckp[k]), (q, p)) if go_right else ((q, p), (ckq[k], ckp[k]))
                    dq = b[0] - a[0]
                    margins.append(min(abs(dq @ a[1]), abs(dq @ b[1])) / (np.abs(dq) @ np.abs(a[1]) + 1e-300))
                    turn_sub |= ok and _turning(a[0], a[1], b[0], b[1])
            sub_ok = ok and not turn_sub
            if not sub_ok:
                break
        if go_right:
            qr, pr, gr = q, p, gq
        else:
            ql, pl, gl = q, p, gq
        complete = sub_ok
        if complete:
            margins.append(abs(log_u_sub - (sub_log_w - log_w)))
            if log_u_sub < sub_log_w - log_w:
                zs, gs, lps, moved, energy = sub_q, sub_g, sub_lp, True, sub_e
            log_w = _log_add_exp(log_w, sub_log_w)
            dq = qr - ql
            margins.append(min(abs(dq @ pl), abs(dq @ pr)) / (np.abs(dq) @ np.abs(pl) + 1e-300))
        active = complete and not _turning(ql, pl, qr, pr)
        depth += 1
    return dict(z=zs, lp=lps, g=gs, moved=moved, log_accept_ratio=float(np.log(max(sum_acc / max(n_leap, 1), 1e-30))),
                leapfrogs=n_leap, diverged=bool(diverged), margin=min(margins) if margins else np.inf,
                energy=float(energy))
