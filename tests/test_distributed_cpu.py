"""world_size-2 gloo test of the row-sharded path: dataset.shard -> local partial sums ->
one all-reduce of [C, 1+P] -> priors and bijector terms added once.  The local partial sums
come from the fp64 oracle here (no GPU in this suite); on the GPU box the same
ShardedLogLikelihood.logp_grad / Posterior code runs over the HIP kernels and RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bayes_vi_amd import bijectors as tfb
from bayes_vi_amd import distributions as tfd
from bayes_vi_amd.fused_target import Posterior, ShardedLogLikelihood
from bayes_vi_amd.inference.mcmc import samplers
from bayes_vi_amd.utils.datasets import TensorDataset
from oracle import closed_form as cf

N, D, C = 1001, 5, 6


class OracleLinreg(ShardedLogLikelihood):
    def __init__(self, X, y, group=None):
        super().__init__('cpu', group)
        self.X, self.y = X.numpy(), y.numpy()
        self.n_params = X.shape[1] + 2

    def _call(self, theta, logp, grad):
        lp, g = cf.linreg(self.X, self.y, theta.numpy())
        logp.copy_(torch.tensor(lp, dtype=torch.float32))
        if grad is not None:
            grad.copy_(torch.tensor(g, dtype=torch.float32))


def make_data():
    g = torch.Generator().manual_seed(0)
    X = torch.randn(N, D, generator=g); w = torch.randn(D, generator=g)
    y = 0.3 + X @ w + torch.randn(N, generator=g)
    z = torch.cat([0.3 + 0.1 * torch.randn(C, 1, generator=g), w + 0.1 * torch.randn(C, D, generator=g),
                   0.5 + 0.1 * torch.randn(C, 1, generator=g)], -1)
    return X, y, z


def make_posterior(X, y, group):
    return Posterior(OracleLinreg(X, y, group),
                     [(tfd.Normal(0., 10.), 1), (tfd.Normal(torch.zeros(D), 10.), D), (tfd.HalfNormal(10.), 1)],
                     [tfb.Identity(), tfb.Identity(), tfb.Softplus()])


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    X, y, z = make_data()
    shard = TensorDataset(X, y).shard(rank, world)
    post = make_posterior(shard.features, shard.targets, dist.group.WORLD)
    assert post.likelihood.sharded
    lp, g = post.unconstrained_log_prob_and_grad(z)
    lpv, _ = post.unconstrained_log_prob_and_grad(z, want_grad=False)
    # the fixed-order variant (all-gather + sum in rank order) gives the same totals on every rank
    post.likelihood.fixed_order_reduction = True
    lpf, gf = post.unconstrained_log_prob_and_grad(z)
    post.likelihood.fixed_order_reduction = False
    both = [torch.empty_like(gf) for _ in range(world)]
    dist.all_gather(both, gf)
    assert all(torch.equal(both[0], t) for t in both)
    assert torch.allclose(lpf, lp, rtol=1e-6) and torch.allclose(gf, g, rtol=1e-5, atol=1e-4)
    # a few HMC steps with the same seed on every rank: chains stay replicated bit for bit
    k = samplers.HamiltonianMonteCarloKernel(post.unconstrained_log_prob_and_grad, step_size=0.005, num_leapfrog_steps=3)
    gen = torch.Generator().manual_seed(123)
    zs, tr = samplers.sample_chain(k, z, 5, 2, generator=gen)
    gathered = [torch.empty_like(zs) for _ in range(world)]
    dist.all_gather(gathered, zs)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    if rank == 0:
        torch.save(dict(lp=lp, g=g, lpv=lpv, zs=zs, rows=shard.cardinality()), out)
    dist.destroy_process_group()


def test_row_sharded_posterior_world2(tmp_path):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    X, y, z = make_data()
    assert res['rows'] == N // 2
    post = make_posterior(X, y, None)
    assert not post.likelihood.sharded
    lp, g = post.unconstrained_log_prob_and_grad(z)
    # shard sums are float32 totals of fp64 partials: equal to the unsharded result to f32 rounding
    np.testing.assert_allclose(res['lp'].numpy(), lp.numpy(), rtol=2e-6)
    np.testing.assert_allclose(res['g'].numpy(), g.numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(res['lpv'].numpy(), lp.numpy(), rtol=2e-6)
    # priors are counted once, not once per shard
    spec = cf.linreg_model(X.numpy(), y.numpy())
    rlp, rg = spec.log_post_unconstrained(z.numpy())
    np.testing.assert_allclose(res['lp'].numpy(), rlp, rtol=2e-6)
    k = samplers.HamiltonianMonteCarloKernel(post.unconstrained_log_prob_and_grad, step_size=0.005, num_leapfrog_steps=3)
    zs, _ = samplers.sample_chain(k, z, 5, 2, generator=torch.Generator().manual_seed(123))
    np.testing.assert_allclose(res['zs'].numpy(), zs.numpy(), rtol=1e-4, atol=1e-5)


def test_row_sharded_posterior_world8_rows_not_divisible(tmp_path):
    """Eight ranks on 1001 rows (one shard of 126 rows, seven of 125): the all-reduce and the fixed-order all-gather variant
    (both run inside `_worker`, which also checks that every rank ends with the same bits and that the replicated
    HMC chains stay identical) give the unsharded posterior; priors counted once."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(8, port, out), nprocs=8, join=True)
    res = torch.load(out)
    X, y, z = make_data()
    assert res['rows'] in (N // 8, N // 8 + 1)          # contiguous blocks [N r / 8, N (r + 1) / 8)
    post = make_posterior(X, y, None)
    lp, g = post.unconstrained_log_prob_and_grad(z)
    np.testing.assert_allclose(res['lp'].numpy(), lp.numpy(), rtol=2e-6)
    np.testing.assert_allclose(res['g'].numpy(), g.numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(res['lpv'].numpy(), lp.numpy(), rtol=2e-6)
    k = samplers.HamiltonianMonteCarloKernel(post.unconstrained_log_prob_and_grad, step_size=0.005, num_leapfrog_steps=3)
    zs, _ = samplers.sample_chain(k, z, 5, 2, generator=torch.Generator().manual_seed(123))
    np.testing.assert_allclose(res['zs'].numpy(), zs.numpy(), rtol=1e-4, atol=1e-5)


def test_shard_levels_follow_row_shares():
    """The arithmetic of bvi_logp_rebase_shard (include/bvi.h) restated on the host for eight shards of unequal sizes:
    every shard's level moves by its share of the rows, the shares sum to one, so the shard levels always add up to the
    total level, and a shard whose log-likelihood is its share of the total keeps relative values of O(fluctuation)."""
    rows = np.array([126, 125, 125, 125, 125, 125, 125, 125], np.float64) * 1000
    share = rows / rows.sum()
    assert share.sum() == pytest.approx(1.0, abs=1e-15)
    rng = np.random.default_rng(0)
    own, total = np.zeros(8), 0.0
    per_row = -1.4
    for step in range(50):
        # the chains' best total log-likelihood drifts by a few units per transition; each shard holds its share
        # of it plus a fluctuation of the size of sqrt(rows)
        best_total = per_row * rows.sum() + 3.0 * step + rng.normal()
        shard_ll = best_total * share + rng.normal(size=8) * np.sqrt(rows)
        shard_ll += (best_total - shard_ll.sum()) / 8
        rel = np.float32(shard_ll - own)                     # what each shard's kernel returns (float)
        m = np.float32(rel.astype(np.float64).sum())         # the all-reduced relative value of the best chain
        own += float(m) * share; total += float(m)           # bvi_logp_rebase_shard on every rank
        assert own.sum() == pytest.approx(total, rel=1e-14)
        if step:
            assert np.abs(rel).max() < 5 * np.sqrt(rows.max()) + 10
    assert total == pytest.approx(best_total, abs=1.0)


def _fit_worker(rank, world, port, out):
    """MCMC.fit without a seed on a row-sharded posterior whose kernel draws from torch generators (a host kernel):
    every rank has its own global RNG state, and the chains must still be identical on all ranks."""
    import collections
    from bayes_vi_amd.inference.mcmc import MCMC
    from bayes_vi_amd.inference.mcmc.transition_kernels import HamiltonianMonteCarlo
    from bayes_vi_amd.model import Model
    from bayes_vi_amd import symbolic
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(1000 + rank)
    X, y, _ = make_data()
    shard = TensorDataset(X, y).shard(rank, world)
    priors = collections.OrderedDict(b=tfd.Normal(0., 10.), w=tfd.Normal(torch.zeros(D), 10.), scale=tfd.HalfNormal(10.))
    model = Model(priors, lambda b, w, scale, features: tfd.Normal(loc=b + symbolic.matvec(features, w), scale=scale))
    mcmc = MCMC.__new__(MCMC)                      # the driver object around the oracle-backed shard likelihood
    mcmc.model, mcmc.dataset = model, shard
    mcmc.posterior = make_posterior(shard.features, shard.targets, dist.group.WORLD)
    mcmc.device, mcmc.group = torch.device('cpu'), dist.group.WORLD
    mcmc.transition_kernel = HamiltonianMonteCarlo(step_size=0.005, num_leapfrog_steps=3)
    res = mcmc.fit(num_chains=3, num_samples=6, num_burnin_steps=2, progress_bar=False)
    flat = torch.cat([s.reshape(6, 3, -1) for s in res.samples], -1)
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    acc = res.trace['is_accepted'].to(torch.float32)
    both = [torch.empty_like(acc) for _ in range(world)]
    dist.all_gather(both, acc)
    assert all(torch.equal(both[0], t) for t in both)
    if rank == 0:
        torch.save(dict(flat=flat), out)
    dist.destroy_process_group()


def test_unseeded_fit_keeps_row_sharded_chains_replicated(tmp_path):
    outs = []
    for i in range(2):                 # two unseeded runs are different runs
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        out = str(tmp_path / ('fit%d.pt' % i))
        mp.spawn(_fit_worker, args=(2, port, out), nprocs=2, join=True)
        outs.append(torch.load(out)['flat'])
    assert torch.isfinite(outs[0]).all()
    assert not torch.equal(outs[0], outs[1])


def test_shard_covers_all_rows_once():
    ds = TensorDataset({'a': torch.arange(10.)}, torch.arange(10.))
    for world in (1, 2, 3, 4, 7, 8):
        rows = torch.cat([ds.shard(r, world).targets for r in range(world)])
        assert torch.equal(rows, torch.arange(10.))
        sizes = [ds.shard(r, world).cardinality() for r in range(world)]
        assert max(sizes) - min(sizes) <= 1
