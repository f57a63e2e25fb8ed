"""Two ranks sharing one GPU (gloo group, reductions staged through the host) drive the real kernels through the
row-sharded path: shard -> fused partial sums on the device -> one reduction of [C, 1+P] -> device-side assembly,
HMC and NUTS state kernels.  Chains must stay replicated bit for bit across the ranks and agree with the
single-process run on the whole dataset to float32 rounding of the shard sums."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, D, C = 40000, 24, 20


def _make():
    g = torch.Generator().manual_seed(5)
    X = torch.randn(N, D, generator=g); w = torch.randn(D, generator=g)
    y = 0.3 + X @ w + torch.randn(N, generator=g)
    z = torch.cat([0.3 + 0.01 * torch.randn(C, 1, generator=g), w + 0.01 * torch.randn(C, D, generator=g),
                   0.55 + 0.01 * torch.randn(C, 1, generator=g)], -1)
    return X, y, z


def _posterior(X, y, group):
    sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
    from bayes_vi_amd import bijectors as tfb, distributions as tfd
    from bayes_vi_amd.fused_target import LinearRegression, Posterior
    lik = LinearRegression(X.cuda(), y.cuda(), group=group)
    return Posterior(lik, [(tfd.Normal(0., 10.), 1), (tfd.Normal(torch.zeros(D), 10.), D), (tfd.HalfNormal(10.), 1)],
                     [tfb.Identity(), tfb.Identity(), tfb.Softplus()])


def _run(post, z):
    from bayes_vi_amd.inference.mcmc.device_hmc import (BatchedDeviceNoUTurnSamplerKernel,
                                                        DeviceHamiltonianMonteCarloKernel)
    zt = z.cuda()
    lp, g = post.unconstrained_log_prob_and_grad(zt)
    hmc = DeviceHamiltonianMonteCarloKernel(post, step_size=2e-3, num_leapfrog_steps=4, seed=11)
    zs_h, _, _, tr_h = hmc.many_steps(zt, hmc.bootstrap(zt), 4)
    nuts = BatchedDeviceNoUTurnSamplerKernel(post, step_size=2e-3, max_tree_depth=4, seed=12)
    zn, st = zt, nuts.bootstrap(zt)
    leaps = []
    for _ in range(3):
        zn, st, tr = nuts.one_step(zn, st)
        leaps.append(tr['leapfrogs_taken'].cpu())
    return dict(lp=lp.cpu(), g=g.cpu(), zs_h=zs_h.cpu(), acc_h=tr_h['is_accepted'].cpu(), zn=zn.cpu(),
                leaps=torch.stack(leaps))


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
    from bayes_vi_amd.utils.datasets import TensorDataset
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    X, y, z = _make()
    shard = TensorDataset(X, y).shard(rank, world)
    post = _posterior(shard.features, shard.targets, dist.group.WORLD)
    assert post.likelihood.sharded
    res = _run(post, z)
    # the public drivers: ranks that seeded their own generators differently still run the same chains / the same
    # surrogate, because initial states, surrogate parameters and the device's random stream come from rank 0
    import collections
    from bayes_vi_amd import distributions as tfd, symbolic
    from bayes_vi_amd.model import Model
    from bayes_vi_amd.inference.mcmc import MCMC
    from bayes_vi_amd.inference.mcmc.transition_kernels import HamiltonianMonteCarlo
    from bayes_vi_amd.inference.vi import VI
    from bayes_vi_amd.inference.vi.vi import Adam
    from bayes_vi_amd.inference.vi.surrogate_posteriors import ADVI
    priors = collections.OrderedDict(b=tfd.Normal(0., 10.), w=tfd.Normal(torch.zeros(D), 10.),
                                     scale=tfd.HalfNormal(10.))
    model = Model(priors, lambda b, w, scale, features: tfd.Normal(loc=b + symbolic.matvec(features, w), scale=scale))
    ds = TensorDataset(shard.features.cuda(), shard.targets.cuda())
    torch.manual_seed(100 + rank)
    mc = MCMC(model, ds, HamiltonianMonteCarlo(step_size=1e-3, num_leapfrog_steps=3), group=dist.group.WORLD)
    out = mc.fit(num_chains=3, num_samples=4, num_burnin_steps=2, progress_bar=False)
    res['mcmc_w'] = out.samples[1].cpu()
    torch.manual_seed(200 + rank)
    vi = VI(model, ds, ADVI(model, mean_field=True), group=dist.group.WORLD)
    _, losses = vi.fit(Adam(1e-2), num_steps=6, sample_size=8, num_samples_to_approx_marginals=10, progress_bar=False)
    res['vi_losses'] = torch.tensor(losses)
    torch.save(res, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_stay_replicated_and_match_the_unsharded_run(gpu_device, tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=150)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    assert not hung, 'a rank did not finish'
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    r0, r1 = (torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in range(2))
    assert set(r0) == set(r1)
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k           # replicated chains / surrogate: same bits on every rank
    assert bool(torch.isfinite(r0['vi_losses']).all()) and r0['mcmc_w'].shape == (4, 3, D)
    X, y, z = _make()
    ref = _run(_posterior(X, y, None), z)
    for k in ('mcmc_w', 'vi_losses'):
        r0.pop(k)
    np.testing.assert_allclose(r0['lp'].numpy(), ref['lp'].numpy(), rtol=2e-6)
    gs = np.abs(ref['g'].numpy()).max()
    np.testing.assert_allclose(r0['g'].numpy(), ref['g'].numpy(), rtol=1e-4, atol=1e-5 * gs)
    assert torch.equal(r0['acc_h'], ref['acc_h']) and torch.equal(r0['leaps'], ref['leaps'])
    np.testing.assert_allclose(r0['zs_h'].numpy(), ref['zs_h'].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(r0['zn'].numpy(), ref['zn'].numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_bench_multi_rank_path_rehearsal(gpu_device):
    """`python bench.py --gpus 2` started without a launcher, as the round-end driver may start it: bench.py itself
    spawns torch.distributed.run in a child process, the two ranks share this box's one GPU (BVI_BENCH_REHEARSAL=1:
    gloo instead of RCCL, which refuses two ranks on one device), each packs and streams its row shard, every step
    reduces the [C, 1+P] partial sums, and rank 0 prints exactly one result line."""
    import json
    import subprocess
    env = dict(os.environ, BVI_BENCH_REHEARSAL='1', MASTER_ADDR='127.0.0.1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--rows', '400001',
                           '--chains', '256', '--steps', '3', '--warmup', '1', '--no-cpu-baseline'],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=280)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['rccl_world'] == 2 and out['collective_backend'] == 'gloo'
    assert out['finite'] is True and out['allreduce_ms'] is not None and out['allreduce_ms'] > 0
    assert out['steps'] == 3 and out['warmup'] == 1 and out['scaling'] == 'strong'
    assert out['config']['rows'] == 400001 and out['config']['rows_per_gpu'] == 200001      # rank 0 takes the odd row
    assert out['value'] > 0 and out['roofline']['kernel_ms'] > 0
    assert 'cpu_baseline' not in out and 'parity_at_bench_point' not in out                  # rank-0, one-GPU legs only
