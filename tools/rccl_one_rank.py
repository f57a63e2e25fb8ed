"""One-rank rehearsal of the RCCL leg of the row-sharded path on a one-GPU box.

A world of one rank cannot show scaling, but it runs exactly the calls a multi-GPU job makes: the nccl (= RCCL)
process group bound to the device, the all-reduce of the flat [C * (1 + P)] buffer issued between the fused pass
and the state kernels on torch's current stream, the rank-0 broadcast of the initial chains, and a short device-HMC
run through them.  The totals must equal the un-sharded ones bit for bit (a sum over one shard)."""
import os, sys, time
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29531')
os.environ.setdefault('RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
group = dist.group.WORLD

from bayes_vi_amd import _native, distributions as tfd, bijectors as tfb
from bayes_vi_amd.fused_target import LinearRegression, Posterior, replicate_from_rank0

n, d, c = 1_000_000, 64, 1024
g = torch.Generator(device=dev); g.manual_seed(1)
X = torch.randn(n, d, generator=g, device=dev)
w = torch.randn(d, generator=g, device=dev)
y = 0.5 + X @ w + torch.randn(n, generator=g, device=dev)
theta = torch.cat([0.5 + 0.01 * torch.randn(c, 1, generator=g, device=dev),
                   w + 0.01 * torch.randn(c, d, generator=g, device=dev),
                   1 + 0.01 * torch.rand(c, 1, generator=g, device=dev)], -1)

plain = LinearRegression(X, y)
sharded = LinearRegression(X, y, group=group)
sharded.sharded = True                      # a world of one is not sharded by itself: force the collective leg
lp0, g0 = plain.logp_grad(theta)
lp1, g1 = sharded.logp_grad(theta)
assert torch.equal(lp0, lp1) and torch.equal(g0, g1), 'all-reduce over one rank changed the totals'
sharded.fixed_order_reduction = True
lp2, g2 = sharded.logp_grad(theta)
assert torch.equal(lp0, lp2) and torch.equal(g0, g2), 'all-gather over one rank changed the totals'
sharded.fixed_order_reduction = False
z = replicate_from_rank0(theta.clone(), group)
assert torch.equal(z, theta)

# the per-evaluation cost of the collective on the launch stream
torch.cuda.synchronize()
for fn, name in ((plain, 'no collective'), (sharded, 'RCCL all-reduce of %d KB' % (c * (d + 3) * 4 // 1024))):
    for _ in range(3):
        fn.logp_grad(theta)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        fn.logp_grad(theta)
    torch.cuda.synchronize()
    print('%-32s %.3f ms per evaluation (N = %d, C = %d)' % (name, (time.perf_counter() - t0) / 50 * 1e3, n, c), flush=True)

post = Posterior(sharded, [(tfd.Normal(0., 10.), 1), (tfd.Normal(torch.zeros(d, device=dev), 10.), d),
                           (tfd.HalfNormal(10.), 1)], [tfb.Identity(), tfb.Identity(), tfb.Softplus()])
post0 = Posterior(plain, [(tfd.Normal(0., 10.), 1), (tfd.Normal(torch.zeros(d, device=dev), 10.), d),
                          (tfd.HalfNormal(10.), 1)], [tfb.Identity(), tfb.Identity(), tfb.Softplus()])
zz = post.bijector.inverse(theta).contiguous()
a = post.unconstrained_log_prob_and_grad(zz)
b = post0.unconstrained_log_prob_and_grad(zz)
assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print('RCCL one-rank rehearsal OK', flush=True)
