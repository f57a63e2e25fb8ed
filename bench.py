#!/usr/bin/env python3
"""Headline benchmark: leapfrog gradient evaluations per second (chains x steps)
of Bayesian linear regression at N = 10M rows, D = 64 features, 1024 chains
(BASELINE.json configs[1]) on MI355X.

A "step" is one leapfrog step of every chain: momentum half-step, position
step, ONE fused pass of the hot path over the whole (row-sharded) dataset --
log-likelihood and gradient for all chains through the C ABI, all-reduced over
the ranks, plus priors and the bijector chain rule -- and the second momentum
half-step.  With --gpus N the rows are split over N ranks (one process per GPU),
chains stay replicated, and every step does one RCCL all-reduce of the
[chains, 1+P] partial sums: strong scaling of a fixed dataset.  Under
torch.distributed.run (WORLD_SIZE set) this process is one of the ranks; started
plainly with --gpus N > 1 it launches torch.distributed.run itself, as a child
process and before anything here touches a GPU, and relays rank 0's line.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak
MFMA_F32_PEAK_TFLOPS = 157.3


def make_shard(n_local, d, rank, device):
    """Synthetic regression shard: X ~ N(0,1), y = 0.5 + X w* + N(0,1); w* is common to all ranks."""
    g = torch.Generator(device=device); g.manual_seed(1000 + rank)
    gw = torch.Generator(device='cpu'); gw.manual_seed(1)
    w_true = torch.randn(d, generator=gw).to(device)
    X = torch.empty(n_local, d, dtype=torch.float32, device=device)
    step = 1 << 20
    for s in range(0, n_local, step):          # chunked so the generator never needs a second 2.5 GB buffer
        X[s:s + step].normal_(generator=g)
    y = 0.5 + X @ w_true + torch.randn(n_local, generator=g, device=device)
    return X, y, w_true


def cpu_baseline(n_total, d, seconds_budget=12.0):
    """Restated TFP-equivalent CPU path (oracle/restated_torch.py) on a bounded sample,
    scaled to the full row count.  Reported only; never a gate."""
    from oracle import restated_torch as rt
    c_s, n_s, chunk = 64, 1_000_000, 50_000
    g = torch.Generator().manual_seed(7)
    X = torch.randn(n_s, d, generator=g); w = torch.randn(d, generator=g)
    y = 0.5 + X @ w + torch.randn(n_s, generator=g)
    theta = torch.cat([0.5 + 0.01 * torch.randn(c_s, 1, generator=g), w + 0.01 * torch.randn(c_s, d, generator=g),
                       1.0 + 0.01 * torch.randn(c_s, 1, generator=g).abs()], -1)
    rt.linreg(X, y, theta, chunk=chunk)        # warm-up
    times = []
    t_start = time.perf_counter()
    while len(times) < 1000 and (time.perf_counter() - t_start) < seconds_budget:
        t0 = time.perf_counter()
        rt.linreg(X, y, theta, chunk=chunk)
        times.append(time.perf_counter() - t0)
    times.sort()
    t = times[len(times) // 2]
    value = c_s / (t * (n_total / n_s))        # chain-steps per second at the full row count
    return {
        'value': value, 'unit': 'grad-evals/s (chains x steps) at N=%d' % n_total,
        'cores': torch.get_num_threads(), 'kind': 'port',
        'sample': ('restated TFP-equivalent torch-CPU fp32 path (materialised [C,N] ops + autograd), '
                   '%d chains x %d rows x D=%d in chunks of %d rows, median of %d runs of %.3f s each (~12 s of CPU work), '
                   'scaled linearly to N=%d rows' % (c_s, n_s, d, chunk, len(times), t, n_total)),
    }


def parity_at_bench_point(X, y, theta, logp, grad, n_chains=64):
    """Observed error of the timed kernel's own outputs on `n_chains` of the bench's chains, against a float64
    evaluation from the sufficient statistics of the augmented design [1, X] (accumulated in float64 on the device,
    outside the timed region): worst |dlogp| / |logp| and worst |dgrad| / max|grad| of the chain."""
    import math
    n, d = X.shape
    dev = X.device
    A = torch.zeros(d + 1, d + 1, dtype=torch.float64, device=dev)
    b = torch.zeros(d + 1, dtype=torch.float64, device=dev)
    yy = torch.zeros((), dtype=torch.float64, device=dev)
    step = 1 << 20
    for s in range(0, n, step):
        Xa = torch.cat([torch.ones(min(step, n - s), 1, dtype=torch.float64, device=dev), X[s:s + step].double()], 1)
        yc = y[s:s + step].double()
        A += Xa.T @ Xa; b += Xa.T @ yc; yy += yc @ yc
    th = theta[:n_chains].double()
    beta, sc = th[:, :d + 1], th[:, d + 1]
    s2 = yy - 2 * beta @ b + ((beta @ A) * beta).sum(-1)
    ref_lp = -0.5 * s2 / sc ** 2 - n * torch.log(sc) - 0.5 * n * math.log(2 * math.pi)
    ref_g = torch.cat([(b[None] - beta @ A) / sc[:, None] ** 2, (-n / sc + s2 / sc ** 3)[:, None]], -1)
    e_lp = ((logp[:n_chains].double() - ref_lp).abs() / ref_lp.abs()).max()
    e_g = ((grad[:n_chains].double() - ref_g).abs().max(-1).values / ref_g.abs().max(-1).values).max()
    return {'chains_checked': int(n_chains), 'reference': 'float64 sufficient statistics of [1, X] on the device',
            'max_rel_err_logp': float(e_lp), 'max_err_grad_over_max_abs_grad': float(e_g)}


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: run the same command under torch.distributed.run in a CHILD
    process (this process has not touched a GPU and never will), relay rank 0's JSON line and the exit code."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_ranks),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{"metric"')]
    for l in proc.stdout.splitlines():
        if not l.startswith('{"metric"'):
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif proc.returncode == 0:
        raise SystemExit('the ranks exited without a result line')
    raise SystemExit(proc.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--rows', type=int, default=10_000_000)
    ap.add_argument('--features', type=int, default=64)
    ap.add_argument('--chains', type=int, default=1024)
    ap.add_argument('--step-size', type=float, default=1e-4)
    ap.add_argument('--precision', default='auto', choices=['auto', 'f32', 'bf16x3', 'bf16x6'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
    args.gpus = world
    # BVI_BENCH_REHEARSAL=1: exercise the multi-rank code path on a box with fewer GPUs than ranks (ranks
    # share devices, collectives go through gloo); the numbers of such a run mean nothing
    rehearsal = os.environ.get('BVI_BENCH_REHEARSAL') == '1'
    device = torch.device('cuda', local_rank % torch.cuda.device_count() if rehearsal else local_rank)
    torch.cuda.set_device(device)
    group = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)
        group = dist.group.WORLD

    def all_reduce(t, op):
        if rehearsal:                      # gloo moves host memory only
            h = t.cpu()
            dist.all_reduce(h, op=op, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op, group=group)

    from bayes_vi_amd import _native, distributions as tfd, bijectors as tfb
    from bayes_vi_amd.fused_target import LinearRegression, Posterior

    n, d, c = args.rows, args.features, args.chains
    n_local = n // world + (1 if rank < n % world else 0)
    X, y, w_true = make_shard(n_local, d, rank, device)
    prec = {'auto': _native.PREC_AUTO, 'f32': _native.PREC_F32, 'bf16x3': _native.PREC_BF16X3,
            'bf16x6': _native.PREC_BF16X6}[args.precision]
    lik = LinearRegression(X, y, group=group, precision=prec)
    post = Posterior(lik,
                     [(tfd.Normal(0., 10.), 1), (tfd.Normal(torch.zeros(d), 10.), d), (tfd.HalfNormal(10.), 1)],
                     [tfb.Identity(), tfb.Identity(), tfb.Softplus()])

    # chains start near the posterior mode (w* + 0.01 N(0,1)): residuals O(1), as in a running sampler
    g0 = torch.Generator(device='cpu'); g0.manual_seed(2)
    theta0 = torch.cat([0.5 + 0.01 * torch.randn(c, 1, generator=g0),
                        w_true.cpu() + 0.01 * torch.randn(c, d, generator=g0),
                        1.0 + 0.01 * torch.randn(c, 1, generator=g0).abs()], -1).to(device)
    z = post.bijector.inverse(theta0).contiguous()
    pu, pc = post.n_unconstrained, post.n_constrained
    blocks = post.device_blocks(); nb = len(blocks)
    lib, ptr = _native.lib(), _native._ptr
    p = _native.philox_normal(3, 0, c, pu, device)          # counter-based momenta, identical on every rank
    eps = args.step_size
    theta = torch.empty(c, pc, dtype=torch.float32, device=device)
    buf = torch.empty(c * (1 + pc), dtype=torch.float32, device=device)
    logp, grad = buf[:c], buf[c:].view(c, pc)
    lp = torch.empty(c, dtype=torch.float32, device=device)
    g = torch.empty(c, pu, dtype=torch.float32, device=device)
    kernel_events, reduce_events = [], []

    def likelihood_pass(timed):
        # HIP events bracket the fused dataset pass on the stream it is launched on (torch's current stream)
        if timed:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        lik._call(theta, logp, grad)
        if timed:
            e1.record(); kernel_events.append((e0, e1))
        if lik.sharded:
            all_reduce(buf, dist.ReduceOp.SUM)
            if timed:      # the collective runs on RCCL's stream; the current stream waits for it, so e1 -> e2
                e2 = torch.cuda.Event(enable_timing=True)      # spans the hand-over, the reduction and the way back
                e2.record(); reduce_events.append((e1, e2))

    # one-time packing of the dataset for the matrix-core kernel (outside `value`): timed here, on the first use
    pack_ms = None
    if lik.matrix_core_format(c) is not None:
        tiny = LinearRegression(X[:4096], y[:4096], precision=prec)       # loads the pack kernels first
        tiny.packed = _native.linreg_pack(tiny.X, tiny.y, fmt=lik.matrix_core_format(c))
        del tiny
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        lik.packed = _native.linreg_pack(lik.X, lik.y, fmt=lik.matrix_core_format(c))
        lik.packed_format = lik.matrix_core_format(c)
        e1.record(); torch.cuda.synchronize()
        pack_ms = e0.elapsed_time(e1)

    st = _native._stream()
    _native.check(lib.bvi_constrain(blocks, nb, c, ptr(z), ptr(theta), st), 'bvi_constrain')
    likelihood_pass(False)
    _native.check(lib.bvi_posterior_assemble(blocks, nb, c, ptr(z), ptr(theta), ptr(logp), ptr(grad), ptr(lp),
                                             ptr(g), st), 'bvi_posterior_assemble')

    # One trajectory of warmup + steps leapfrog steps, as the device sampler runs it: the first half step once
    # (bvi_leapfrog_pre), then per step the fused data pass [+ all-reduce] and bvi_leapfrog_mid -- the second half of
    # this step (priors, Jacobian terms, chain rule, p += eps/2 g) and the first half of the next (z += eps p,
    # theta = f(z)) in one launch.  Every step evaluates the gradient once and advances position and momentum once.
    _native.check(lib.bvi_leapfrog_pre(blocks, nb, c, eps, ptr(z), ptr(p), ptr(g), ptr(theta), st), 'pre')

    def leapfrog_step(timed):
        likelihood_pass(timed)
        _native.check(lib.bvi_leapfrog_mid(blocks, nb, c, eps, ptr(z), ptr(p), ptr(theta), ptr(logp), ptr(grad),
                                           ptr(lp), ptr(g), _native._stream()), 'mid')

    for _ in range(args.warmup):
        leapfrog_step(False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        leapfrog_step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        all_reduce(t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(lp).all().item())

    kern_ms = sum(a.elapsed_time(b) for a, b in kernel_events) / max(len(kernel_events), 1)
    flops = 4.0 * c * n_local * d
    alg_bytes = 4.0 * n_local * (d + 1) + 8.0 * c * (d + 2)
    fmt = lik.matrix_core_format(c)
    uses_mfma16 = fmt is not None
    kernel_name = {None: 'linreg_f32_kernel', _native.PACK_F16: 'linreg_f16_pipe_kernel<64, 8, true>',
                   _native.PACK_BF16X3: 'linreg_bf16x3_kernel<64, %d, true>' % (2 if prec == _native.PREC_BF16X3 else 3)}[fmt]
    peak = MFMA_BF16_PEAK_TFLOPS if uses_mfma16 else MFMA_F32_PEAK_TFLOPS
    achieved = flops / (kern_ms * 1e-3) / 1e12
    # HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/, same
    # command, full 10M-row shard only); null when the run is not that configuration
    traffic = None
    traffic_source = None
    try:
        if uses_mfma16 and n_local == 10_000_000 and c == 1024 and d == 64:
            import glob
            # newest summary first (profiles/r<round>_pmc_traffic.json); the first that measured this kernel wins
            for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
                with open(path) as f:
                    for k, v in json.load(f)['kernels'].items():
                        if kernel_name in k and traffic is None:
                            traffic = v.get('hbm_bytes_per_launch_corrected')
                            traffic_source = os.path.basename(path)
    except (OSError, ValueError, KeyError):
        traffic = None
    out = {
        'metric': 'leapfrog grad-evals/sec (chains x steps) at N=10M, D=64; % of HBM roofline',
        'value': c * args.steps / elapsed,
        'unit': 'grad-evals/s',
        'n_gpus': world, 'rccl_world': dist.get_world_size() if dist.is_initialized() else 1,
        'collective_backend': (dist.get_backend() if dist.is_initialized() else None),
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True,
        'scaling': 'strong',
        'vs_baseline': None,
        'dtype': {None: 'f32', _native.PACK_F16: 'f32 (f16 MFMA on one f16 plane, residuals in 2 pieces; remainders of X and W in exact closed form; f32 accumulate)',
                  _native.PACK_BF16X3: 'f32 (bf16 3-piece split MFMA, f32 accumulate)'}[fmt],
        'data': 'synthetic',
        'config': {'workload': 'Bayesian linear regression, HMC leapfrog gradient evaluations (BASELINE.json configs[1])',
                   'rows': n, 'rows_per_gpu': n_local, 'features': d, 'chains': c, 'precision': args.precision,
                   'parallelism': 'rows sharded over %d GPU(s), chains replicated, all-reduce of [C,1+P] per step' % world},
        'roofline': {
            'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
            'traffic': traffic, 'traffic_source': traffic_source,
            'kernel': kernel_name,
            'kernel_ms': kern_ms, 'algorithmic_flops_per_launch': flops,
            'hbm': {'algorithmic_bytes_per_launch': alg_bytes, 'achieved_GBps': alg_bytes / (kern_ms * 1e-3) / 1e9,
                    'peak_GBps': HBM_PEAK_GBPS, 'frac': alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        },
        'finite': finite,
        'allreduce_ms': (sum(a.elapsed_time(b) for a, b in reduce_events) / len(reduce_events)) if reduce_events else None,
        'pack_ms': pack_ms,
    }
    if rank == 0 and world == 1:
        # what the timed kernel returned at the bench's own operating point, against float64
        th_chk = theta.clone()
        lik._call(th_chk, logp, grad)
        torch.cuda.synchronize()
        out['parity_at_bench_point'] = parity_at_bench_point(X, y, th_chk, logp, grad, min(64, c))
    if rank == 0 and world == 1 and d <= 64:
        # the HBM-bound regime of the same path: 16 chains on the same dataset (reported next to the headline
        # because the metric asks for a fraction of the HBM roof, which only binds at few chains).  `achieved` is
        # the algorithmic bytes of the f32 dataset over the time; the kernel itself reads the f16 plane, half of that
        c16 = 16
        th16 = theta[:c16].contiguous()
        buf16 = torch.empty(c16 * (1 + pc), dtype=torch.float32, device=device)
        lp16, g16 = buf16[:c16], buf16[c16:].view(c16, pc)
        fmt16 = lik.matrix_core_format(c16)
        for _ in range(3):
            lik._call(th16, lp16, g16)
        times = []
        for _ in range(20):
            # four calls queued back to back, as a sampler issues them (a single 0.25 ms call between two events also
            # counts the microseconds the host takes to issue it, during which the GPU waits)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _q in range(4):
                lik._call(th16, lp16, g16)
            e1.record()
            torch.cuda.synchronize(); times.append(e0.elapsed_time(e1) / 4)
        times.sort()
        ms16 = times[len(times) // 2]
        b16 = 4.0 * n_local * (d + 1) + 8.0 * c16 * (d + 2)
        alg = b16 / (ms16 * 1e-3) / 1e9
        hbm = {
            'chains': c16,
            'kernel': ('linreg_f16_kernel<64, 4, 1> + linreg_f16_finalize' if fmt16 == _native.PACK_F16
                       else 'linreg_f32_kernel + linreg_finalize'),
            'bound': 'hbm', 'ms': ms16, 'ms_is': 'per call, four calls queued back to back', 'peak': HBM_PEAK_GBPS,
            'unit': 'GB/s',
            'grad_evals_per_s': c16 / (ms16 * 1e-3),
            # the algorithmic bytes of the f32 dataset, 4 N (D+1), over the time: what the metric's definition counts
            'algorithmic_GBps': alg, 'algorithmic_frac': alg / HBM_PEAK_GBPS}
        if fmt16 == _native.PACK_F16:
            # the kernel itself streams the packed f16 plane + y, half the algorithmic bytes (what the plane leaves
            # out is added in closed form): `achieved` / `frac` are the bytes DRAM actually delivers over the time
            dram = 2.0 * n_local * (32 if d <= 32 else 64) + 4.0 * n_local
            hbm.update({'achieved': dram / (ms16 * 1e-3) / 1e9, 'frac': dram / (ms16 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        'achieved_is': 'bytes of the packed f16 plane + y that the kernel reads, over the time',
                        'dram_bytes_per_launch': dram})
        else:
            hbm.update({'achieved': alg, 'frac': alg / HBM_PEAK_GBPS,
                        'achieved_is': 'algorithmic bytes of the f32 dataset, 4 N (D+1), over the time'})
        out['roofline_hbm_regime'] = hbm
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(n, d)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
