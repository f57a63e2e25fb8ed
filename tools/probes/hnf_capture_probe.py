"""Which part of a Hamiltonian-flow VI step does not survive HIP-graph capture at Pu = 66?  One variant per process:
   python tools/probes/hnf_capture_probe.py <rows> <hidden> <n_flows> <lift:0|1> <samples>"""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'bayesian-inference_amd'))
import bayes_vi_amd.distributions as tfd
from bayes_vi_amd import symbolic
from bayes_vi_amd.model import Model
from bayes_vi_amd.inference.vi import VI
from bayes_vi_amd.inference.vi.surrogate_posteriors import NormalizingFlow
from bayes_vi_amd.inference.vi.flow_bijectors import Chain, HamiltonianFlow, make_scale_fn, make_shift_fn
from bayes_vi_amd.inference.vi.vi import Adam
from bayes_vi_amd.utils.datasets import TensorDataset

n, hidden, n_flows, lift, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
d = int(os.environ.get('D', 64))
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(0)
X = torch.randn(n, d, generator=g, device=dev)
w = torch.randn(d, generator=g, device=dev)
y = 0.5 + X @ w + torch.randn(n, generator=g, device=dev)
priors = collections.OrderedDict(b=tfd.Normal(0., 10.), w=tfd.Normal(torch.zeros(d), 10.), scale=tfd.HalfNormal(10.))
model = Model(priors, lambda b, w, scale, features: tfd.Normal(loc=b + symbolic.matvec(features, w), scale=scale))
nd = model.flat_unconstrained_param_event_ndims
flows = Chain([HamiltonianFlow(event_dims=nd, step_sizes=torch.nn.Parameter(torch.tensor(0.1)), num_integration_steps=2,
                               hidden_layers=[hidden, hidden]) for _ in range(n_flows)])
def make_lift():          # the networks must sit in the lambda's closure for NormalizingFlow to find (and move) them
    shift_fn, scale_fn = make_shift_fn(nd, hidden_layers=[hidden, hidden]), make_scale_fn(nd, hidden_layers=[hidden, hidden])
    return lambda q: tfd.MultivariateNormalDiag(loc=shift_fn(q), scale_diag=scale_fn(q))
kw = {'posterior_lift_distribution': make_lift()} if lift else {}
sur = NormalizingFlow(model, flow_bijector=flows, extra_ndims=nd, **kw)
vi = VI(model, TensorDataset(X, y), sur)
params = [p for p in sur.parameters() if p.requires_grad]
phases = os.environ.get('PHASES', '')
if 'n' in phases:
    eps = torch.randn(S, 2 * nd, generator=g, device=dev)
    sur.base_noise = lambda m, dim: eps
if '1' in phases:
    for p in params:
        p.grad = None
    l = vi.loss(S); l.backward()
    print('phase 1 ok', flush=True)
if '2' in phases:
    for lo, hi in ((0, n // 2 + 3), (n // 2 + 3, n)):
        vs = VI(model, TensorDataset(X[lo:hi], y[lo:hi]), sur)
        for p in params:
            p.grad = None
        l = vs.loss(S); l.backward()
        del vs
    print('phase 2 ok', flush=True)
if '3' in phases:
    p_c = model.flat_param_event_ndims
    prior_only = sur.get_corrected_target_log_prob_fn(lambda sample: vi.posterior.log_prior(sample[:, :p_c], False)[0])
    for p in params:
        p.grad = None
    q, logq = sur.sample_and_log_prob(S)
    l0 = (logq - prior_only(q)).mean(); l0.backward()
    print('phase 3 ok', flush=True)
opt = Adam(learning_rate=1e-3).bind(params)
for _ in range(3):
    vi._step(opt, params, S)
torch.cuda.synchronize()
print('eager ok', flush=True)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    static_loss = vi._step(opt, params, S)
print('capture ok', flush=True)
graph.replay(); torch.cuda.synchronize()
print('replay ok', float(static_loss), flush=True)
