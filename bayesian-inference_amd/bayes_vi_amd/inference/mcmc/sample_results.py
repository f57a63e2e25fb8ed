"""Result holder (mirror of bayes_vi/inference/mcmc/sample_results.py:19-34)."""
import torch


class SampleResult:
    """samples: list (one per parameter) of [num_samples, num_chains, *event];
    trace: dict metric -> [num_samples, ...]; accept_ratios: [num_chains]."""

    def __init__(self, samples, trace):
        self.samples = samples
        self.trace = trace
        if self.trace is not None:
            self.accept_ratios = self.trace['is_accepted'].to(torch.float32).mean(0)
