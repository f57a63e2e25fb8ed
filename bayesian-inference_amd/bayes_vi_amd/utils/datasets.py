"""Datasets as device tensors (mirror of bayes_vi/utils/datasets.py:7-57).

The reference builds a `tf.data.Dataset` that the inference drivers immediately
collapse into one full batch (bayes_vi/inference/mcmc/mcmc.py:52).  Here the
dataset IS that full batch: `features` (a dict of [N] columns, one [N, F] tensor,
or a zeros placeholder for generative models) and `targets`, as float32 torch
tensors placed on the GPU once.
"""
import pandas as pd
import torch


class TensorDataset:
    """(features, targets) held as whole tensors."""

    def __init__(self, features, targets):
        self.features, self.targets = features, targets

    def cardinality(self):
        return int(self.targets.shape[0])

    def __len__(self):
        return self.cardinality()

    def full_batch(self):
        """What `list(dataset.batch(N).take(1))[0]` yields in the reference."""
        return self.features, self.targets

    def shard(self, rank, world_size):
        """Contiguous row block of this dataset for one rank (rows are split as evenly as possible)."""
        n = self.cardinality()
        lo = (n * rank) // world_size
        hi = (n * (rank + 1)) // world_size
        f = self.features
        f = {k: v[lo:hi] for k, v in f.items()} if isinstance(f, dict) else f[lo:hi]
        return TensorDataset(f, self.targets[lo:hi])

    def to(self, device):
        f = self.features
        f = {k: v.to(device) for k, v in f.items()} if isinstance(f, dict) else f.to(device)
        return TensorDataset(f, self.targets.to(device))


def _default_device(device):
    if device is not None:
        return torch.device(device)
    return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


def make_dataset_from_df(df, target_names, feature_names=None, format_features_as='dict', device=None):
    """Same arguments and errors as bayes_vi/utils/datasets.py:7-57, plus `device`."""
    if not isinstance(df, pd.DataFrame):
        raise TypeError('df has to be of Type {}'.format(pd.DataFrame))
    if format_features_as not in ['tensor', 'dict']:
        raise ValueError('format_features_as has to be in ["tensor", "dict"]')
    if not isinstance(target_names, list) and target_names != []:
        raise TypeError('target_names has to be a non-empty List[str]')
    dev = _default_device(device)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32).to(dev)
    targets = df[target_names[0]] if len(target_names) == 1 else df[target_names]
    targets = t(targets.values)
    if isinstance(feature_names, list) and feature_names != []:
        if format_features_as == 'tensor':
            features = df[feature_names[0]] if len(feature_names) == 1 else df[feature_names]
            return TensorDataset(t(features.values), targets)
        return TensorDataset({k: t(df[k].values) for k in feature_names}, targets)
    return TensorDataset(torch.zeros_like(targets), targets)
