"""Base class of the inference algorithms (mirror of bayes_vi/inference/inference.py:27-58)."""


class Inference:
    """An inference algorithm = a Bayesian `Model` + a dataset; subclasses implement `fit`."""

    def __init__(self, model, dataset):
        self.model = model
        self.dataset = dataset
        self.num_datapoints = int(self.dataset.cardinality())

    def fit(self, *args, **kwargs):
        raise NotImplementedError('fit is not yet implemented!')
