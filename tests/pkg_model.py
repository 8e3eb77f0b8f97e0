"""A stand-in model class for the torch.package checkpoint test (interned into the archive by the exporter): parameter names
and constructor attributes of the reference's VampNet, no arithmetic."""
import torch


class VampNet(torch.nn.Module):
    def __init__(self, state_dict: dict, **kwargs):
        super().__init__()
        for k, v in kwargs.items():
            setattr(self, k, v)
        self._names = list(state_dict)
        for i, (k, v) in enumerate(state_dict.items()):
            self.register_parameter(f"p{i}", torch.nn.Parameter(v.clone(), requires_grad=False))

    def state_dict(self, *a, **k):
        return {n: getattr(self, f"p{i}").detach() for i, n in enumerate(self._names)}
