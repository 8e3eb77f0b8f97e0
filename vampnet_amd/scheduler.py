"""`vampnet_amd.scheduler.NoamScheduler` — the reference's learning-rate schedule object (vampnet/scheduler.py:6-50) for
callers that drive an optimiser themselves; `Trainer` evaluates the same formula on the host (`train.noam_lr`) and passes the
rate into `vn_train_update`.  `optimizer` is anything with `param_groups` (a torch optimiser, or the Trainer)."""
from .train import noam_lr


class NoamScheduler:
    def __init__(self, optimizer, d_model: int = 512, factor: float = 1.0, warmup: int = 4000):
        self.warmup = warmup
        self.factor = factor
        self.d_model = d_model
        self.lr = None
        self.steps = 0
        self.optimizer = optimizer

    def state_dict(self):
        return {key: value for key, value in self.__dict__.items() if key != "optimizer"}

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)

    def step(self):
        self.steps += 1
        self.lr = noam_lr(self.steps, self.d_model, self.factor, self.warmup)
        for p in self.optimizer.param_groups:
            p["lr"] = self.lr
