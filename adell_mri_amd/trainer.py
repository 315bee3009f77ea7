"""Minimal optimisation loop for the HIP path when Lightning is not installed:
the part of ``lightning.Trainer.fit`` the reference's training step relies on
(adell_mri/entrypoints/segmentation/train.py:799-819): zero_grad ->
training_step -> backward -> gradient exchange -> optimizer.step."""
import os

from .parallel import GradSync

# ADELL_GRAD_COLLECT=0: keep p.grad as views of the flat buffer (autograd accumulates into them)
_SET_TO_NONE = os.environ.get("ADELL_GRAD_COLLECT", "1") != "0"


class StepRunner:
    def __init__(self, module, optimizer=None, sync=None):
        self.module = module
        if optimizer is None:
            optimizer = module.configure_optimizers()["optimizer"]
        self.optimizer = optimizer
        self.sync = sync if sync is not None else GradSync(optimizer)
        self.sync.broadcast_parameters(module=module)
        self.step_idx = 0

    def train_step(self, batch):
        self.optimizer.zero_grad(set_to_none=_SET_TO_NONE)
        loss = self.module.training_step(batch, self.step_idx)
        loss.backward()
        self.sync.all_reduce()
        self.optimizer.step()
        self.step_idx += 1
        return loss


def fit_steps(module, batches, optimizer=None):
    runner = StepRunner(module, optimizer)
    module.train()
    return [runner.train_step(b).detach() for b in batches]
