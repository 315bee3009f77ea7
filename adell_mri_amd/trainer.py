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

    def reserve_memory(self, main_bytes=None, side_bytes=None):
        """Pre-size the caching allocator's pools after the first steps of a workload: one large
        block is allocated and released again on the training stream and on the weight-gradient
        stream (functional.side_run), so that later steps split cached memory instead of asking
        the driver for more. A fresh device allocation inside a step is cleared by the driver at
        ~65 GB/s on some boxes (4-7 GB of pool growth = a 60-110 ms stall in one step, round 3's
        driver record); with 288 GB of HBM per GPU the head-room is free. Defaults: half of what
        is reserved now (>= 4 GiB) for the main pool, a quarter (>= 2 GiB) for the side pool.
        Returns (main_bytes, side_bytes)."""
        import torch

        from . import functional as HF

        dev = next(self.module.parameters()).device
        if dev.type != "cuda":
            return 0, 0
        reserved = torch.cuda.memory_stats(dev).get("reserved_bytes.all.current", 0)
        free, _ = torch.cuda.mem_get_info(dev)
        if main_bytes is None:
            main_bytes = max(4 << 30, reserved // 2)
        if side_bytes is None:
            side_bytes = max(2 << 30, reserved // 4)
        # never more than half of what the device still has free
        main_bytes = int(min(main_bytes, free // 2))
        side_bytes = int(min(side_bytes, max(0, free // 2 - main_bytes)))
        if main_bytes > 0:
            t = torch.empty(main_bytes, dtype=torch.uint8, device=dev)
            del t
        side = HF._SIDE["stream"]
        if side is not None and side.device == dev and side_bytes > 0:
            with torch.cuda.stream(side):
                t = torch.empty(side_bytes, dtype=torch.uint8, device=dev)
                del t
        else:
            side_bytes = 0
        return main_bytes, side_bytes

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
