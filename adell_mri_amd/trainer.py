"""Minimal optimisation loop for the HIP path when Lightning is not installed:
the part of ``lightning.Trainer.fit`` the reference's training step relies on
(adell_mri/entrypoints/segmentation/train.py:799-819): zero_grad ->
training_step -> backward -> gradient exchange -> optimizer.step."""
import itertools
import os

from .parallel import GradSync

# ADELL_GRAD_COLLECT=0: keep p.grad as views of the flat buffer (autograd accumulates into them)
_SET_TO_NONE = os.environ.get("ADELL_GRAD_COLLECT", "1") != "0"


def _offsets_drawn():
    """The next dropout offset an eager step would draw (functional._dropout_counter, left as found)."""
    from . import functional as HF

    v = next(HF._dropout_counter)
    HF._dropout_counter = itertools.count(v)
    return v


class StepRunner:
    def __init__(self, module, optimizer=None, sync=None):
        self.module = module
        if optimizer is None:
            optimizer = module.configure_optimizers()["optimizer"]
        self.optimizer = optimizer
        self.sync = sync if sync is not None else GradSync(optimizer)
        self.sync.broadcast_parameters(module=module)
        self.step_idx = 0
        self._graph = None

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

    # ---- one captured HIP graph per step ------------------------------------------------------------
    def enable_graph(self, batch, warmup=3):
        """Capture zero_grad -> training_step -> backward -> gradient gather of ONE step in a HIP
        graph (torch.cuda.CUDAGraph) and replay it from then on: the ~800 launches of a UNETR /
        ConvNeXt step cost the host nothing (the reference's loop is host-paced the same way:
        Lightning over eager torch ops, train.py:799-819). What stays outside the graph, eager, is
        what carries host-side scalars: the gradient exchange and ``optimizer.step()`` (learning
        rate, step counts) -- a handful of launches.

        * ``batch`` gives the shapes: its tensors become the static inputs (``train_step`` copies a
          different batch into them);
        * ``warmup`` eager steps run first on the capture stream (allocator pools, packed weights,
          launch plans);
        * dropout: the (seed, offset) words of the kernels are frozen in the graph; the graph's last
          node advances the library's replay counter by the number of offsets a step draws
          (ops.rng_advance), so replay r draws the masks eager step r would have drawn -- losses are
          bit-identical to the eager loop (tests/test_graph_step_gpu.py);
        * data-parallel runs: the bucketed all-reduce is issued from backward hooks, which a replay
          does not run -- refused unless the exchange is the single all-reduce after backward.
        Returns the static loss tensor (updated in place by every replay)."""
        import torch

        from . import functional as HF
        from . import ops

        if self.sync.overlap:
            raise RuntimeError("StepRunner.enable_graph: gradient buckets are sent from backward hooks "
                               "(GradSync(overlap=True)); build GradSync(optimizer, overlap=False)")
        if getattr(self.module, "ema", None) is not None:
            raise RuntimeError("StepRunner.enable_graph: the module's EMA update takes its decay from a "
                               "host-side schedule inside training_step; a replay would freeze it")
        if warmup < 1:
            raise ValueError("StepRunner.enable_graph: at least one warm-up step (it leaves the packed "
                             "weight tables, staging rings and launch plans the capture may not build)")
        import gc
        gc.collect()      # dead modules leave the weight-pack registry NOW, not inside the capture
        dev = next(self.module.parameters()).device
        self._static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        stream = torch.cuda.Stream(dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):
            for _ in range(warmup):
                self._eager_step(self._static)
        torch.cuda.current_stream(dev).wait_stream(stream)
        torch.cuda.synchronize(dev)
        c0 = _offsets_drawn()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            self.optimizer.zero_grad(set_to_none=_SET_TO_NONE)
            loss = self.module.training_step(self._static, self.step_idx)
            loss.backward()
            collect = getattr(self.optimizer, "collect_grads", None)
            if collect is not None:
                collect()
            drawn = _offsets_drawn() - c0                   # offsets one step draws
            if drawn > 0:
                ops.rng_advance(drawn)                      # last node: the next replay's masks
        HF._dropout_counter = itertools.count(c0)           # the capture executed nothing
        self._graph, self._graph_loss, self._graph_draws = graph, loss, drawn
        return loss

    def disable_graph(self):
        """Back to eager steps (the replay counter returns to zero: eager offsets are absolute)."""
        from . import ops

        if getattr(self, "_graph", None) is not None:
            ops.rng_advance(0, set_value=True)
            # the replays advanced the device word by step * draws: the eager counter takes over there
        self._graph = None

    def _eager_step(self, batch):
        self.optimizer.zero_grad(set_to_none=_SET_TO_NONE)
        loss = self.module.training_step(batch, self.step_idx)
        loss.backward()
        self.sync.all_reduce()
        self.optimizer.step()
        self.step_idx += 1
        return loss

    def train_step(self, batch):
        if getattr(self, "_graph", None) is None:
            return self._eager_step(batch)
        import torch

        from . import functional as HF

        for k, v in batch.items():
            st = self._static.get(k)
            if torch.is_tensor(v) and v is not st and v.data_ptr() != st.data_ptr():
                st.copy_(v, non_blocking=True)
        self._graph.replay()
        if self._graph_draws:
            # the host counter stays where an eager loop would be (disable_graph continues there)
            HF._dropout_counter = itertools.count(_offsets_drawn() + self._graph_draws)
        self.sync.all_reduce()
        self.optimizer.step()
        self.step_idx += 1
        return self._graph_loss


def fit_steps(module, batches, optimizer=None):
    runner = StepRunner(module, optimizer)
    module.train()
    return [runner.train_step(b).detach() for b in batches]
