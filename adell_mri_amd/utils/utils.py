"""``loss_factory`` (adell_mri/utils/utils.py:39-59) and ``ExponentialMovingAverage`` (mirror of adell_mri/utils/utils.py:396-522).

Same contract: the first ``update(model)`` deep-copies the model into ``self.shadow``
(eval mode, requires_grad False), later calls apply
``shadow -= (1 - decay) * (shadow - param)`` to every parameter whose name does not
contain "shadow", then advance the linear decay schedule. On the HIP path the update
is ONE kernel over the model's flat parameter buffer (the fused optimisers keep all
trainable parameters in one allocation); when the model is not flat-backed it is one
kernel per parameter.
"""
from collections import OrderedDict
from copy import deepcopy

import torch

from .. import ops
from ..modules.segmentation import losses as _seg_losses


# adell_mri/utils/utils.py:39-59: name -> loss function, per target family (every entry is a HIP
# kernel: cross-entropy / focal / dice and the Tversky / combo / hybrid / unified focal family).
loss_factory = {
    "binary": {
        "cross_entropy": _seg_losses.binary_cross_entropy,
        "focal": _seg_losses.binary_focal_loss,
        "dice": _seg_losses.binary_generalized_dice_loss,
        "tversky_focal": _seg_losses.binary_focal_tversky_loss,
        "combo": _seg_losses.combo_loss,
        "hybrid_focal": _seg_losses.hybrid_focal_loss,
        "unified_focal": _seg_losses.unified_focal_loss,
    },
    "categorical": {
        "cross_entropy": _seg_losses.cat_cross_entropy,
        "focal": _seg_losses.mc_focal_loss,
        "dice": _seg_losses.mc_generalized_dice_loss,
        "tversky_focal": _seg_losses.mc_focal_tversky_loss,
        "combo": _seg_losses.mc_combo_loss,
        "hybrid_focal": _seg_losses.mc_hybrid_focal_loss,
        "unified_focal": _seg_losses.mc_unified_focal_loss,
    },
}


class ExponentialMovingAverage(torch.nn.Module):
    def __init__(self, decay: float, final_decay: float | None = None, n_steps=None):
        super().__init__()
        self.decay = decay
        self.final_decay = final_decay
        self.n_steps = n_steps
        self.shadow = None
        self.step = 0
        self._plan = None
        if self.final_decay is None:
            self.slope = None
            self.intercept = None
        else:
            self.slope = (self.final_decay - self.decay) / self.n_steps
            self.intercept = self.decay

    def set_requires_grad_false(self, model: torch.nn.Module):
        for _, p in model.named_parameters():
            if p.requires_grad is True:
                p.requires_grad = False

    # -- flat fast path ---------------------------------------------------------------
    def _make_plan(self, names, model_params, shadow_params):
        """If every updated model parameter lives in one allocation, lay the shadow out
        with the same offsets so that the whole update is a single launch."""
        order = sorted(names, key=lambda n: model_params[n].data_ptr())
        first = model_params[order[0]]
        storage = first.untyped_storage().data_ptr()
        if any(model_params[n].untyped_storage().data_ptr() != storage for n in order):
            return None
        base = first.data_ptr()
        offs = [(model_params[n].data_ptr() - base) // 4 for n in order]
        total = sum(model_params[n].numel() for n in order)
        span = offs[-1] + model_params[order[-1]].numel()
        if span > total + 4 * len(order) or base % 16 != 0:
            return None  # other tensors are interleaved: not a pure parameter buffer
        end = 0
        for n, o in zip(order, offs):
            if o < end or not model_params[n].is_contiguous():
                return None
            end = o + model_params[n].numel()
        flat = torch.zeros(span, device=first.device, dtype=torch.float32)
        for n, o in zip(order, offs):
            sp = shadow_params[n]
            view = flat[o:o + sp.numel()].view(sp.shape)
            view.copy_(sp.data)
            sp.data = view
        return {"flat": flat, "base": base, "span": span, "first": order[0],
                "names": tuple(names)}

    def _model_flat(self, model_params):
        plan = self._plan
        first = model_params[plan["first"]]
        if first.data_ptr() != plan["base"]:
            return None
        return torch.as_strided(first.data, (plan["span"],), (1,), first.storage_offset())

    @torch.no_grad()
    def update(self, model: torch.nn.Module, exclude_keys: list[str] = None):
        if self.shadow is None:
            # this effectively skips the first epoch
            self.shadow = deepcopy(model)
            self.shadow.training = False
            self.set_requires_grad_false(self.shadow)
            return
        if exclude_keys is None:
            exclude_keys = []
        model_params = OrderedDict(model.named_parameters())
        shadow_params = OrderedDict(self.shadow.named_parameters())
        sd_model_shadow = set(shadow_params.keys()) - set(model_params.keys())
        sd_shadow_model = [x for x in set(model_params.keys()) - set(shadow_params.keys())
                           if "shadow" not in x]
        assert len(sd_model_shadow) == 0
        assert len(sd_shadow_model) == 0
        names = [n for n in model_params if n not in exclude_keys and "shadow" not in n]
        if names:
            if not all(model_params[n].is_cuda for n in names):
                raise RuntimeError("ExponentialMovingAverage (HIP): parameters must be on the GPU")
            if self._plan is None or self._plan["names"] != tuple(names) \
                    or self._model_flat(model_params) is None:
                self._plan = self._make_plan(names, model_params, shadow_params)
            if self._plan is not None:
                ops.ema_update(self._plan["flat"], self._model_flat(model_params), self.decay)
            else:
                for n in names:
                    ops.ema_update(shadow_params[n].data, model_params[n].data.contiguous(),
                                   self.decay)
        if self.final_decay:
            self.decay = self.step * self.slope + self.intercept
        if self.decay > 1.0:
            self.decay = 1.0
        self.step += 1

    def forward(self, *args, **kwargs):
        return self.shadow.forward(*args, **kwargs)

    def state_dict(self, *args, **kwargs) -> dict[str, torch.Tensor]:
        if args or kwargs:  # nested call from a parent module's state_dict()
            return super().state_dict(*args, **kwargs)
        return self.shadow.state_dict()

    def load_state_dict(self, state_dict: dict[str, torch.Tensor]):
        self.shadow.load_state_dict(state_dict)
