"""Training-step wrappers (mirror of adell_mri/modules/segmentation/pl.py:197-763).

The reference wraps its networks as ``class UNetPL(UNet, UNetBasePL)`` with
``UNetBasePL(pl.LightningModule)``. Lightning is optional here: when it is
importable the base is a LightningModule (so ``Trainer.fit`` works unchanged);
otherwise the same methods live on a plain ``torch.nn.Module`` and
``adell_mri_amd.trainer.fit_steps`` drives them. What is kept: method names and
arithmetic of ``step`` / ``training_step`` / ``calculate_loss`` /
``configure_optimizers`` (pl.py:218-222, 284-317, 382-421, 529-595) and the
``UNetPL`` constructor (pl.py:680-700). The optimiser is the fused flat-buffer
HIP one (``adell_mri_amd.optim``) instead of ``torch.optim``.
"""
from typing import Callable

import torch
import torch.nn.functional as F

from ... import functional as HF
from ..learning_rate import CosineAnnealingWithWarmupLR
from .unet import BrUNet, UNet
from .unetpp import UNetPlusPlus
from .unetr import SWINUNet, UNETR

try:  # pragma: no cover - lightning is not installed in the build image
    import lightning.pytorch as pl

    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    _Base = torch.nn.Module


# the factory lives where the reference keeps it (utils/optimizer_factory.py); re-exported here
# because round-2 code imported it from this module
from ...utils.optimizer_factory import (OPTIMIZER_MATCH, get_optimizer,  # noqa: E402,F401
                                        optimizer_eps_from_precision)


class UNetBasePL(_Base):
    def __init__(self):
        super().__init__()
        self.train_batch_size = None
        self.raise_nan_loss = False
        self.make_uniform = False
        self.bottleneck_classification = False
        self.feature_conditioning_key = None
        self.skip_conditioning_key = None

    if _Base is torch.nn.Module:
        current_epoch = 0

        def log(self, *args, **kwargs):  # Lightning's logger hook: nothing to do without it
            return None

        @property
        def device(self):
            return next(self.parameters()).device

    def calculate_loss(self, prediction, y):
        loss = self.loss_fn(prediction, y)
        if isinstance(loss, list):
            loss = torch.stack([loss_value.mean() for loss_value in loss])
        return loss

    def step(self, x, y, y_class, x_cond, x_fc):
        y = torch.round(y)
        output = self.forward(X=x, X_skip_layer=x_cond, X_feature_conditioning=x_fc)
        if self.deep_supervision is False:
            prediction, pred_class = output
            deep_outputs = None
        else:
            prediction, pred_class, deep_outputs = output
        loss = self.calculate_loss(prediction, y)
        if self.deep_supervision is True:
            t = len(deep_outputs)
            additional = torch.zeros_like(loss)
            for i, o in enumerate(deep_outputs):
                S = o.shape[-self.spatial_dimensions:]
                # F.interpolate(y, S, mode=linear family, align_corners=True) > 0 (pl.py:305-309)
                y_small = (HF.resize_linear_aligned(y, S) > 0).float()
                additional = additional + self.calculate_loss(o, y_small).mean() / (2 ** (t - i)) / (t + 1)
            loss = loss + additional
        class_loss = None
        if self.bottleneck_classification is True:
            class_loss = self.loss_fn_class(pred_class, y_class.type_as(pred_class)).mean()
        return prediction, pred_class, loss, class_loss

    def unpack_batch(self, batch):
        x, y = batch[self.image_key], batch[self.label_key]
        x_cond = batch[self.skip_conditioning_key] if self.skip_conditioning_key is not None else None
        y_class = y.flatten(start_dim=1).max(1).values if self.bottleneck_classification else None
        x_fc = batch[self.feature_conditioning_key] if self.feature_conditioning_key is not None else None
        return x, x_cond, x_fc, y, y_class

    def calculate_loss_class(self, prediction, y):
        return self.loss_fn_class(prediction, y.type_as(prediction)).mean()

    def crop_if_necessary(self, y, prediction):
        """Centre-crop the ground truth to the prediction's spatial size when ``make_uniform`` is
        set (pl.py:258-282)."""
        if self.make_uniform is True:
            extra = [a - b for a, b in zip(y.shape[2:], prediction.shape[2:])]
            if any(e > 0 for e in extra):
                window = tuple(slice(e // 2, n - (e - e // 2)) for e, n in zip(extra, y.shape[2:]))
                y = y[(slice(None), slice(None), *window)]
        return y, prediction

    def unpack_batch_prediction(self, batch):
        x_cond = batch[self.skip_conditioning_key] if self.skip_conditioning_key is not None else None
        x_fc = (batch[self.feature_conditioning_key]
                if self.feature_conditioning_key is not None else None)
        return batch[self.image_key], x_cond, x_fc

    def predict_step(self, batch, batch_idx=0, return_only_segmentation=False, *args, **kwargs):
        """Forward pass on a batch or on a single un-batched volume (pl.py:347-373)."""
        x, x_cond, x_fc = self.unpack_batch_prediction(batch)
        single = x.dim() == self.spatial_dimensions + 1
        if single:
            x, x_cond, x_fc = (None if t is None else t.unsqueeze(0) for t in (x, x_cond, x_fc))
        output = self.forward(X=x, X_skip_layer=x_cond, X_feature_conditioning=x_fc, *args,
                              **kwargs)
        if return_only_segmentation is True:
            output = output[0]
        return output[0] if single else output

    def _evaluation_loss(self, batch):
        """What validation_step / test_step return (pl.py:423-524): the step loss over micro-batches
        of the training batch size, averaged. (Metric objects and PI-CAI lists are Lightning /
        torchmetrics bookkeeping outside the path.)"""
        x, x_cond, x_fc, y, y_class = self.unpack_batch(batch)
        total = torch.zeros((), device=x.device, dtype=x.dtype)
        bs = x.shape[0]
        mbs = self.batch_size if self.train_batch_size is None else self.train_batch_size
        for m in range(0, bs, mbs):
            part = slice(m, m + mbs)
            _, _, loss, class_loss = self.step(
                x[part], y[part], None if y_class is None else y_class[part],
                None if x_cond is None else x_cond[part],
                x_fc[part] if x_cond is not None else None)   # sic: keyed on x_cond, pl.py:440
            total = total + (loss.mean() if class_loss is None
                             else loss.mean() + class_loss) / (bs // mbs)
        return total

    def validation_step(self, batch, batch_idx):
        return self._evaluation_loss(batch)

    def test_step(self, batch, batch_idx):
        return self._evaluation_loss(batch)

    def log_loss(self, key, loss, **kwargs):
        for i in range(loss.nelement()):
            self.log(f"{key}_{i}", loss[i], sync_dist=True, prog_bar=True, **kwargs)
        self.log(key, loss.mean(), sync_dist=True, prog_bar=True, **kwargs)

    def training_step(self, batch, batch_idx):
        x, x_cond, x_fc, y, y_class = self.unpack_batch(batch)
        pred_final, pred_class, loss, class_loss = self.step(x, y, y_class, x_cond, x_fc)
        if _Base is not torch.nn.Module:
            self.log_loss("train_loss", loss, batch_size=y.shape[0])
        self.train_batch_size = x.shape[0]
        return loss.mean() if class_loss is None else loss.mean() + class_loss

    def configure_optimizers(self) -> dict:
        encoder_params, rest_of_params = [], []
        for k, p in self.named_parameters():
            (encoder_params if ("encoding" in k or "encoder" in k) else rest_of_params).append(p)
        if self.lr_encoder is None:
            parameters = encoder_params + rest_of_params
        else:
            parameters = [{"params": encoder_params, "lr": self.lr_encoder},
                          {"params": rest_of_params}]
        opt_str = getattr(self, "optimizer_str", "sgd")
        if opt_str == "sgd":
            optimizer_params = {"momentum": 0.99, "nesterov": True}
        else:
            optimizer_params = {"eps": self.optimizer_eps}
        optimizer = get_optimizer(opt_str, parameters, lr=self.learning_rate,
                                  weight_decay=self.weight_decay, **optimizer_params)
        self.cosine_decay = any([
            isinstance(self.start_decay, float) and (self.start_decay < 1.0),
            isinstance(self.start_decay, int) and (self.start_decay < self.n_epochs),
            self.warmup_steps > 0,
        ])
        if self.cosine_decay:
            sched = CosineAnnealingWithWarmupLR(optimizer, T_max=self.n_epochs,
                                                start_decay=self.start_decay,
                                                n_warmup_steps=self.warmup_steps)
            sched.last_epoch = self.current_epoch
            return {"optimizer": optimizer, "lr_scheduler": sched, "monitor": "val_loss"}
        return {"optimizer": optimizer, "monitor": "val_loss"}


class UNetPL(UNet, UNetBasePL):
    """Standard U-Net training wrapper (pl.py:673-763)."""

    def __init__(self, image_key: str = "image", label_key: str = "label",
                 skip_conditioning_key: str = None, feature_conditioning_key: str = None,
                 optimizer_str: str = "sgd", optimizer_eps: float = 1e-8,
                 learning_rate: float = 0.001, lr_encoder: float = None,
                 start_decay: float = 1.0, warmup_steps: int = 0, batch_size: int = 4,
                 n_epochs: int = 100, weight_decay: float = 0.005,
                 training_dataloader_call: Callable = None,
                 loss_fn: Callable = F.binary_cross_entropy, picai_eval: bool = False,
                 *args, **kwargs) -> torch.nn.Module:
        super().__init__(*args, **kwargs)
        self.image_key = image_key
        self.label_key = label_key
        self.skip_conditioning_key = skip_conditioning_key
        self.feature_conditioning_key = feature_conditioning_key
        self.optimizer_str = optimizer_str
        self.optimizer_eps = optimizer_eps
        self.learning_rate = learning_rate
        self.lr_encoder = lr_encoder
        self.start_decay = start_decay
        self.warmup_steps = warmup_steps
        self.batch_size = batch_size
        self.n_epochs = n_epochs
        self.weight_decay = weight_decay
        self.training_dataloader_call = training_dataloader_call
        self.loss_fn = loss_fn
        self.picai_eval = picai_eval
        self.loss_fn_class = torch.nn.BCEWithLogitsLoss()


class BrUNetPL(BrUNet, UNetBasePL):
    """Multi-branch U-Net training wrapper (pl.py:1324-1560): a batch carries one tensor per
    ``image_keys`` entry plus ``<key>_weight`` [B] branch weights; deep-supervision targets are
    resized with nearest-neighbour sampling (pl.py:1437-1445)."""

    def __init__(self, image_keys: str = ["image"], label_key: str = "label",
                 skip_conditioning_key: str = None, feature_conditioning_key: str = None,
                 optimizer_str: str = "sgd", optimizer_eps: float = 1e-8,
                 learning_rate: float = 0.001, lr_encoder: float = None,
                 start_decay: float = 1.0, warmup_steps: int = 0, batch_size: int = 4,
                 n_epochs: int = 100, weight_decay: float = 0.005,
                 training_dataloader_call: Callable = None,
                 loss_fn: Callable = F.binary_cross_entropy, picai_eval: bool = False,
                 *args, **kwargs) -> torch.nn.Module:
        super().__init__(*args, **kwargs)
        self.image_keys = image_keys
        self.label_key = label_key
        self.skip_conditioning_key = skip_conditioning_key
        self.feature_conditioning_key = feature_conditioning_key
        self.optimizer_str = optimizer_str
        self.optimizer_eps = optimizer_eps
        self.learning_rate = learning_rate
        self.lr_encoder = lr_encoder
        self.start_decay = start_decay
        self.warmup_steps = warmup_steps
        self.batch_size = batch_size
        self.n_epochs = n_epochs
        self.weight_decay = weight_decay
        self.training_dataloader_call = training_dataloader_call
        self.loss_fn = loss_fn
        self.picai_eval = picai_eval
        self.loss_fn_class = torch.nn.BCEWithLogitsLoss()
        self.all_pred = []
        self.all_true = []
        self.bn_mult = 0.1

    def step(self, x, x_weights, y, y_class, x_cond, x_fc):
        y = torch.round(y)
        output = self.forward(x, x_weights, X_skip_layer=x_cond, X_feature_conditioning=x_fc)
        if self.deep_supervision is False:
            prediction, pred_class = output
            deep_outputs = None
        else:
            prediction, pred_class, deep_outputs = output
        loss = self.calculate_loss(prediction, y)
        if self.deep_supervision is True:
            t = len(deep_outputs)
            additional = torch.zeros_like(loss)
            y5 = y if y.dim() == 5 else y.unsqueeze(2)
            for i, o in enumerate(deep_outputs):
                S = list(o.shape[-self.spatial_dimensions:])
                y_small = HF.interpolate_nearest(y5, S if y.dim() == 5 else [1] + S)
                y_small = y_small if y.dim() == 5 else y_small.squeeze(2)
                additional = additional + self.calculate_loss(o, y_small).mean() / (2 ** (t - i)) / (t + 1)
            loss = loss + additional
        class_loss = None
        if self.bottleneck_classification is True:
            class_loss = self.loss_fn_class(pred_class, y_class.type_as(pred_class)).mean()
        return prediction, pred_class, loss, class_loss

    def unpack_batch(self, batch):
        x, y = [batch[k] for k in self.image_keys], batch[self.label_key]
        x_weights = [batch[k + "_weight"] for k in self.image_keys]
        x_cond = batch[self.skip_conditioning_key] if self.skip_conditioning_key is not None else None
        y_class = y.flatten(start_dim=1).max(1).values if self.bottleneck_classification else None
        x_fc = batch[self.feature_conditioning_key] if self.feature_conditioning_key is not None else None
        return x, x_weights, y, x_cond, x_fc, y_class

    def training_step(self, batch, batch_idx):
        x, x_weights, y, x_cond, x_fc, y_class = self.unpack_batch(batch)
        pred_final, pred_class, loss, class_loss = self.step(x, x_weights, y, y_class, x_cond, x_fc)
        if _Base is not torch.nn.Module:
            self.log_loss("train_loss", loss, batch_size=y.shape[0])
        self.train_batch_size = y.shape[0]
        return loss.mean() if class_loss is None else loss.mean() + class_loss


def _training_attributes(module, hp: dict):
    """Store the optimisation hyper-parameters the reference's wrappers keep as attributes
    (pl.py:829-853 and its copies at :918-944, :1180-1205) on ``module``."""
    for name in ("image_key", "label_key", "skip_conditioning_key", "feature_conditioning_key",
                 "optimizer_str", "optimizer_eps", "learning_rate", "lr_encoder", "start_decay",
                 "warmup_steps", "batch_size", "n_epochs", "weight_decay",
                 "training_dataloader_call", "loss_fn", "picai_eval"):
        setattr(module, name, hp[name])
    module.loss_fn_class = torch.nn.BCEWithLogitsLoss()
    module.all_pred, module.all_true = [], []   # the reference's AUC / AP accumulators
    module.bn_mult = 0.1


class UNETRPL(UNETR, UNetBasePL):
    """UNETR training wrapper (pl.py:766-851): ``UNETR`` keyword arguments plus the
    optimisation hyper-parameters of ``UNetPL``; ``step`` / ``training_step`` /
    ``configure_optimizers`` come from ``UNetBasePL``."""

    def __init__(self, image_key: str = "image", label_key: str = "label",
                 skip_conditioning_key: str = None, feature_conditioning_key: str = None,
                 optimizer_str: str = "sgd", optimizer_eps: float = 1e-8,
                 learning_rate: float = 0.001, lr_encoder: float = None,
                 start_decay: float = 1.0, warmup_steps: int = 0, batch_size: int = 4,
                 n_epochs: int = 100, weight_decay: float = 0.005,
                 training_dataloader_call: Callable = None,
                 loss_fn: Callable = F.binary_cross_entropy, picai_eval: bool = False,
                 *args, **kwargs) -> torch.nn.Module:
        hp = {k: v for k, v in locals().items() if k not in ("self", "args", "kwargs", "__class__")}
        super().__init__(*args, **kwargs)
        _training_attributes(self, hp)


class SWINUNetPL(SWINUNet, UNetBasePL):
    """SWIN-UNet training wrapper (pl.py:854-941)."""

    def __init__(self, image_key: str = "image", label_key: str = "label",
                 skip_conditioning_key: str = None, feature_conditioning_key: str = None,
                 optimizer_str: str = "sgd", optimizer_eps: float = 1e-8,
                 learning_rate: float = 0.001, lr_encoder: float = None,
                 start_decay: float = 1.0, warmup_steps: int = 0, batch_size: int = 4,
                 n_epochs: int = 100, weight_decay: float = 0.005,
                 training_dataloader_call: Callable = None,
                 loss_fn: Callable = F.binary_cross_entropy, picai_eval: bool = False,
                 *args, **kwargs) -> torch.nn.Module:
        hp = {k: v for k, v in locals().items() if k not in ("self", "args", "kwargs", "__class__")}
        super().__init__(*args, **kwargs)
        _training_attributes(self, hp)


class UNetPlusPlusPL(UNetPlusPlus, UNetBasePL):
    """U-Net++ training wrapper (pl.py:1121-1205). ``deep_supervision`` is forced on after
    construction: ``UNetPlusPlus.forward`` returns ``(pred, bn_out, aux)`` and
    ``UNetBasePL.step`` then weights the auxiliary heads as deep-supervision outputs against
    aligned-corner resized targets (pl.py:1199, 298-316)."""

    def __init__(self, image_key: str = "image", label_key: str = "label",
                 skip_conditioning_key: str = None, feature_conditioning_key: str = None,
                 optimizer_str: str = "sgd", optimizer_eps: float = 1e-8,
                 learning_rate: float = 0.001, lr_encoder: float = None,
                 start_decay: float = 1.0, warmup_steps: int = 0, batch_size: int = 4,
                 n_epochs: int = 100, weight_decay: float = 0.005,
                 training_dataloader_call: Callable = None,
                 loss_fn: Callable = F.binary_cross_entropy, picai_eval: bool = False,
                 *args, **kwargs) -> torch.nn.Module:
        hp = {k: v for k, v in locals().items() if k not in ("self", "args", "kwargs", "__class__")}
        super().__init__(*args, **kwargs)
        _training_attributes(self, hp)
        self.deep_supervision = True
