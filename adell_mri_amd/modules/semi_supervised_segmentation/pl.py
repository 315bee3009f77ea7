"""``UNetContrastiveSemiSL`` (mirror of adell_mri/modules/semi_supervised_segmentation/pl.py:16-592):
the supervised U-Net step plus a local contrastive term between the decoder features of two
un-annotated views -- the student's features of view 1 against the EMA teacher's (or the
stop-gradient student's) linearly transformed features of view 2, weighted by ``ssl_weight`` = 0.01.
Method names, batch layout ({"supervised": ..., "self_supervised": ...}) and arithmetic follow the
reference; Lightning logging and the torchmetrics bookkeeping are not part of the path."""
from typing import Callable

import numpy as np
import torch

from ..segmentation.pl import UNetBasePL, _Base
from .losses import LocalContrastiveLoss
from .unet import UNetSemiSL

OPTIMIZER_EPS_DEFAULT = 1e-8


class UNetContrastiveSemiSL(UNetSemiSL, UNetBasePL):
    def __init__(self, image_key: str = "image", semi_sl_image_key_1: str = "semi_sl_image_1",
                 semi_sl_image_key_2: str = "semi_sl_image_2", label_key: str = "label",
                 skip_conditioning_key: str | None = None,
                 feature_conditioning_key: str | None = None, optimizer_str: str = "sgd",
                 optimizer_eps: float = OPTIMIZER_EPS_DEFAULT, learning_rate: float = 0.001,
                 lr_encoder: float | None = None, start_decay: float | int = 1.0,
                 warmup_steps: float | int = 0, batch_size: int = 4, n_epochs: int = 100,
                 weight_decay: float = 0.005, training_dataloader_call: Callable | None = None,
                 loss_fn: Callable = torch.nn.functional.binary_cross_entropy,
                 loss_params: dict | None = None, loss_fn_semi_sl: Callable = None,
                 ema: torch.nn.Module = None, stop_gradient: bool = True,
                 picai_eval: bool = False, *args, **kwargs) -> torch.nn.Module:
        super().__init__(*args, **kwargs)
        self.image_key = image_key
        self.semi_sl_image_key_1 = semi_sl_image_key_1
        self.semi_sl_image_key_2 = semi_sl_image_key_2
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
        self.loss_params = loss_params
        # the reference defaults to torch's mse_loss, an eager op; the factory always passes
        # LocalContrastiveLoss (network_factories.py:616), which is the default here
        self.loss_fn_semi_sl = LocalContrastiveLoss() if loss_fn_semi_sl is None else loss_fn_semi_sl
        self.ema = ema
        self.stop_gradient = stop_gradient
        self.picai_eval = picai_eval
        self.loss_fn_class = torch.nn.BCEWithLogitsLoss()
        if self.ema is not None:
            self.ema.update(self)
        self.all_pred, self.all_true = [], []
        self.bn_mult = 0.1
        self.ssl_weight = 0.01
        self.semi_supervised = (self.semi_sl_image_key_1 is not None
                                and self.semi_sl_image_key_2 is not None)

    # ---- batches (pl.py:148-198) ---------------------------------------------------------------
    def unpack_batch(self, batch):
        if self.semi_supervised is True and "supervised" in batch:
            batch = batch["supervised"]
        return super().unpack_batch(batch)

    def unpack_batch_semi_sl(self, batch):
        if self.semi_supervised is True:
            batch = batch["self_supervised"]
        x_1 = batch[self.semi_sl_image_key_1]
        x_2 = batch[self.semi_sl_image_key_2]
        x_cond = batch[self.skip_conditioning_key] if self.skip_conditioning_key is not None else None
        x_fc = (batch[self.feature_conditioning_key]
                if self.feature_conditioning_key is not None else None)
        return x_1, x_2, x_cond, x_fc

    def forward_features_ema_stop_grad(self, **kwargs):
        """Teacher features (pl.py:200-218): the EMA shadow when there is one, else this network;
        without gradient when ``stop_gradient``."""
        op = self.ema.shadow.forward_features if self.ema is not None else self.forward_features
        if self.stop_gradient is True:
            with torch.no_grad():
                return op(**kwargs)
        return op(**kwargs)

    def coerce_batch_size(self, *tensors):
        batch_sizes = [x.shape[0] if x is not None else np.inf for x in tensors]
        n = int(min(batch_sizes))
        return [x[:n] if x is not None else None for x in tensors]

    # ---- the contrastive term (pl.py:244-281) --------------------------------------------------
    def step_semi_sl_loco(self, x_1, x_2, x_cond, x_fc, *args, **kwargs):
        features_1 = self.forward_features(X=x_1, X_skip_layer=x_cond,
                                           X_feature_conditioning=x_fc)
        features_2 = self.forward_features_ema_stop_grad(
            X=x_2, X_skip_layer=x_cond, X_feature_conditioning=x_fc,
            apply_linear_transformation=True)
        return self.loss_fn_semi_sl(features_1, features_2, *args, **kwargs).mean() * self.ssl_weight

    def step_semi_sl_anchors(self, x, x_1, x_2, x_cond, x_fc, *args, **kwargs):
        """pl.py:283-337: anchors from the two views, features from the annotated image; needs a
        three-argument loss (LocalContrastiveLossWithAnchors in the reference)."""
        x, x_1, x_2, x_cond, x_fc = self.coerce_batch_size(x, x_1, x_2, x_cond, x_fc)
        with torch.no_grad():
            anchor_1 = (self.forward_features(X=x_1, X_skip_layer=x_cond,
                                              X_feature_conditioning=x_fc)
                        if x_1 is not None else None)
            anchor_2 = (self.forward_features_ema_stop_grad(
                X=x_2, X_skip_layer=x_cond, X_feature_conditioning=x_fc,
                apply_linear_transformation=True) if x_2 is not None else None)
        features = self.forward_features(X=x, X_skip_layer=x_cond, X_feature_conditioning=x_fc)
        return (self.loss_fn_semi_sl(features, anchor_1, anchor_2, *args, **kwargs).mean()
                * self.ssl_weight)

    def step_semi_sl(self, x, x_1, x_2, x_cond, x_fc, *args, **kwargs):
        if x is not None:
            return self.step_semi_sl_anchors(x, x_1, x_2, x_cond, x_fc, *args, **kwargs)
        return self.step_semi_sl_loco(x_1, x_2, x_cond, x_fc, *args, **kwargs)

    # ---- steps (pl.py:371-450, 452-533) --------------------------------------------------------
    def training_step(self, batch, batch_idx):
        y = None
        output_loss = torch.as_tensor(0.0, device=self.device)
        if self.label_key is not None:
            x, x_cond, x_fc, y, y_class = self.unpack_batch(batch)
            pred_final, pred_class, loss, class_loss = self.step(x, y, y_class, x_cond, x_fc)
            output_loss = loss.mean() if class_loss is None else loss.mean() + class_loss
            if _Base is not torch.nn.Module:
                self.log_loss("train_loss", loss, batch_size=y.shape[0])
            self.train_batch_size = x.shape[0]
        if self.semi_sl_image_key_1 is not None and self.semi_sl_image_key_2 is not None:
            x_1, x_2, x_cond, x_fc = self.unpack_batch_semi_sl(batch)
            self_sl_loss = self.step_semi_sl(None, x_1, x_2, x_cond, x_fc)
            self.log("train_self_sl_loss", self_sl_loss,
                     batch_size=y.shape[0] if y is not None else x_1.shape[0], prog_bar=True,
                     sync_dist=True)
            output_loss = output_loss + self_sl_loss
            if self.ema is not None:
                self.ema.update(self)
        return output_loss

    def validation_step(self, batch, batch_idx):
        output_loss = torch.as_tensor(0.0, device=self.device)
        y = None
        if self.label_key is not None:
            x, x_cond, x_fc, y, y_class = self.unpack_batch(batch)
            bs = x.shape[0]
            mbs = self.batch_size if self.train_batch_size is None else self.train_batch_size
            for i in range(0, bs, mbs):
                m, M = i, i + mbs
                _, _, loss, class_loss = self.step(
                    x[m:M], y[m:M], y_class[m:M] if y_class is not None else None,
                    x_cond[m:M] if x_cond is not None else None,
                    x_fc[m:M] if x_cond is not None else None)   # sic: x_cond, as pl.py:485
                output_loss = output_loss + (
                    loss.mean() if class_loss is None else loss.mean() + class_loss) / (bs // mbs)
        if self.semi_sl_image_key_1 is not None and self.semi_sl_image_key_2 is not None:
            x_1, x_2, x_cond, x_fc = self.unpack_batch_semi_sl(batch)
            output_loss = output_loss + self.step_semi_sl(None, x_1, x_2, x_cond, x_fc)
        return output_loss
