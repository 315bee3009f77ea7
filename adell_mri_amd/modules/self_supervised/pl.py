"""Self-supervised training-step wrappers (mirror of
adell_mri/modules/self_supervised/pl.py:170-267 ``SelfSLBasePL`` and :759-985
``SelfSLConvNeXtPL``), VICReg / SimSiam / BYOL on the ConvNeXt backbone.

Kept: constructor arguments, ``step`` arithmetic (which head feeds which loss argument,
stop-gradient, EMA forward and update, loss symmetrisation), ``configure_optimizers``
(AdamW over decay + no-decay parameters in one group, cosine schedule with warm-up).
Lightning is optional as in ``segmentation/pl.py``. SimCLR (NT-Xent) and VICRegL have no
HIP loss kernel yet and raise at construction.
"""
import warnings
from typing import Callable

import torch

from ...optim import FusedAdamW
from ..layers.conv_next import ConvNeXt
from ..layers.res_net import ResNet
from ..segmentation.unet import UNet
from ..learning_rate import CosineAnnealingWithWarmupLR
from .losses import NTXentLoss, VICRegLoss, byol_loss, simsiam_loss

try:  # pragma: no cover - lightning is not installed in the build image
    import lightning.pytorch as pl

    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    _Base = torch.nn.Module

OPTIMIZER_EPS_DEFAULT = 1e-8


class SelfSLBasePL(_Base):
    def __init__(self):
        super().__init__()
        self.optimizer_eps = OPTIMIZER_EPS_DEFAULT

    if _Base is torch.nn.Module:
        def log(self, *args, **kwargs):
            return None

        def save_hyperparameters(self, *args, **kwargs):
            return None

    def update_metrics(self, y1, y2, metrics, log=True):
        for k in metrics:
            metrics[k].update(y1.flatten(), y2.flatten())

    def init_loss(self):
        """pl.py:202-212: SimSiam by default, BYOL with an EMA target, else by ``ssl_method``."""
        self.loss = simsiam_loss
        if getattr(self, "ema", None) is not None:
            self.loss = byol_loss
        if self.ssl_method == "vicreg":
            self.loss = VICRegLoss(**self.vic_reg_loss_params)
        if self.ssl_method == "vicregl":
            raise NotImplementedError("ssl_method='vicregl' (VICRegLocalLoss) has no HIP kernel")
        if self.ssl_method == "simclr":
            self.loss = NTXentLoss(temperature=self.temperature)

    def calculate_loss(self, y1, y2, *args):
        if self.stop_gradient is False:
            return self.loss(y1, y2, *args)
        return self.loss(y1, y2.detach(), *args)

    def safe_sum(self, X):
        if isinstance(X, torch.Tensor):
            return X.sum()
        return sum(X)

    def configure_optimizers(self) -> dict:
        if self.n_steps is not None:
            interval, n = "step", self.n_steps
        else:
            interval, n = "epoch", self.n_epochs
        params_no_decay, params_decay = [], []
        for k, p in self.named_parameters():
            (params_no_decay if "normalization" in k else params_decay).append(p)
        optimizer = FusedAdamW(params_decay + params_no_decay, lr=self.learning_rate,
                               weight_decay=self.weight_decay, eps=self.optimizer_eps)
        sched = CosineAnnealingWithWarmupLR(optimizer, T_max=n, start_decay=self.start_decay,
                                            n_warmup_steps=self.warmup_steps, eta_min=0.0)
        return {"optimizer": optimizer,
                "lr_scheduler": {"scheduler": sched, "interval": interval, "frequency": 1},
                "monitor": "val_loss"}

    def setup_metrics(self):
        self.train_metrics = torch.nn.ModuleDict({})
        self.val_metrics = torch.nn.ModuleDict({})
        self.test_metrics = torch.nn.ModuleDict({})


_SSL_HYPERPARAMETERS = (
    "aug_image_key_1", "aug_image_key_2", "box_key_1", "box_key_2", "learning_rate", "batch_size",
    "weight_decay", "training_dataloader_call", "n_epochs", "n_steps", "warmup_steps",
    "start_decay", "ssl_method", "temperature", "vic_reg_loss_params", "stop_gradient",
    "channels_to_batch")


class _TwoViewSSL:
    """What ``SelfSLResNetPL`` (pl.py:312-535) and ``SelfSLConvNeXtPL`` (:759-985) share: two
    augmented views through one network with three heads (``ret`` = representation /
    projection / prediction), an optional EMA or stop-gradient target branch, and a loss that
    returns a list of terms. The backbone class comes first in the MRO of the concrete class."""

    def _init_two_view(self, hp, args, kwargs):
        for k in _SSL_HYPERPARAMETERS:   # plain attributes: set before Module.__init__, as the
            object.__setattr__(self, k, hp[k])   # reference does
        if hp["channels_to_batch"] is True:
            if self._has_heads:
                kwargs["backbone_args"]["in_channels"] = 1
            else:
                kwargs["in_channels"] = 1
        if not self._has_heads:
            kwargs["encoder_only"] = True     # the U-Net encoder alone: forward(x) -> bottleneck
        super().__init__(*args, **kwargs)
        self.optimizer_eps = hp["optimizer_eps"]
        self.ema = hp["ema"]
        if self.ssl_method not in ["vicreg", "vicregl", "simclr"] and self.stop_gradient is False:
            warnings.warn("stop_gradient=False should not (in theory) be used with "
                          "vic_reg=False, vic_reg_local=False or simclr=False")
        self.init_loss()
        if self.ema is not None:
            self.ema.update(self)
        self.loss_str_dict = {"standard": [None], "vicreg": ["inv", "var", "cov"],
                              "vicregl": ["inv", "var", "cov", "local"]}
        self.save_hyperparameters()
        self.setup_metrics()

    _has_heads = True   # forward(x, ret=...) selects representation / projection / prediction

    def _view(self, x, ret):
        return self.forward(x, ret=ret) if self._has_heads else self.forward(x)

    def forward_ema_stop_grad(self, x, ret=None):
        op = self.ema.shadow.forward if self.ema is not None else self.forward
        args = (x, ret) if self._has_heads else (x,)
        if self.stop_gradient is True:
            with torch.no_grad():
                return op(*args)
        return op(*args)

    def _views_share_a_pass(self, ret_1, ret_2):
        """Both views can go through the network as ONE batch of 2B items: same network for both
        (no EMA target, no stop-gradient), heads that nest (projection inside prediction), and no
        layer that couples the items of a batch in training (batch norm). Per item the arithmetic
        is the one of two separate passes; the launches -- this step is bound by the host -- and
        the gradient accumulations of the second pass are halved."""
        if self.ema is not None or self.stop_gradient is True or not self._has_heads:
            return False
        if (ret_1, ret_2) not in (("prediction", "projection"), ("projection", "projection")):
            return False
        if getattr(self, "_batch_coupled", None) is None:
            self._batch_coupled = any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm)
                                      for m in self.modules())
        return not self._batch_coupled

    def _both_views(self, x1, x2, ret_1):
        z = self.forward(torch.cat([x1, x2], 0), ret="projection")
        z1, z2 = z[:x1.shape[0]], z[x1.shape[0]:]
        return (self.prediction_head(z1) if ret_1 == "prediction" else z1), z2

    def _heads_for_method(self, batch):
        """(head of view 1, head of view 2, extra loss arguments), pl.py:457-470."""
        if self.ssl_method == "simclr":
            return "projection", "projection", []
        if self.ssl_method == "vicregl":
            return "representation", "representation", [batch[self.box_key_1],
                                                        batch[self.box_key_2]]
        return "prediction", "projection", []

    def step(self, batch, loss_str: str, metrics: dict, train=False):
        ret_1, ret_2, other_args = self._heads_for_method(batch)
        x1, x2 = batch[self.aug_image_key_1], batch[self.aug_image_key_2]
        if self.channels_to_batch is True:
            x1 = x1.reshape(-1, 1, *x1.shape[2:])
            x2 = x2.reshape(-1, 1, *x2.shape[2:])
        if x1.shape == x2.shape and self._views_share_a_pass(ret_1, ret_2):
            y1, y2 = self._both_views(x1, x2, ret_1)
        else:
            y1 = self._view(x1, ret_1)
            y2 = self.forward_ema_stop_grad(x2, ret=ret_2)
        losses = self.calculate_loss(y1, y2, *other_args)
        self.update_metrics(y1, y2, metrics)
        symmetric_already = self.ssl_method in ("vicreg", "vicregl", "simclr")
        if not symmetric_already:   # SimSiam / BYOL: add the loss with the two views swapped
            y1_ = self.forward_ema_stop_grad(x1, ret=ret_1)
            y2_ = self._view(x2, ret_2)
            losses = losses + self.calculate_loss(y2_, y1_, *other_args)
            self.update_metrics(y2_, y1_, metrics)
        if self.ema is not None and train is True:
            self.ema.update(self)
        loss = self.safe_sum(losses)
        log_kw = dict(batch_size=x1.shape[0], on_epoch=True, on_step=False, prog_bar=True,
                      sync_dist=True)
        self.log(loss_str, loss, **log_kw)
        if self.ssl_method in ("vicregl", "vicreg"):
            for s, loss_value in zip(self.loss_str_dict[self.ssl_method], losses):
                self.log("{}:{}".format(loss_str, s), loss_value, **log_kw)
        self.last_losses = losses
        return loss

    def training_step(self, batch, batch_idx):
        return self.step(batch, "loss", self.train_metrics, train=True)

    def validation_step(self, batch, batch_idx):
        return self.step(batch, "val_loss", self.val_metrics)

    def test_step(self, batch, batch_idx):
        return self.step(batch, "test_loss", self.test_metrics)


class SelfSLConvNeXtPL(_TwoViewSSL, ConvNeXt, SelfSLBasePL):
    """ConvNeXt backbone (pl.py:759-985)."""

    def __init__(self, aug_image_key_1: str = "aug_image_1", aug_image_key_2: str = "aug_image_2",
                 box_key_1: str = "box_1", box_key_2: str = "box_2", learning_rate: float = 0.001,
                 batch_size: int = 4, weight_decay: float = 0.005,
                 training_dataloader_call: Callable = None, n_epochs: int = 1000,
                 n_steps: int = None, warmup_steps: int = 0, start_decay: int = None,
                 ema: torch.nn.Module = None, ssl_method: str = "simclr",
                 temperature: float = 1.0, vic_reg_loss_params: dict = {},
                 stop_gradient: bool = True, channels_to_batch: bool = False,
                 optimizer_eps: float = OPTIMIZER_EPS_DEFAULT, *args, **kwargs):
        hp = {k: v for k, v in locals().items() if k not in ("self", "args", "kwargs", "__class__")}
        self._init_two_view(hp, args, kwargs)


class SelfSLResNetPL(_TwoViewSSL, ResNet, SelfSLBasePL):
    """ResNet backbone (pl.py:312-535): what ``get_ssl_network`` builds for simclr / byol /
    vicreg / vicregl, transferable to a U-Net encoder (utils/handoff.py)."""

    def __init__(self, aug_image_key_1: str = "aug_image_1", aug_image_key_2: str = "aug_image_2",
                 box_key_1: str = "box_1", box_key_2: str = "box_2", learning_rate: float = 0.001,
                 batch_size: int = 4, weight_decay: float = 0.005,
                 training_dataloader_call: Callable = None, n_epochs: int = 1000,
                 n_steps: int = None, warmup_steps: int = 0, start_decay: int = None,
                 ema: torch.nn.Module = None, ssl_method: str = "simclr",
                 temperature: float = 1.0, vic_reg_loss_params: dict = {},
                 stop_gradient: bool = True, channels_to_batch: bool = False,
                 optimizer_eps: float = OPTIMIZER_EPS_DEFAULT, *args, **kwargs):
        hp = {k: v for k, v in locals().items() if k not in ("self", "args", "kwargs", "__class__")}
        self._init_two_view(hp, args, kwargs)


class SelfSLUNetPL(_TwoViewSSL, UNet, SelfSLBasePL):
    """U-Net encoder as the self-supervised backbone (pl.py:538-756): ``UNet(encoder_only=True)``
    returns the bottleneck feature map, the VICReg loss takes its spatial means
    (``VICRegLoss.flatten_if_necessary``); the trained encoder transfers to a segmentation U-Net
    by ``state_dict`` (same ``encoding_operations.*`` keys)."""

    _has_heads = False

    def __init__(self, aug_image_key_1: str = "aug_image_1", aug_image_key_2: str = "aug_image_2",
                 box_key_1: str = "box_1", box_key_2: str = "box_2", learning_rate: float = 0.001,
                 batch_size: int = 4, weight_decay: float = 0.005,
                 training_dataloader_call: Callable = None, n_epochs: int = 1000,
                 n_steps: int = None, warmup_steps: int = 0, start_decay: int = None,
                 ema: torch.nn.Module = None, ssl_method: str = "simclr",
                 temperature: float = 1.0, vic_reg_loss_params: dict = {},
                 stop_gradient: bool = True, channels_to_batch: bool = False,
                 optimizer_eps: float = OPTIMIZER_EPS_DEFAULT, *args, **kwargs):
        hp = {k: v for k, v in locals().items() if k not in ("self", "args", "kwargs", "__class__")}
        self._init_two_view(hp, args, kwargs)
