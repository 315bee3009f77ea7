"""Network factories of the hot path: ``get_segmentation_network``
(adell_mri/utils/network_factories.py:493-701) and ``get_ssl_network`` (:705-1028) on this
package's HIP-backed modules. Same argument lists, same keyword plumbing into the ``*PL``
constructors, same ``ValueError`` for an unknown ``net_type``. Members outside the built
path (the MONAI wrappers, the ViT-based SSL methods) raise ``NotImplementedError`` --
there is no eager-torch fallback.
"""
from typing import Any, Callable

import torch

from ..modules.segmentation.pl import (BrUNetPL, SWINUNetPL, UNETRPL, UNetPL,
                                       UNetPlusPlusPL)
from ..modules.self_supervised.pl import SelfSLConvNeXtPL, SelfSLResNetPL, SelfSLUNetPL

OPTIMIZER_EPS_DEFAULT = 1e-8
ALLOWED_NET_TYPES = {
    "segmentation": ["unet", "brunet", "unetpp", "unetr", "monai_unetr", "swin", "monai_swin"],
}


def _first_given(*sizes):
    for size in sizes:
        if size is not None:
            return size
    return None


def get_segmentation_network(
        net_type: str, network_config: dict[str, Any], bottleneck_classification: bool,
        clinical_feature_keys: list[str], all_aux_keys: list[str],
        clinical_feature_params: dict[str, torch.Tensor], clinical_feature_key_net: str,
        aux_key_net: str, max_epochs: int, encoding_operations: list[torch.nn.Module],
        picai_eval: bool, lr_encoder: float, encoder_checkpoint: str,
        res_config_file: str | None, deep_supervision: bool, n_classes: int, keys: list[str],
        optimizer_str: str = "sgd", optimizer_eps: float = OPTIMIZER_EPS_DEFAULT,
        start_decay: float | int = 1.0, warmup_steps: float | int = 0.0,
        train_loader_call: Callable = None, random_crop_size: list[int] = None,
        crop_size: list[int] = None, pad_size: list[int] = None, resize_size: list[int] = None,
        semi_supervised: bool = False, max_steps_optim: int = None, seed: int = 42):
    """``network_config`` is what ``parse_config_unet`` returned; ``encoding_operations`` is the
    list the entrypoint prepares (``[None]`` without an SSL backbone, train.py:672-734)."""
    if net_type not in ALLOWED_NET_TYPES["segmentation"]:
        raise ValueError(f"net_type '{net_type}' not valid, has to be one of "
                         f"{ALLOWED_NET_TYPES['segmentation']}")
    # image size for the transformer members: the first of random crop / crop / pad / resize
    size = _first_given(random_crop_size, crop_size, pad_size, resize_size)
    common = dict(
        training_dataloader_call=train_loader_call, label_key="mask", n_classes=n_classes,
        bottleneck_classification=bottleneck_classification,
        skip_conditioning=len(all_aux_keys), skip_conditioning_key=aux_key_net,
        feature_conditioning=len(clinical_feature_keys),
        feature_conditioning_params=clinical_feature_params,
        feature_conditioning_key=clinical_feature_key_net, n_epochs=max_epochs,
        picai_eval=picai_eval, lr_encoder=lr_encoder, start_decay=start_decay,
        warmup_steps=warmup_steps, optimizer_str=optimizer_str, optimizer_eps=optimizer_eps)

    if net_type == "unet" and semi_supervised is True:   # network_factories.py:602-620
        from ..modules.semi_supervised_segmentation import (LocalContrastiveLoss,
                                                            UNetContrastiveSemiSL)
        from .utils import ExponentialMovingAverage

        ema = ExponentialMovingAverage(decay=0.99, final_decay=1.0, n_steps=max_steps_optim)
        return UNetContrastiveSemiSL(
            encoding_operations=encoding_operations[0], image_key="image",
            semi_sl_image_key_1="semi_sl_image_1", semi_sl_image_key_2="semi_sl_image_2",
            deep_supervision=deep_supervision, ema=ema,
            loss_fn_semi_sl=LocalContrastiveLoss(seed=seed), **common, **network_config)
    if net_type in ("monai_unetr", "monai_swin"):
        raise NotImplementedError(f"{net_type}: the MONAI wrappers are outside the HIP path "
                                  "(SURVEY.md section 2: out of scope)")
    if net_type == "brunet":
        network_config["in_channels"] = network_config["in_channels"] // len(keys)
        net = BrUNetPL(encoders=encoding_operations, image_keys=keys,
                       n_input_branches=len(keys), deep_supervision=deep_supervision,
                       **common, **network_config)
        if encoder_checkpoint is not None and res_config_file is None:
            for encoder, ckpt in zip(net.encoders, encoder_checkpoint):
                encoder.load_state_dict(torch.load(ckpt, weights_only=False)["state_dict"])
        return net
    if net_type == "unetpp":
        return UNetPlusPlusPL(encoding_operations=encoding_operations[0], image_key="image",
                              **common, **network_config)
    if net_type == "unet":
        return UNetPL(encoding_operations=encoding_operations[0], image_key="image",
                      deep_supervision=deep_supervision, **common, **network_config)
    sd = network_config["spatial_dimensions"]
    network_config["image_size"] = size[:sd]
    if net_type == "unetr":
        network_config["patch_size"] = network_config["patch_size"][:sd]
        return UNETRPL(image_key="image", deep_supervision=deep_supervision, **common,
                       **network_config)
    return SWINUNetPL(image_key="image", deep_supervision=deep_supervision, **common,
                      **network_config)


_RESNET_DEFAULTS = dict(
    backbone_args={"spatial_dim": 2, "in_channels": 1,
                   "structure": [(64, 64, 3, 2), (128, 128, 3, 2), (256, 256, 3, 2),
                                 (512, 512, 3, 2)],
                   "maxpool_structure": [2, 2, 2, 2], "adn_fn": torch.nn.Identity,
                   "res_type": "resnet"},
    projection_head_args={"in_channels": 512, "structure": [512, 128],
                          "adn_fn": torch.nn.Identity},
    prediction_head_args={"in_channels": 128, "structure": [512, 128],
                          "adn_fn": torch.nn.Identity})


def get_ssl_network(train_loader_call: Callable, max_epochs: int, max_steps_optim: int,
                    warmup_steps: int, ssl_method: str, ema: torch.nn.Module, net_type: str,
                    network_config: dict[str, Any], stop_gradient: bool,
                    optimizer_eps: float = OPTIMIZER_EPS_DEFAULT):
    """``network_config`` is ``parse_config_ssl``'s second return value. As in the reference,
    simclr / byol / vicreg / vicregl always build the ResNet wrapper from the three ``*_args``
    dictionaries (defaults :754-790); every other method name with ``net_type="convnext"``
    builds ``SelfSLConvNeXtPL`` from the whole configuration (:998-1026)."""
    common = {"training_dataloader_call": train_loader_call, "n_epochs": max_epochs,
              "n_steps": max_steps_optim, "warmup_steps": warmup_steps, "ema": ema,
              "batch_size": network_config.get("batch_size", 32),
              "optimizer_eps": optimizer_eps}
    for key in ("learning_rate", "weight_decay"):
        if key in network_config:
            common[key] = network_config[key]
    if ssl_method in ("simclr", "byol", "vicreg", "vicregl"):
        config = {k: network_config.get(k, v) for k, v in _RESNET_DEFAULTS.items()}
        if ssl_method == "simclr":
            config["prediction_head_args"] = None
        config.update(ssl_method=ssl_method, stop_gradient=stop_gradient,
                      temperature=network_config.get("temperature", 0.1),
                      vic_reg_loss_params=network_config.get("vic_reg_loss_params", {}))
        return SelfSLResNetPL(**{**common, **config})
    if ssl_method in ("ijepa", "mae", "dino", "ibot", "barlow"):
        raise NotImplementedError(f"ssl_method {ssl_method!r} (ViT / Barlow-Twins wrappers) is "
                                  "outside the HIP path (SURVEY.md section 2: out of scope)")
    boilerplate = {"training_dataloader_call": train_loader_call,
                   "aug_image_key_1": "augmented_image_1",
                   "aug_image_key_2": "augmented_image_2", "box_key_1": "box_1",
                   "box_key_2": "box_2", "n_epochs": max_epochs, "n_steps": max_steps_optim,
                   "warmup_steps": warmup_steps, "ssl_method": ssl_method, "ema": ema,
                   "stop_gradient": stop_gradient, "temperature": 0.1,
                   "optimizer_eps": optimizer_eps}
    if net_type == "unet_encoder":
        return SelfSLUNetPL(**boilerplate, **network_config)
    if net_type == "convnext":
        network_config["backbone_args"] = {k: v for k, v in network_config["backbone_args"].items()
                                           if k != "res_type"}
        return SelfSLConvNeXtPL(**boilerplate, **network_config)
    return SelfSLResNetPL(**boilerplate, **network_config)
