"""SSL backbone -> U-Net encoder hand-off: the caller-side glue of
`adell_mri/entrypoints/segmentation/train.py:672-734` as one function, for encoders pre-trained
with the VICReg / ResNet path (config 2b)."""
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ..modules.layers.res_net import ResNet, resnet_to_encoding_ops


def unet_encoder_from_ssl(network_config: dict, network_config_ssl: dict,
                          encoder_state_dicts: Optional[Sequence[Dict[str, torch.Tensor]]] = None,
                          lr_encoder: Optional[float] = None, n_encoders: int = 1
                          ) -> Tuple[dict, torch.nn.ModuleList, List[ResNet]]:
    """Builds the ResNet backbone(s) described by the SSL configuration, optionally loads the
    Lightning checkpoints' ``["state_dict"]`` (``strict=False``, train.py:691-698), derives the
    U-Net ``depth`` / ``kernel_sizes`` / ``strides`` from the backbone structure (:700-713),
    freezes the encoder when ``lr_encoder == 0.0`` and a checkpoint was loaded (:718-724) and
    repackages stem / stages / pools as ``encoding_operations`` (:726-733).

    Returns (updated copy of network_config, encoding_operations, the ResNet objects)."""
    cfg_ssl = {k: v for k, v in network_config_ssl.items()
               if k not in ("weight_decay", "learning_rate", "batch_size")}
    res_nets = [ResNet(**cfg_ssl) for _ in range(n_encoders)]
    if encoder_state_dicts is not None:
        for net, sd in zip(res_nets, encoder_state_dicts):
            net.load_state_dict(sd, strict=False)
    backbone = res_nets[0].backbone
    cfg = dict(network_config)
    cfg["depth"] = [backbone.structure[0][0], *[x[0] for x in backbone.structure]]
    cfg["kernel_sizes"] = [3 for _ in cfg["depth"]]
    mpl = (cfg_ssl["backbone_args"]["maxpool_structure"] if "backbone_args" in cfg_ssl
           else cfg_ssl["maxpool_structure"])
    cfg["strides"] = [2, *mpl]
    if encoder_state_dicts is not None and lr_encoder == 0.0:
        for net in res_nets:
            bb = net.backbone
            for op in [bb.input_layer, *bb.operations]:
                for p in op.parameters():
                    p.requires_grad = False
    return cfg, resnet_to_encoding_ops(res_nets), res_nets
