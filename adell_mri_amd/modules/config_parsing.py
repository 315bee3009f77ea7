"""YAML -> constructor keyword arguments for the segmentation / SSL networks of the hot path.

Restates ``parse_config_unet`` (adell_mri/modules/config_parsing.py:30-58) and
``parse_config_ssl`` (:78-136) on this package's classes, so that the reference's sample
configurations (``sample_configs/u-net-3d-resnet.yaml``, ``unetr.yaml``, ``unet-swin.yaml``,
``ssl-2d-convnext.yaml``, ``ssl-resnet.yaml``) can be consumed as they ship.

One tolerated reference defect: the three segmentation sample YAMLs
spell the channel count ``n_channels`` while the parser and every constructor read
``in_channels`` -- as shipped the reference raises ``KeyError``. Here ``n_channels`` is
accepted as an alias of ``in_channels`` (and removed from the returned dictionary, since no
constructor takes it).
"""
import yaml

from ..utils.utils import loss_factory
from .activations import activation_factory
from .layers.adn_fn import get_adn_fn
from .segmentation.losses import CompoundLoss

# constructor arguments of UNet that may appear in a configuration file
unet_args = [
    "spatial_dimensions", "encoding_operations", "conv_type", "link_type", "upscale_type",
    "interpolation", "norm_type", "dropout_type", "padding", "dropout_param", "activation_fn",
    "in_channels", "n_classes", "depth", "kernel_sizes", "strides", "bottleneck_classification",
    "skip_conditioning",
]


def _load(config_file):
    if isinstance(config_file, dict):   # already parsed (tests, bench.py)
        return dict(config_file)
    with open(config_file, "r") as o:
        return yaml.safe_load(o)


def parse_config_unet(config_file, n_keys, n_classes):
    """-> (network_config, loss_keys). ``network_config`` is splatted into a ``*PL`` constructor
    by ``get_segmentation_network``: activation resolved through ``activation_factory``,
    ``loss_fn`` mapping turned into a ``CompoundLoss`` over ``loss_factory[binary|categorical]``,
    ``spatial_dimensions`` / ``batch_size`` defaulted to 3 / 1, ``in_channels`` multiplied by
    the number of image keys."""
    network_config = _load(config_file)
    if "n_channels" in network_config:   # alias, see the module docstring
        alias = network_config.pop("n_channels")
        network_config.setdefault("in_channels", alias)
    if "activation_fn" in network_config:
        network_config["activation_fn"] = activation_factory[network_config["activation_fn"]]
    family = "binary" if n_classes == 2 else "categorical"
    loss_keys, pairs = [], []
    for key, params in network_config["loss_fn"].items():
        pairs.append((loss_factory[family][key], params))
        loss_keys.append(key)
    network_config["loss_fn"] = CompoundLoss(pairs, network_config.get("loss_weights"))
    network_config.setdefault("spatial_dimensions", 3)
    network_config.setdefault("batch_size", 1)
    network_config["in_channels"] = n_keys * network_config["in_channels"]
    return network_config, loss_keys


def parse_config_ssl(config_file, dropout_param, n_keys, is_vit=False):
    """-> (network_config, network_config_correct): ``adn_fn`` factories built from the
    ``norm_fn`` / ``act_fn`` keys for the backbone (spatial) and the heads (1-D), backbone
    ``in_channels`` multiplied by the number of image keys; the second dictionary is the first
    without the ``norm_fn`` / ``act_fn`` keys (what the module constructors accept)."""
    network_config = _load(config_file)
    network_config.setdefault("batch_size", 1)
    backbone_key = "backbone_args" if "backbone_args" in network_config else "encoder_args"
    backbone = network_config[backbone_key]
    def head_adn():
        return get_adn_fn(1, network_config["norm_fn"], network_config["act_fn"],
                          dropout_param=dropout_param)

    if not is_vit:
        backbone["adn_fn"] = get_adn_fn(backbone["spatial_dim"], network_config["norm_fn"],
                                        network_config["act_fn"], dropout_param=dropout_param)
        network_config["projection_head_args"]["adn_fn"] = head_adn()   # required key
    for head in ("prediction_head_args", "projection_head_args"):
        if head in network_config:
            network_config[head]["adn_fn"] = head_adn()
    network_config_correct = {k: v for k, v in network_config.items()
                              if k not in ("norm_fn", "act_fn")}
    backbone["in_channels"] = n_keys if is_vit else n_keys * backbone["in_channels"]
    return network_config, network_config_correct
