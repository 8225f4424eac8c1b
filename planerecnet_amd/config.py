"""Config "plugin" surface of the reference (data/config.py), kept attribute-for-attribute:

  * `Config` is an attribute bag with shallow `.copy(overrides)`, in-place `.replace(dict|Config)` and `.print()`
    (reference data/config.py:42-81);
  * one global `cfg` object that every module captures; `set_cfg(name)` mutates it in place so captures stay
    valid (data/config.py:531-540); `set_dataset(name)` (:543-545);
  * `cfg.backbone.type` is the backbone *class* and `cfg.backbone.args` its positional args -- the plugin hook
    used by `construct_backbone` (models/backbone.py:233-243);
  * the named configs PlaneRecNet_{base,101,50}_config with the reference's values (:407-528).

Only the values are shared with the reference; the tables below are built by small helpers instead of
repeating the literal blocks.
"""
import math  # noqa: F401  (kept for eval() of user config expressions)


class Config(object):
    def __init__(self, entries):
        for k, v in entries.items():
            setattr(self, k, v)

    def copy(self, overrides=None):
        new = Config(vars(self))            # shallow on purpose: nested Config objects stay shared
        for k, v in (overrides or {}).items():
            setattr(new, k, v)
        return new

    def replace(self, other):
        for k, v in (vars(other) if isinstance(other, Config) else other).items():
            setattr(self, k, v)

    def print(self):
        for k, v in vars(self).items():
            print(k, " = ", v)


# BGR ImageNet statistics used by FastBaseTransform (data/config.py:33-34)
MEANS = (103.94, 116.78, 123.68)
STD = (57.38, 57.12, 58.40)
PLANE_CLASSES = ("plane",)
PLANE_LABEL_MAP = {1: 1}
COLORS = ((244, 67, 54), (233, 30, 99), (156, 39, 176), (103, 58, 183), (63, 81, 181), (33, 150, 243), (3, 169, 244),
          (0, 188, 212), (0, 150, 136), (76, 175, 80), (139, 195, 74), (205, 220, 57), (255, 235, 59), (255, 193, 7),
          (255, 152, 0), (255, 87, 34), (121, 85, 72), (158, 158, 158), (96, 125, 139))

# ------------------------------------------------------------------------------------------------ datasets
dataset_base = Config(dict(
    name="PlaneAnnoDataset", train_images="", train_info="", valid_images="", valid_info="", has_gt=True, has_pos=True,
    class_names=PLANE_CLASSES, label_map=PLANE_LABEL_MAP, depth_resolution=None, min_depth=None, max_depth=None,
    scale_factor=None))


scannet_dataset = dataset_base.copy(dict(
    name="ScanNetDataset",
    train_images="./scannet/scans/", train_info="./scannet/scannet_train.json",
    valid_images="./scannet/scans/", valid_info="./scannet/scannet_val.json",
    eval_images="./scannet/scans/", eval_info="./scannet/scannet_eval.json",
    class_names=PLANE_CLASSES, label_map=PLANE_LABEL_MAP,
    depth_resolution=1 / 1000, min_depth=1 / 1000, max_depth=40, scale_factor=1))

nyu_eval = dataset_base.copy(dict(
    name="NYUDataset", eval_images="./NYU/nyu_images/", eval_info="./NYU/nyu_eval.json", scale_factor=1,
    min_depth=1 / 1000, max_depth=40, has_pos=False, depth_resolution=1 / 65535.0 * 9.99547))

S2D3DS_dataset = dataset_base.copy(dict(
    name="S2D3DSDataset",
    train_images="./S2D3DS/images/", train_info="./S2D3DS/s2d3ds_train.json",
    valid_images="./S2D3DS/images_val/", valid_info="./S2D3DS/s2d3ds_val.json",
    depth_resolution=1 / 512, min_depth=1 / 512, max_depth=40, scale_factor=0.5))

data_augment = Config(dict(photometric_distort=True, random_mirror=True, random_flip=True, random_rot90=False,
                           motion_blur=False, gaussian_noise=False))
resnet_transform = Config(dict(channel_order="RGB", normalize=True, subtract_means=False, to_float=False))


# ------------------------------------------------------------------------------------------------ backbones
def _backbones():
    from .backbone import ResNetBackbone      # late import: backbone.py does not import this module at import time
    base = Config(dict(name="Base Backbone", path="path/to/pretrained/weights", type=object, args=tuple(),
                       transform=resnet_transform, selected_layers=list()))
    r101 = base.copy(dict(name="ResNet101", path="resnet101_reducedfc.pth", type=ResNetBackbone, args=([3, 4, 23, 3],),
                          transform=resnet_transform, selected_layers=list(range(3, 7))))
    r101_dcn = r101.copy(dict(name="ResNet101_DCN_Interval3", args=([3, 4, 23, 3], [0, 4, 23, 3], 3)))
    r50 = r101.copy(dict(name="ResNet50", path="resnet50-19c8e357.pth", type=ResNetBackbone, args=([3, 4, 6, 3],),
                         transform=resnet_transform))
    r50_dcn = r50.copy(dict(name="ResNet50_DCNv2", args=([3, 4, 6, 3], [0, 4, 6, 3])))
    return base, r101, r101_dcn, r50, r50_dcn


backbone_base, resnet101_backbone, resnet101_dcn_inter3_backbone, resnet50_backbone, resnet50_dcnv2_backbone = _backbones()

fpn_base = Config(dict(selected_layers=list(range(0, 4)), start_level=None, num_features=256, interpolation_mode="bilinear",
                       high_level_mode=None, relu_pred_layers=True))
depth_fpn = Config(dict(selected_layers=list(range(0, 4)), skip_layers=list(range(0, 4)), use_refle=True))

_NMS = dict(nms_pre=500, score_thr=0.1, nms_type="matrix", mask_thr=0.1, update_thr=0.15, nms_kernel="gaussian", nms_sigma=2,
            top_k=100)
_SOLO_COMMON = dict(use_dcn_in_instance=False, sigma=0.2, use_coord_conv=True, norm="GN", focal_loss_init_pi=0.01, **_NMS)

solov2_base = Config(dict(
    num_kernels=256, masks_in_features=["p2", "p3", "p4", "p5"], masks_channels=128, num_masks=256,
    instance_in_features=["p2", "p3", "p4", "p5", "p6"], instance_channels=512, fpn_instance_strides=[8, 8, 16, 32, 32],
    fpn_scale_ranges=((1, 96), (48, 192), (96, 384), (192, 768), (384, 2048)), num_grids=[40, 36, 24, 16, 12],
    num_instance_convs=4, **_SOLO_COMMON))

solov2_light = Config(dict(
    num_kernels=128, masks_in_features=["p2", "p3", "p4", "p5"], masks_channels=128, num_masks=128,
    instance_in_features=["p2", "p3", "p4", "p5"], instance_channels=256, fpn_instance_strides=[8, 8, 16, 32],
    fpn_scale_ranges=((1, 128), (64, 256), (128, 512), (256, 2048)), num_grids=[40, 36, 24, 16],
    num_instance_convs=3, **_SOLO_COMMON))

# ------------------------------------------------------------------------------------------------ model configs
PlaneRecNet_base_config = Config(dict(
    name="PlaneRecNet_base", dataset=scannet_dataset, num_classes=len(scannet_dataset.class_names) + 1, augment=data_augment,
    max_iter=125000, lr_steps=(62500, 100000), lr=1e-4, momentum=0.9, decay=5e-4, freeze_bn=False,
    lr_warmup_init=1e-6, lr_warmup_until=2000, gamma=0.1, delayed_settings=[],
    backbone=resnet101_backbone.copy(dict(selected_layers=list(range(2, 4)))),
    fpn=fpn_base.copy(dict(start_level=0, high_level_mode="original")),
    depth=depth_fpn, solov2=solov2_base,
    dice_weight=3.0, focal_weight=1.0, depth_weight=5.0, use_lava_loss=False, use_plane_loss=False,
    lava_weight=0.5, pln_weight=1.0, focal_gamma=2.0, focal_alpha=0.25,
    discard_box_width=4 / 640, discard_box_height=4 / 640, max_size=640, device="cuda", preserve_aspect_ratio=False))

PlaneRecNet_101_config = PlaneRecNet_base_config.copy(dict(
    name="PlaneRecNet_101", lr_steps=(62500, 100000),
    backbone=resnet101_dcn_inter3_backbone.copy(dict(selected_layers=list(range(2, 4)))),
    fpn=fpn_base.copy(dict(start_level=0, high_level_mode=None)),
    solov2=solov2_light.copy(dict(instance_in_features=["p2", "p3", "p4", "p5"], num_grids=[40, 36, 24, 16],
                                  fpn_instance_strides=[8, 8, 16, 32])),
    use_lava_loss=True, use_plane_loss=True, lava_weight=1.0, pln_weight=1.0))

PlaneRecNet_50_config = PlaneRecNet_101_config.copy(dict(
    name="PlaneRecNet_50", backbone=resnet50_dcnv2_backbone.copy(dict(selected_layers=list(range(2, 4))))))

cfg = PlaneRecNet_base_config.copy()


def set_cfg(config_name: str):
    """Switch the active config IN PLACE (modules that did `from ...config import cfg` keep seeing it)."""
    cfg.replace(eval(config_name))
    if cfg.name is None:
        cfg.name = config_name.split("_config")[0]


def set_dataset(dataset_name: str):
    cfg.dataset = eval(dataset_name)
