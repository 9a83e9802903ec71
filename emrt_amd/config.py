"""Configuration surface of the reference, without yacs (not installed on the target image).

Mirrors /root/reference/semantic_segmentation/config.py: the same default tree (:5-219), yaml merge with recursive
`BASE` inheritance (:221-232), `update_config(config, args)` (:234-247, returns a DEFROSTED node) and `get_config()`
(:249-251).  CfgNode reproduces the yacs behaviour this path relies on: attribute access, merge_from_file with
key checking, clone/freeze/defrost, and `literal_eval` of yaml strings such as "(256, 256)".
"""
import ast
import copy
import os

import yaml


class CfgNode(dict):
    _IMMUTABLE = "__immutable__"

    def __init__(self, init=None):
        super().__init__()
        self.__dict__[CfgNode._IMMUTABLE] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode._IMMUTABLE]:
            raise AttributeError("Attempted to set %s to %s, but CfgNode is immutable" % (name, value))
        self[name] = value

    def is_frozen(self):
        return self.__dict__[CfgNode._IMMUTABLE]

    def _set_immutable(self, flag):
        self.__dict__[CfgNode._IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        out.__dict__[CfgNode._IMMUTABLE] = self.__dict__[CfgNode._IMMUTABLE]
        return out

    @staticmethod
    def _decode(v):
        if isinstance(v, str):
            try:
                return ast.literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError("Non-existent config key: %s" % full)
            if isinstance(self[k], CfgNode):
                if v is None:
                    continue
                if not isinstance(v, dict):
                    raise ValueError("Type mismatch for config key %s" % full)
                self[k]._merge(v, path + [k])
            else:
                v = CfgNode._decode(copy.deepcopy(v))
                old = self[k]
                if isinstance(old, tuple) and isinstance(v, list):
                    v = tuple(v)
                elif isinstance(old, list) and isinstance(v, tuple):
                    v = list(v)
                dict.__setitem__(self, k, v)

    def merge_from_file(self, cfg_filename):
        with open(cfg_filename, "r") as f:
            loaded = yaml.safe_load(f) or {}
        self._merge(loaded, [])


_DEFAULTS = {
    "BASE": [""],
    "DATA": {"BATCH_SIZE": 4, "BATCH_SIZE_VAL": 1, "DATASET": "PascalContext",
             "DATA_PATH": "/home/ssd3/wutianyi/datasets/pascal_context", "CROP_SIZE": (480, 480), "NUM_CLASSES": 60,
             "NUM_WORKERS": 0},
    "MODEL": {
        "NAME": "SETR_MLA",
        "ENCODER": {"TYPE": "ViT_MLA", "OUT_INDICES": [5, 11, 17, 23], "MULTI_GRID": False, "MULTI_DILATION": None},
        "DECODER_TYPE": "ViT_MLAHead", "RESUME": None, "PRETRAINED": None, "NUM_CLASSES": 1000, "DROPOUT": 0.0,
        "ATTENTION_DROPOUT": 0.0, "DROP_PATH": 0.1, "OUTPUT_STRIDE": 16, "BACKBONE_SCALE": 1.0,
        "TRANS": {"HYBRID": False, "PATCH_GRID": None, "PATCH_SIZE": None, "HIDDEN_SIZE": 768, "MLP_RATIO": 4,
                  "NUM_HEADS": None, "NUM_LAYERS": None, "QKV_BIAS": True, "WINDOW_SIZE": 7, "IN_CHANNELS": 3,
                  "EMBED_DIM": 96, "STAGE_DEPTHS": [2, 2, 6, 2], "QK_SCALE": None, "APE": False, "PATCH_NORM": True,
                  "KEEP_CLS_TOKEN": False, "NUM_STAGES": 4, "STRIDES": [4, 2, 2, 2], "SR_RATIOS": [8, 4, 2, 1],
                  "SPLIT_SIZES": None, "FOCAL_STAGES": None, "FOCAL_LEVELS": None, "FOCAL_WINDOWS": None,
                  "EXPAND_STAGES": None, "EXPAND_SIZES": None, "USE_CONV_EMBED": True},
        "MLA": {"MLA_CHANNELS": 256, "MLAHEAD_CHANNELS": 128, "AUXIHEAD": False, "MLAHEAD_ALIGN_CORNERS": False},
        "PUP": {"INPUT_CHANNEL": 1024, "NUM_CONV": 4, "NUM_UPSAMPLE_LAYER": 4, "CONV3x3_CONV1x1": True, "ALIGN_CORNERS": False},
        "AUXPUP": {"INPUT_CHANNEL": 1024, "NUM_CONV": 2, "NUM_UPSAMPLE_LAYER": 2, "CONV3x3_CONV1x1": True, "ALIGN_CORNERS": False},
        "UPERHEAD": {"IN_CHANNELS": [96, 192, 384, 768], "CHANNELS": 512, "IN_INDEX": [0, 1, 2, 3], "POOL_SCALES": [1, 2, 3, 6],
                     "DROP_RATIO": 0.1, "ALIGN_CORNERS": False},
        "AUX": {"AUXIHEAD": True, "AUXHEAD_ALIGN_CORNERS": False, "LOSS": True, "AUX_WEIGHT": 0.4},
        "AUXFCN": {"IN_CHANNELS": 384, "UP_RATIO": 16},
        "DPT": {"HIDDEN_FEATURES": [256, 512, 1024, 1024], "FEATURES": 256, "READOUT_PROCESS": "project"},
        "SEGMENTER": {"NUM_LAYERS": 2},
        "SEGFORMER": {"IN_CHANNELS": [32, 64, 160, 256], "CHANNELS": 256, "ALIGN_CORNERS": False},
        "TRANS2SEG": {"EMBED_DIM": 256, "DEPTH": 4, "NUM_HEADS": 8, "MLP_RATIO": 3.0, "HID_DIM": 64},
        "RSDECODER": {"EMBED_DIM": 256, "DEPTH": 4, "NUM_HEADS": 8, "MLP_RATIO": 3.0, "HID_DIM": 64},
        "DEFORMABLE": {"EMBED_DIM": 256, "DEPTH": 4, "NUM_HEADS": 8, "MLP_RATIO": 3.0, "HID_DIM": 64},
    },
    "TRAIN": {
        "LOSS": "MixSoftmaxCrossEntropyLoss", "WEIGHTS": [1, 0.4, 0.4, 0.4, 0.4], "USE_GPU": True, "LAST_EPOCH": 0,
        "BASE_LR": 0.001, "END_LR": 1e-4, "DECODER_LR_COEF": 1.0, "ITERS": 80000, "POWER": 0.9, "DECAY_STEPS": 80000,
        "APEX": False, "IGNORE_INDEX": 255,
        "LR_SCHEDULER": {"NAME": "PolynomialDecay", "WARM_UP_STEPS": 0, "WARM_UP_LR_INIT": 0.0, "MILESTONES": [30, 60, 90],
                         "POWER": 0.9, "GAMMA": 0.1},
        "OPTIMIZER": {"NAME": "SGD", "EPS": 1e-8, "BETAS": (0.9, 0.999), "MOMENTUM": 0.9, "NESTEROV": False, "WEIGHT_DECAY": 0.0,
                      "CENTERTED": False, "RHO": 0.95, "GRAD_CLIP": None},
    },
    "VAL": {"USE_GPU": True, "MULTI_SCALES_VAL": False, "SCALE_RATIOS": [0.5, 0.75, 1.0, 1.25, 1.5, 1.75], "IMAGE_BASE_SIZE": None,
            "KEEP_ORI_SIZE": False, "RESCALE_FROM_ORI": False, "CROP_SIZE": [480, 480], "STRIDE_SIZE": [320, 320],
            "MEAN": [123.675, 116.28, 103.53], "STD": [58.395, 57.12, 57.375]},
    "SAVE_DIR": "./output", "KEEP_CHECKPOINT_MAX": 1, "TAG": "default", "SAVE_FREQ_CHECKPOINT": 2000, "LOGGING_INFO_FREQ": 100,
    "VALIDATE_FREQ": 2000, "SEED": 0, "EVAL": False, "LOCAL_RANK": 0,
}

_C = CfgNode(_DEFAULTS)


def _update_config_from_file(config, cfg_file):
    config.defrost()
    with open(cfg_file, "r") as infile:
        yaml_cfg = yaml.safe_load(infile) or {}
    for cfg in yaml_cfg.setdefault("BASE", [""]):
        if cfg:
            _update_config_from_file(config, os.path.join(os.path.dirname(cfg_file), cfg))
    print("merging config from {}".format(cfg_file))
    config.merge_from_file(cfg_file)
    config.freeze()


def update_config(config, args):
    """Update config from an argparse namespace (`args.cfg` = yaml path; optional `args.pretrained_backbone`)."""
    if getattr(args, "cfg", None):
        _update_config_from_file(config, args.cfg)
    config.defrost()
    if "pretrained_backbone" in vars(args) if not isinstance(args, dict) else "pretrained_backbone" in args:
        config.MODEL.PRETRAINED = args.pretrained_backbone
    return config


def get_config():
    return _C.clone()
