"""A small attribute dictionary with the semantics the reference code expects of
``addict.Dict`` (missing key -> empty node that compares equal to ``{}``), plus the
hyper-parameters of ``configs/hybrid_switch.yml`` for the synthetic driver.

The drop-in classes accept the reference's own ``cfg`` / ``cfg_spec`` objects as well;
this module exists because ``addict`` is not installed here and the bench / tests need a
config without reading the reference checkout.
"""


class Cfg(dict):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return self[name]

    def __setattr__(self, name, value):
        self[name] = value

    def __missing__(self, name):
        node = Cfg()
        object.__setattr__(node, "_parent", (self, name))
        return node

    def __setitem__(self, name, value):
        if isinstance(value, dict) and not isinstance(value, Cfg):
            value = Cfg.from_dict(value)
        super().__setitem__(name, value)
        parent = self.__dict__.pop("_parent", None)
        if parent is not None:
            parent[0][parent[1]] = self

    @classmethod
    def from_dict(cls, d):
        out = cls()
        for k, v in d.items():
            out[k] = v
        return out


def unset(v):
    """True for keys the yml does not define (addict yields {} for them) or None."""
    return v is None or (isinstance(v, dict) and len(v) == 0)


def hybrid_switch_cfg(width=1024, height=512, device="cuda:0", snapshot_dir="/tmp/onda_snapshots", batch_size=4):
    """(cfg, cfg_spec) equivalent to configs/hybrid_switch.yml:25-80 at the BASELINE resolution
    (the yml ships RESOLUTION [128, 64]; [1024, 512] is its commented alternative, :11-12)."""
    cfg = Cfg.from_dict({
        "SCHEME": {"RESOLUTION": [width, height]},
        "MODEL": {"NAME": "DeepLabv2-Resnet50", "MULTI_LEVEL": False, "CLASSIFIER": "ProDA", "LOAD": None,
                  "LR_RATIO": "80:10"},
        "TRAINING": {"REPLAY_BUFFER": 1000, "BUFFER_DYNAMIC": False, "PERC_FILL_PER_DOMAIN": 0.0,
                     "RANDOM_SEED": 123, "BATCH_SIZE": batch_size, "SHUFFLE": True},
        "OTHERS": {"NUM_WORKERS": 7, "SNAPSHOT_DIR": snapshot_dir, "GENERATE_SAMPLES_EVERY": 3,
                   "VALIDATION": "all", "ECE_SKIP": True, "DEVICE": device},
        "NUM_CLASSES": 19,
        "METHOD": {"ADAPTATION": {"NAME": "PROTO_ONLINE_HYBRIDSWITCH"}},
    })
    spec = Cfg.from_dict({
        "EXP_MONITOR_CONST": 0.003, "DEV_MONITOR_FUNC": "hamming", "LEARNING_RATE_D": 1.0e-5,
        "LEARNING_RATE": 1.0e-5, "WEIGHT_DECAY": 0.0001, "MOMENTUM": 0.9, "AVG_MONITOR_SIZE": 200,
        "GRAY_AREA": [0.83, 0.9], "DEV_THRESH": 0.0002, "SOFT_TRANS": True, "BUFF_CE": 1.0, "BUFF_RCE": 0.0,
        "RCE_ALPHA": 0.1, "RCE_BETA": 1, "EMA_UPDATE": 0.999, "EMA_LAMBDA": 0, "STATIC_LAMBDA": 1,
        "DYNAMIC_LAMBDA": 1, "BN_MOMENTUM": 0.01, "MA_LAMBDA": 0.9995, "TAU": 1, "SKIP_CALC": False,
        "DISTANCE_MEASURE": "mahalanobis", "PSEUDO_THRESH": 0.3, "SOURCE_REPEAT": 1, "REGULARIZER_WEIGHT": 0.1,
        "REGULARIZER": "MRKLD", "FORCE_TARGET_COMPUTE": True, "KEEP_PROTO": True, "JS_D": 0, "LOAD_PROTO": None,
        "STARTING_PROTO": "source", "POWER": 0, "EPOCHS": 3, "BN_POLICY": "freeze", "SKIP_PROTO_EVAL": True,
    })
    spec["set_"] = (25,)
    cfg.METHOD.ADAPTATION.PROTO_ONLINE_HYBRIDSWITCH = spec
    return cfg, spec
