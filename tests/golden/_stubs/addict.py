"""Minimal stand-in for the `addict` package (absent from this image), written for
the golden-vector generator only.  Behaviour needed by the reference
(config_ouda.py:24-79 and the `cfg_spec.X == {}` idiom): attribute access maps
to item access, a missing key yields an empty child Dict that compares equal to
`{}` and attaches itself to its parent on first assignment."""


class Dict(dict):
    def __init__(self, *args, **kwargs):
        object.__setattr__(self, "_parent", kwargs.pop("__parent", None))
        object.__setattr__(self, "_key", kwargs.pop("__key", None))
        super().__init__()
        for a in args:
            if not a:
                continue
            for k, v in (a.items() if isinstance(a, dict) else a):
                self[k] = self._wrap(v)
        for k, v in kwargs.items():
            self[k] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, Dict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(i) for i in v)
        return v

    def __getattr__(self, name):
        return self[name]

    def __setattr__(self, name, value):
        self[name] = value

    def __missing__(self, name):
        return Dict(__parent=self, __key=name)

    def __setitem__(self, name, value):
        super().__setitem__(name, value)
        p = object.__getattribute__(self, "_parent")
        k = object.__getattribute__(self, "_key")
        if p is not None:
            p[k] = self
            object.__setattr__(self, "_parent", None)
            object.__setattr__(self, "_key", None)

    def __deepcopy__(self, memo):
        import copy
        out = Dict()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out
