"""Option structs of the reference's module as make_dataclass exposes them (pixsfm/_pixsfm/src/helpers.h:147-290): constructible
from a dict or from keyword arguments, defaults of the C++ struct for everything left out, attribute access, mergedict() that
refuses unknown fields, summary().  They ARE dicts, which is what the optimizers of pixsfm_amd.api take."""
from copy import deepcopy


class OptionStruct(dict):
    _defaults = {}
    _open = ("solver", "loss")        # nested structs of other libraries (ceres::Solver::Options, LossFunction): any key goes

    def __init__(self, *args, **kwargs):
        dict.__init__(self, deepcopy(self._defaults))
        self.mergedict(dict(*args, **kwargs))

    def mergedict(self, values):
        for key, value in dict(values).items():
            if key not in self:
                raise AttributeError("%s has no attribute %r" % (type(self).__name__, key))
            if isinstance(self[key], dict) and isinstance(value, dict) and key in self._open:
                self[key] = {**self[key], **value}
            else:
                self[key] = value

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError("%s has no attribute %r" % (type(self).__name__, key)) from None

    def __setattr__(self, key, value):
        if key not in self:
            raise AttributeError("%s has no attribute %r" % (type(self).__name__, key))
        self[key] = value

    def summary(self, write_type=False):
        lines = [type(self).__name__ + ":"]
        for key in sorted(self):
            v = self[key]
            if isinstance(v, dict):
                lines.append("    %s:" % key)
                lines += ["        %s = %r" % (k, v[k]) for k in sorted(v)]
            else:
                lines.append("    %s%s = %r" % (key, ": " + type(v).__name__ if write_type else "", v))
        return "\n".join(lines) + "\n"


def struct(name, defaults, doc, base=OptionStruct, extra=None):
    """A named option struct with the given defaults (deep-copied per instance)."""
    body = {"_defaults": dict(defaults), "__doc__": doc}
    body.update(extra or {})
    return type(name, (base,), body)
