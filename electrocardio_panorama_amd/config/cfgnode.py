"""A small work-alike of yacs.config.CfgNode (yacs is not installed here).

Covers what reference codes/config/default.py:1-55 and codes/main.py:22-23 use: attribute access,
nested nodes, `merge_from_file` (YAML), `merge_from_list`, `clone`, `freeze/defrost`.  Like yacs it
rejects unknown keys and type mismatches, and decodes string scalars with `ast.literal_eval`, which is
what turns the YAML string '1e-1' of nef_net.yml:12 into the float 0.1.
"""
import ast
import copy

import yaml


class CfgNode(dict):
    IMMUTABLE = "__immutable__"

    def __init__(self, init_dict=None):
        super().__init__()
        self.__dict__[CfgNode.IMMUTABLE] = False
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode.IMMUTABLE]:
            raise AttributeError(f"Attempted to set {name} to {value}, but CfgNode is immutable")
        self[name] = value

    def __str__(self):
        def fmt(node, indent):
            lines = []
            for k in sorted(node.keys()):
                v = node[k]
                if isinstance(v, CfgNode):
                    lines.append(" " * indent + f"{k}:")
                    lines.extend(fmt(v, indent + 2))
                else:
                    lines.append(" " * indent + f"{k}: {v}")
            return lines
        return "\n".join(fmt(self, 0))

    __repr__ = __str__

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def _set_immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def clone(self):
        return copy.deepcopy(self)

    @staticmethod
    def _decode(v):
        if isinstance(v, dict):
            return CfgNode(v)
        if not isinstance(v, str):
            return v
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v

    @staticmethod
    def _coerce(new, old, key):
        if old is None or type(new) is type(old):
            return new
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        raise ValueError(f"Type mismatch ({type(old)} vs. {type(new)}) for config key: {key}")

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError(f"Non-existent config key: {full}")
            v = self._decode(copy.deepcopy(v))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError(f"Type mismatch for config key: {full}")
                self[k]._merge(v, path + [k])
            else:
                self[k] = self._coerce(v, self[k], full)

    def merge_from_file(self, cfg_filename):
        with open(cfg_filename, "r") as f:
            loaded = yaml.safe_load(f) or {}
        self._merge(loaded, [])

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0
        for full, v in zip(cfg_list[0::2], cfg_list[1::2]):
            node = self
            keys = full.split(".")
            for k in keys[:-1]:
                if k not in node:
                    raise KeyError(f"Non-existent config key: {full}")
                node = node[k]
            if keys[-1] not in node:
                raise KeyError(f"Non-existent config key: {full}")
            node[keys[-1]] = self._coerce(self._decode(v), node[keys[-1]], full)
