import os

from .cfgnode import CfgNode  # noqa: F401
from .default import cfg, get_defaults  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))


def resolve_config_path(path):
    """Accept both spellings of the shipped config: `nef_net.yml` (on disk in the reference,
    codes/config/nef_net.yml) and `nef-net.yml` (reference README.md:32)."""
    if os.path.exists(path):
        return path
    alt = os.path.join(os.path.dirname(path), os.path.basename(path).replace("-", "_"))
    if os.path.exists(alt):
        return alt
    pkg = os.path.join(_HERE, os.path.basename(path).replace("-", "_"))
    if os.path.exists(pkg):
        return pkg
    return path
