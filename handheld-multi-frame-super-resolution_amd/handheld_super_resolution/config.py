"""Configuration objects for the burst super-resolution hot path.

The reference drives everything from an OmegaConf ``DictConfig`` built from
``configs/default.yaml`` (reference run_handheld.py:94-116) and *mutates it in
place* to carry derived values (reference super_resolution.py:239-242, 274,
280-296; params.py:69-93).  ``omegaconf`` is not installed in this image, so
this module provides a small attribute-dict with the subset of the OmegaConf
API the hot path touches (attribute + item access, ``get``, ``update``,
``OmegaConf.create/load/merge/to_container``).  A real ``DictConfig`` works
too: the hot path only uses attribute access, ``.get`` and ``.update``.

Key names and defaults follow the reference schema; the
defaults live in ``DEFAULTS`` below rather than in a YAML file.
"""
from __future__ import annotations

import copy
from typing import Any, Mapping

__all__ = ["Config", "OmegaConf", "DEFAULTS", "default_config"]


class Config(dict):
    """dict with attribute access; nested mappings are wrapped recursively."""

    def __init__(self, data: Mapping | None = None, **kw):
        super().__init__()
        if data:
            for k, v in dict(data).items():
                self[k] = v
        for k, v in kw.items():
            self[k] = v

    @staticmethod
    def _wrap(v: Any) -> Any:
        if isinstance(v, Config):
            return v
        if isinstance(v, Mapping):
            return Config(v)
        if isinstance(v, tuple):
            return [Config._wrap(x) for x in v]
        if isinstance(v, list):
            return [Config._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config._wrap(v))
        object.__setattr__(self, "_ver", self.version() + 1)

    def __delitem__(self, k):
        super().__delitem__(k)
        object.__setattr__(self, "_ver", self.version() + 1)

    # every mutating dict method goes through the edit counter (a graph replay must notice ANY in-place edit)
    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]

    def pop(self, k, *default):
        if k in self:
            v = self[k]
            del self[k]
            return v
        if default:
            return default[0]
        raise KeyError(k)

    def popitem(self):
        k = next(reversed(self))
        return k, self.pop(k)

    def clear(self):
        for k in list(self):
            del self[k]

    def version(self):
        """Number of in-place edits of THIS mapping (not of nested ones): lets a long-lived consumer — the HIP-graph
        replay of distributed.HipEngine — notice that a configuration was mutated, as the reference's process() does."""
        return self.__dict__.get("_ver", 0)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"Missing key {k}") from None

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k) from None

    def get(self, k, default=None):
        return self[k] if k in self else default

    def update(self, other=(), **kw):  # type: ignore[override]
        for k, v in dict(other, **kw).items():
            self[k] = v

    def copy(self):  # deep: configs are mutated in place by the pipeline
        return Config(copy.deepcopy(dict(self)))

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _to_container(c):
    if isinstance(c, Mapping):
        return {k: _to_container(v) for k, v in c.items()}
    if isinstance(c, (list, tuple)):
        return [_to_container(v) for v in c]
    return c


def _merge_into(dst: Config, src: Mapping):
    for k, v in src.items():
        if isinstance(v, Mapping) and isinstance(dst.get(k), Mapping):
            _merge_into(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)


class OmegaConf:
    """The four OmegaConf entry points the reference's callers use."""

    @staticmethod
    def create(obj=None):
        return Config(obj or {})

    @staticmethod
    def load(path):
        import yaml

        with open(path, "r") as f:
            return Config(yaml.safe_load(f) or {})

    @staticmethod
    def merge(*configs):
        out = Config()
        for c in configs:
            _merge_into(out, c)
        return out

    @staticmethod
    def to_container(cfg, resolve=True):
        return _to_container(cfg)

    @staticmethod
    def from_dotlist(items):
        """``k.sub=v`` overrides (reference run_handheld.py:104-116)."""
        import yaml

        out = Config()
        for it in items:
            key, _, val = it.partition("=")
            node = out
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    node[p] = {}
                node = node[p]
            node[parts[-1]] = yaml.safe_load(val)
        return out


# Same schema / values as the reference's shipped defaults.
DEFAULTS = {
    "scale": 1,
    "mode": "bayer",
    "debug": False,
    "verbose": 1,
    "grey_method": "FFT",
    "noise_model": {"alpha": None, "beta": None},
    "block_matching": {
        "tuning": {
            # fine-to-coarse
            "factors": [1, 2, 4, 4],
            "tile_size": "SNR_based",
            "tile_size_factors": [1, 1, 1, 0.5],
            "search_radii": [1, 4, 4, 4],
            "metrics": ["L1", "L2", "L2", "L2"],
            "flow_upscale_mode": "nearest",
        }
    },
    "ica": {"tuning": {"n_iter": 3, "sigma_blur": 0}},
    "robustness": {
        "enabled": True,
        "save_mask": True,
        "tuning": {"t": 0.12, "s1": 2, "s2": 12, "Mt": 0.8},
    },
    "merging": {
        "kernel": "steerable",
        "selection_law": "linear",
        "tuning": {
            "k_detail": "SNR_based",
            "k_denoise": "SNR_based",
            "D_th": "SNR_based",
            "D_tr": "SNR_based",
            "k_stretch": 4,
            "k_shrink": 2,
        },
    },
    "postprocessing": {
        "enabled": True,
        "do_color_correction": False,
        "do_gamma_correction": True,
        "do_tonemapping": False,
        "sharpening": {"enabled": True, "amount": 1.5, "radius": 3},
        "do_devignetting": False,
    },
    "accumulated_robustness_denoiser": {
        "median": {"enabled": False, "radius_max": 3, "max_frame_count": 8},
        "gauss": {"enabled": False, "sigma_max": 1.5, "max_frame_count": 8},
        "merge": {"enabled": False, "rad_max": 2, "max_multiplier": 8, "max_frame_count": 2},
    },
    # build-specific switches (absent from the reference schema; all optional)
    "compat": {
        # D2 — reproduce the ts=64 ICA row off-by-one
        "ica64_row_bug": True,
    },
}


def default_config(**overrides) -> Config:
    cfg = Config(copy.deepcopy(DEFAULTS))
    if overrides:
        _merge_into(cfg, overrides)
    return cfg
