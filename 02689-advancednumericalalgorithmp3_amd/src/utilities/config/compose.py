"""Hydra-free composition of the ``conf/`` tree (PyYAML only).

The reference is launched through Hydra 1.3 (``@hydra.main(config_path="conf", config_name="config")``,
main.py:228) which is not installed here.  This module implements the subset of Hydra's grammar the
reference's configs and documented command lines use:

* defaults lists with ``_self_``, nested defaults with absolute paths (``/solver/spectral/sg``),
  ``override /group: option`` inside appended experiment files;
* ``# @package`` headers (``_global_``, ``solver``, ``hydra.sweeper`` ...);
* ``${a.b}`` / ``${oc.env:VAR[,default]}`` / ``${now:%fmt}`` interpolation;
* command line: ``group=option``, ``+group=option``, ``dotted.key=value``, ``-m`` with comma lists
  (cartesian product), ``interval(a,b)`` / ``choice(...)`` search spaces, and the
  ``hydra.sweeper.params`` block of experiment files as the sweep source.
"""
from __future__ import annotations

import copy
import datetime as _dt
import itertools
import os
import re
from dataclasses import dataclass
from pathlib import Path

import yaml


@dataclass(frozen=True)
class Interval:
    low: float
    high: float


class ConfigError(ValueError):
    pass


# ------------------------------------------------------------------------------- helpers
def _deep_merge(dst: dict, src: dict) -> dict:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(cfg: dict, dotted: str, value):
    keys = dotted.split(".")
    for k in keys[:-1]:
        cfg = cfg.setdefault(k, {})
        if not isinstance(cfg, dict):
            raise ConfigError(f"cannot set {dotted}: {k} is not a mapping")
    cfg[keys[-1]] = value


def _get_path(cfg: dict, dotted: str):
    cur = cfg
    for k in dotted.split("."):
        if not isinstance(cur, dict) or k not in cur:
            raise KeyError(dotted)
        cur = cur[k]
    return cur


def parse_value(text: str):
    """YAML scalar semantics for a command-line value (``1e-6`` is a float like in OmegaConf)."""
    t = text.strip()
    if re.fullmatch(r"[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+", t):
        return float(t)
    try:
        return yaml.safe_load(t)
    except yaml.YAMLError:
        return t


_INTERVAL = re.compile(r"^\s*interval\(\s*([^,]+),\s*([^)]+)\)\s*$")
_CHOICE = re.compile(r"^\s*choice\((.*)\)\s*$")


def parse_sweep_value(value):
    """'a, b, c' -> list; 'interval(lo, hi)' -> Interval; scalar -> [scalar]."""
    if isinstance(value, str):
        m = _INTERVAL.match(value)
        if m:
            return Interval(float(m.group(1)), float(m.group(2)))
        m = _CHOICE.match(value)
        body = m.group(1) if m else value
        if m or "," in body:
            return [parse_value(p) for p in body.split(",") if p.strip()]
        return [parse_value(value)]
    if isinstance(value, (list, tuple)):
        return list(value)
    return [value]


def _package_of(text: str, default: str) -> str:
    m = re.match(r"\s*#\s*@package\s+(\S+)", text)
    return m.group(1) if m else default


# ------------------------------------------------------------------------------- composition
class Composer:
    def __init__(self, conf_dir):
        self.root = Path(conf_dir)
        if not (self.root / "config.yaml").exists():
            raise ConfigError(f"{self.root}/config.yaml not found")

    def is_group(self, key: str) -> bool:
        return (self.root / key).is_dir()

    def _load(self, group: str, option: str):
        path = self.root / group / f"{option}.yaml"
        if not path.exists():
            raise ConfigError(f"Could not find '{group}/{option}' ({path})")
        text = path.read_text()
        body = yaml.safe_load(text) or {}
        return body, _package_of(text, group.replace("/", "."))

    def _place(self, cfg: dict, package: str, body: dict):
        if package == "_global_":
            _deep_merge(cfg, body)
        else:
            node = cfg
            for k in package.split("."):
                node = node.setdefault(k, {})
            _deep_merge(node, body)

    def _merge_group(self, cfg: dict, group: str, option: str, choices: dict, seen: set):
        """Merge group/option (its own nested defaults first); collect ``override`` requests."""
        if (group, option) in seen:
            return
        seen.add((group, option))
        body, package = self._load(group, option)
        for d in body.pop("defaults", []) or []:
            if d == "_self_":
                continue
            if isinstance(d, str):                         # '/solver/spectral/sg' : absolute file
                g, _, o = d.strip("/").rpartition("/")
                self._merge_group(cfg, g, o, choices, seen)
                continue
            (k, v), = d.items()
            if k.startswith("override "):
                continue                                   # handled in the pre-pass
            self._merge_group(cfg, k.strip("/"), v, choices, seen)
        self._place(cfg, package, body)

    def _overrides_in(self, group: str, option: str) -> dict:
        body, _ = self._load(group, option)
        out = {}
        for d in body.get("defaults", []) or []:
            if isinstance(d, dict):
                (k, v), = d.items()
                if k.startswith("override "):
                    out[k[len("override "):].strip().strip("/")] = v
        return out

    def compose(self, overrides=()):
        """Returns (cfg, sweep_overrides): cfg is fully merged but NOT interpolated."""
        group_choice, appended, values = {}, [], []
        for ov in overrides:
            if "=" not in ov:
                raise ConfigError(f"bad override '{ov}' (expected key=value)")
            key, val = ov.split("=", 1)
            plus = key.startswith("+")
            key = key.lstrip("+")
            if self.is_group(key) and "," not in val:
                if plus:
                    appended.append((key, val))
                else:
                    group_choice[key] = val
            else:                                          # a value, or a swept group (solver=a,b)
                values.append((key, val))

        root = yaml.safe_load((self.root / "config.yaml").read_text()) or {}
        defaults = root.pop("defaults", [])
        order, choices = [], {}
        for d in defaults:
            if d == "_self_":
                order.append("_self_")
            else:
                (g, o), = d.items()
                order.append(g)
                choices[g] = o
        # pre-pass: experiment files may override earlier groups; the command line wins
        requested = {}
        for g, o in appended:
            requested.update(self._overrides_in(g, o))
        for g, o in requested.items():
            if g not in choices:
                order.append(g)
            choices[g] = o
        for g, o in group_choice.items():
            if g not in choices:
                order.append(g)
            choices[g] = o

        cfg, seen = {}, set()
        if "_self_" not in order:
            order.append("_self_")
        for item in order:
            if item == "_self_":
                _deep_merge(cfg, root)
            elif self.is_group(item) and (self.root / item / f"{choices[item]}.yaml").exists():
                self._merge_group(cfg, item, choices[item], choices, seen)
            elif item.startswith("hydra/launcher"):
                continue                                   # plugin configs that only exist inside Hydra
            else:
                raise ConfigError(f"Could not find '{item}/{choices[item]}'")
        for g, o in appended:
            self._merge_group(cfg, g, o, choices, seen)
        cfg.setdefault("hydra", {}).setdefault("choices", {}).update(choices)
        return cfg, values


def apply_values(cfg: dict, pairs):
    """Apply dotted ``key=value`` pairs (group choices are handled by re-composition)."""
    for key, val in pairs:
        _set_path(cfg, key, val if not isinstance(val, str) else parse_value(val))
    return cfg


def compose_job(composer: Composer, overrides, pairs):
    """Configuration of ONE job: group-valued pairs (e.g. a swept ``solver``) re-compose the tree so
    that experiment-level content still merges on top of the chosen group; the rest are values."""
    pairs = list(pairs)
    group_pairs = [(k, v) for k, v in pairs if composer.is_group(k)]
    keys = {k for k, _ in group_pairs}
    base = [o for o in overrides if o.split("=", 1)[0].lstrip("+") not in keys]
    cfg, cli_values = composer.compose(base + [f"{k}={v}" for k, v in group_pairs])
    swept = {k for k, _ in pairs}
    apply_values(cfg, [(k, v) for k, v in cli_values if k not in swept and not composer.is_group(k)])
    apply_values(cfg, [(k, v) for k, v in pairs if not composer.is_group(k)])
    return cfg


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def resolve(cfg: dict) -> dict:
    """Resolve every interpolation in place (returns cfg)."""
    now = _dt.datetime.now()

    def lookup(expr: str):
        if expr.startswith("oc.env:"):
            name, _, default = expr[len("oc.env:"):].partition(",")
            if name in os.environ:
                return os.environ[name]
            if default != "":
                return parse_value(default)
            raise ConfigError(f"environment variable {name} is not set")
        if expr.startswith("now:"):
            return now.strftime(expr[len("now:"):])
        return res(_get_path(cfg, expr), 0)

    def res(v, depth):
        if depth > 20:
            raise ConfigError("interpolation cycle")
        if isinstance(v, str):
            m = _INTERP.fullmatch(v.strip())
            if m:                                           # whole-value interpolation keeps the type
                return res(lookup(m.group(1)), depth + 1)
            while _INTERP.search(v):
                v = _INTERP.sub(lambda mm: str(lookup(mm.group(1))), v)
            return v
        if isinstance(v, dict):
            return {k: res(x, depth) for k, x in v.items()}
        if isinstance(v, list):
            return [res(x, depth) for x in v]
        return v

    skip = cfg.get("hydra", {})
    out = {k: (res(v, 0) if k != "hydra" else v) for k, v in cfg.items()}
    cfg.clear()
    cfg.update(out)
    if skip:
        cfg["hydra"] = skip
    return cfg


def sweep_space(cfg: dict, cli_values, multirun: bool):
    """Ordered {key: list | Interval}: the experiment's ``hydra.sweeper.params`` then the command line."""
    space = {}
    params = (cfg.get("hydra", {}).get("sweeper", {}) or {}).get("params") or {}
    for k, v in params.items():
        space[k] = parse_sweep_value(v)
    fixed = []
    for k, v in cli_values:
        sv = parse_sweep_value(v)
        if multirun and (isinstance(sv, Interval) or len(sv) > 1):
            space[k] = sv
        else:
            space.pop(k, None)
            fixed.append((k, v))
    return space, fixed


def expand_grid(space: dict) -> list:
    """Cartesian product in Hydra's order (first key slowest)."""
    keys = [k for k, v in space.items() if not isinstance(v, Interval)]
    return [list(zip(keys, combo)) for combo in itertools.product(*(space[k] for k in keys))] or [[]]


def instantiate(node: dict, **extra):
    """``hydra.utils.instantiate`` for a flat ``_target_`` node."""
    import importlib
    node = dict(node)
    target = node.pop("_target_")
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)(**node, **extra)
