"""``MLflowSweepCallback`` for REAL Hydra multiruns (reference src/utilities/mlflow/callback.py:13-352, same import
path so that the reference's ``hydra.callbacks.mlflow_sweep._target_`` resolves unchanged).

Only importable where Hydra is installed; ``main.py --hydra`` (or ``LDC_LAUNCHER=hydra``) uses it.  The three hooks
delegate to ``utilities.tracking.sweep.SweepTracker``, which the Hydra-free launcher calls directly."""
from __future__ import annotations

import logging

from hydra.experimental.callback import Callback
from omegaconf import OmegaConf

from utilities.tracking.sweep import SweepTracker

log = logging.getLogger(__name__)


def _plain(config) -> dict:
    """The job config as plain containers; the ``hydra`` node is left out and unresolvable interpolations
    (``${hydra:...}`` outside a job) are kept as they are."""
    try:
        d = OmegaConf.to_container(config, resolve=True)
    except Exception:
        d = OmegaConf.to_container(config, resolve=False)
    return {k: v for k, v in d.items() if k != "hydra"}


class MLflowSweepCallback(Callback):
    """Creates or reuses parent MLflow runs for Hydra multiruns; child runs find them through
    ``MLFLOW_PARENT_RUN_ID`` (``main.py`` opens the child run before ``solve()``)."""

    def __init__(self) -> None:
        self._tracker = SweepTracker.create()

    def on_multirun_start(self, config, **kwargs) -> None:
        if self._tracker is None:
            return
        raw = OmegaConf.to_container(config, resolve=False).get("sweep_name", "sweep")
        self._tracker.start(_plain(config), raw_sweep_name=raw)

    def on_job_start(self, config, **kwargs) -> None:
        if self._tracker is not None:
            self._tracker.parent_for(_plain(config))

    def on_multirun_end(self, config, **kwargs) -> None:
        if self._tracker is not None:
            self._tracker.finish(_plain(config), records=[])
