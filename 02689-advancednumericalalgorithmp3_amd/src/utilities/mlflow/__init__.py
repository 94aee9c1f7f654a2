"""Import path of the reference's Hydra callback (``conf/config.yaml:40-42`` there targets
``utilities.mlflow.callback.MLflowSweepCallback``); the bookkeeping itself lives in ``utilities.tracking.sweep``."""
