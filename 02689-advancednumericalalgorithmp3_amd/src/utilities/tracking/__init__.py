"""Experiment tracking (MLflow) glue of the launcher; active only when ``mlflow`` is importable."""
