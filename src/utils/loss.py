from cabinet_amd.loss import OhemCELoss  # noqa: F401
