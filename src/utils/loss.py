from cabinet_amd.loss import OhemCELoss, SoftmaxFocalLoss  # noqa: F401
