from .metrics import runningScore  # noqa: F401
