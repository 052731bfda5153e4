"""reference ``loader`` package surface used by train.py on the hot path: transformsgpu, transformmasks
(the PIL dataset loaders are out of scope)."""
from . import transformsgpu, transformmasks, depth_estimator  # noqa: F401
