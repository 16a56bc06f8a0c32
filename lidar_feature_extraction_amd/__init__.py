"""MI355X-native lidar feature extraction (the extraction/ hot path of tier4/lidar_feature_extraction).

HIP kernels + C ABI live in csrc/ (built to _lib/liblfx.so); `FeatureExtraction` is the host-side
mirror of the reference node's operator.  Importing this package does not need a GPU; creating a
`FeatureExtraction` does, and raises if the library or the device is missing (no CPU fallback).
"""
from .extraction import FeatureExtraction, HyperParameters, ScanFeatures, LABEL_NAMES, RING_STATUS_NAMES, layout_from_fields  # noqa: F401
from .synth import POINT_DTYPE, SENSORS, make_scan, make_batch, concat  # noqa: F401
