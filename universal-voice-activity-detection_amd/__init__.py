"""MI355X-native voice-activity inference path: fused log-mel front end + PyanNet2 classifier
as hand-written gfx950 HIP kernels behind a C ABI (libuvad.so), with a Python host that mirrors
the reference's ``src.models`` / ``src.engines`` / ``config`` / ``main`` surface."""
from .config import ConfigDict, load_config
from .engine import VadModel
from .features import Fbank, FbankConfig, make_mel_matrix, make_window
from .models import PyanNet, PyanNet2
from .postprocess import (detection_error, intervals_to_labels, labels_to_intervals, labels_to_intervals_batch, median_filter, median_window,
                          merge_intervals_with_buffer, split_into_windows)
from .pipeline import ForwardPipeline
from .runtime import VadRuntime
from .scripts import predict_vad
from .sincnet import SincNet

__all__ = ["ConfigDict", "load_config", "VadModel", "Fbank", "FbankConfig", "make_mel_matrix", "make_window",
           "PyanNet", "PyanNet2", "SincNet", "VadRuntime", "ForwardPipeline", "labels_to_intervals", "labels_to_intervals_batch", "median_filter", "median_window", "predict_vad",
           "detection_error", "intervals_to_labels", "merge_intervals_with_buffer", "split_into_windows"]
