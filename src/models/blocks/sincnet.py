from uvad_amd.sincnet import Encoder, ParamSincFB, SincNet  # noqa: F401  (reference: src/models/blocks/sincnet.py:30-103)
