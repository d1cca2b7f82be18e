from uvad_amd.sincnet import SincNet  # noqa: F401  (reference: src/models/blocks/sincnet.py)
