from uvad_amd.scripts import predict_vad  # noqa: F401

__all__ = ["predict_vad"]
