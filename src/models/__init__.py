from uvad_amd.models import PyanNet, PyanNet2  # reference: src/models/__init__.py:1-2

__all__ = ["PyanNet", "PyanNet2"]
