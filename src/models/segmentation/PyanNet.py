from uvad_amd.models import PyanNet  # noqa: F401  (reference: src/models/segmentation/PyanNet.py; implementation: universal-voice-activity-detection_amd/models.py)
