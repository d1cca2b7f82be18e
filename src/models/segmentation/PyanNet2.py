from uvad_amd.models import PyanNet2  # noqa: F401  (implementation: universal-voice-activity-detection_amd/models.py)
