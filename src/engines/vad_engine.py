from uvad_amd.engine import VadModel  # noqa: F401  (implementation: universal-voice-activity-detection_amd/engine.py)
