"""Drop-in import surface of the reference (``src.models``, ``src.engines``, ``src.utils``,
``src.scripts``); every name resolves to the MI355X implementation in
``universal-voice-activity-detection_amd``."""
