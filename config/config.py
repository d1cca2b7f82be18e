from uvad_amd.config import ConfigDict, load_config  # noqa: F401
