"""Entry point with the reference's shape (main.py:12-55): ``main(load_config())`` dispatches on
``config.task`` / ``config.function``.  Only the inference path is in scope; every other task of
the reference (corpus download / preparation, training, scoring) raises with a pointer to SURVEY.md."""
import faulthandler

faulthandler.enable()

from config.config import load_config  # noqa: E402
from src.scripts import predict_vad    # noqa: E402


def main(config):
    if config.task == "run":
        if config.function == "predict":
            return predict_vad(**config)
        raise NotImplementedError(f"function={config.function!r}: only 'predict' is on the accelerated path")
    raise NotImplementedError(f"task={config.task!r}: data preparation / scoring tasks are out of scope (SURVEY.md section 2)")


if __name__ == "__main__":
    import torch

    print("GPU:", torch.cuda.is_available())
    main(load_config())
