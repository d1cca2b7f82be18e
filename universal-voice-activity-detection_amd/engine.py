"""Host-side mirror of ``src.engines.vad_engine.VadModel`` (reference vad_engine.py:20-281),
inference surface only: constructor arguments, ``.model`` / ``.model_name``, ``forward``,
``predict_step``, ``_common_step`` and ``load_from_checkpoint`` keep their names, argument
meaning and return shapes.  Training (``training_step``, torchmetrics, Adam) is outside the
accelerated path and raises."""
import torch
import torch.nn as nn

from .models import PyanNet, PyanNet2
from .postprocess import median_filter


class VadModel(nn.Module):
    def __init__(self, model_name: str = "PyanNet2", model_dict: dict = None, learning_rate: float = 1e-3):
        super().__init__()
        model_dict = dict(model_dict or {})
        self.model_name = model_name
        self.model = PyanNet(**model_dict) if model_name == "PyanNet" else PyanNet2(**model_dict)
        self.model.build()
        self.learning_rate = learning_rate

    # -- inference ---------------------------------------------------------------------------
    def forward(self, audio_feats: torch.Tensor) -> torch.Tensor:
        return self.model(audio_feats)

    def _common_step(self, batch, batch_idx):
        """vad_engine.py:247-278.  The reference also evaluates BCE against ``batch["is_voice"]``
        here (its value is unused by predict); it is computed only when labels are present."""
        x = batch["inputs"]
        y_pred = self.model(x.unsqueeze(1)) if self.model_name == "PyanNet" else self.model(x)
        y = batch.get("is_voice")
        loss = None
        if y is not None:
            loss = nn.functional.binary_cross_entropy(y_pred.squeeze(-1), y.to(y_pred.device, y_pred.dtype))
            if torch.isnan(loss):
                return None
        return {"loss": loss}, y_pred, y

    def predict_step(self, batch, batch_idx=0, dataloader_idx=None):
        """vad_engine.py:204-211: probabilities -> threshold 0.5 -> median filter (49 taps at a
        10 ms hop, 25 at 20 ms) -> (batch, frames, 1) of 0/1."""
        _, y_pred, _ = self._common_step(batch, batch_idx)
        window = 0.02 if self.model.encoding_dim == 768 else 0.01
        labels = median_filter(y_pred.squeeze(-1), window=window)
        return labels.unsqueeze(-1)

    # -- checkpoints ---------------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, **kwargs):
        """Accepts a Lightning ``.ckpt`` (dict with ``state_dict`` whose keys carry the ``model.``
        prefix) or a plain ``state_dict`` file.  Loaded with ``weights_only=True``.  As in the
        reference (predict.py:77) constructor arguments are NOT stored in the checkpoint, so
        ``model_dict`` must be passed for anything but the 768-dim default."""
        blob = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=True)
        sd = blob.get("state_dict", blob) if isinstance(blob, dict) else blob
        obj = cls(**kwargs)
        own = {}
        for k, v in sd.items():
            if not torch.is_tensor(v):
                continue
            own[k if k.startswith("model.") else "model." + k] = v
        missing, unexpected = obj.load_state_dict(own, strict=False)
        missing = [m for m in missing if not m.endswith("num_batches_tracked")]
        if missing:
            raise RuntimeError(f"checkpoint is missing tensors: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        return obj

    # -- training is out of scope --------------------------------------------------------------
    def training_step(self, *a, **k):
        raise NotImplementedError("training is outside the accelerated inference path (SURVEY.md section 2, rows 6/12)")

    validation_step = test_step = configure_optimizers = training_step
