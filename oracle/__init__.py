"""CPU ORACLE -- test infrastructure, not the product.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  Nothing under ``universal-voice-activity-detection_amd/`` (the
product) imports it; the product fails loudly when its HIP library is missing.

Two restatements of the same path live here:

* ``c_oracle``  -- ctypes binding of ``uvad_oracle.c`` (scalar C, double accumulation);
                   the parity checker for the HIP kernels.
* ``torch_ref`` -- the same ops written with stock ``torch`` CPU operators
                   (``nn.LSTM``/``nn.Linear``/``torch.fft.rfft``), i.e. the operator
                   sequence of the reference's ``PyanNet2.forward``
                   (src/models/segmentation/PyanNet2.py:154-187) and of lhotse's Fbank;
                   multi-threaded, so it is what ``bench.py`` times as ``cpu_baseline``
                   (kind "port").
"""
