"""Import harness for the upstream reference (test infrastructure only).

This file is used ONLY in the build container, by ``oracle/make_golden.py``, to
import ``/root/reference/src/asr.py`` and run it on CPU so that golden
input/output vectors can be captured under ``tests/golden/``.  Nothing here is
imported by the product package, by ``bench.py`` or by the ``-m gpu`` tests;
``/root/reference`` does not exist on the GPU box.

The reference cannot be imported as shipped: it imports
``editdistance``, ``librosa`` and ``tensorboardX`` (absent from this image, and
unrelated to the ASR arithmetic), it imports a name ``Hypothesis`` that its own
``postprocess.py`` never defines, and on torch >= 2 ``masked_fill_`` rejects
the uint8 mask built at ``src/asr.py:379``.  The harness supplies empty
stand-in *modules* for the unrelated third-party imports and casts the uint8
mask to bool; it does not alter any reference arithmetic and writes nothing
under ``/root/reference``.
"""
import os
import sys
import types

import torch

REF_SRC = os.environ.get("SSASR_REFERENCE_SRC", "/root/reference/src")


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    for key, val in attrs.items():
        setattr(mod, key, val)
    sys.modules[name] = mod
    return mod


def import_reference():
    """Returns the reference's ``asr`` module (imported from REF_SRC)."""
    if not os.path.isdir(REF_SRC):
        raise RuntimeError("reference sources not present at %s" % REF_SRC)
    sys.dont_write_bytecode = True

    def _absent(*_a, **_k):
        raise RuntimeError("third-party dependency absent in this image")

    _stub("editdistance", eval=_absent)
    librosa = _stub("librosa")
    librosa.core = _stub("librosa.core", load=_absent, power_to_db=_absent)
    librosa.feature = _stub("librosa.feature", melspectrogram=_absent)
    librosa.display = _stub("librosa.display", specshow=_absent)
    _stub("tensorboardX", SummaryWriter=object)

    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)

    import postprocess  # noqa: E402  (reference module)
    if not hasattr(postprocess, "Hypothesis"):
        postprocess.Hypothesis = object

    # torch >= 1.2 wants bool masks; the reference builds a ByteTensor.
    if not getattr(torch.Tensor.masked_fill_, "_ssasr_wrapped", False):
        _orig = torch.Tensor.masked_fill_

        def masked_fill_(self, mask, value):
            if mask.dtype == torch.uint8:
                mask = mask.bool()
            return _orig(self, mask, value)

        masked_fill_._ssasr_wrapped = True
        torch.Tensor.masked_fill_ = masked_fill_

    import asr  # noqa: E402  (reference module)
    assert os.path.abspath(asr.__file__).startswith(os.path.abspath(REF_SRC))
    return asr
