"""The C-ABI library loads without a GPU and exports every symbol include/diart_amd.h declares;
host-only entry points (frame geometry, clustering, LSAP) work; GPU entry points fail loudly."""
import ctypes as C
import re
from pathlib import Path

import pytest
import torch

from diart_amd import _lib

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    text = (ROOT / "include" / "diart_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dz_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 35
    lib = C.CDLL(str(_lib.lib_path()))
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/diart_amd.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype in diart_amd/_lib.py"
    assert set(_lib.SIGNATURES) <= set(names)


def test_geometry_and_version():
    lib = _lib.load()
    assert lib.dz_version() == 100
    assert lib.dz_seg_frames_for(80000) == 293 and lib.dz_seg_frames_for(160000) == 589
    assert lib.dz_emb_frames_for(80000) == 279
    assert lib.dz_seg_frames_for(200) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    from diart_amd import functional, models
    from diart_amd.synth import synth_segmentation_state
    with pytest.raises(_lib.DiartAmdError):
        _lib.context(0)                                   # hipGetDeviceCount fails without a GPU
    with pytest.raises(_lib.DiartAmdError):
        functional.overlapped_speech_penalty(torch.rand(1, 10, 3))
    m = models.SegmentationModel.from_state(synth_segmentation_state())
    with pytest.raises(_lib.DiartAmdError):
        m.to(torch.device("cpu"))


def test_product_never_imports_oracle():
    for f in (ROOT / "diart_amd").rglob("*.py"):
        assert "oracle" not in f.read_text().replace("the oracle", ""), f


def test_struct_layouts_match_the_library():
    """The ctypes mirrors of the structs that cross the C ABI have the sizes the library was
    compiled with (load() refuses a mismatching library; this keeps the check itself honest)."""
    import ctypes as C
    from diart_amd import _lib
    lib = _lib.load()
    sizes = (C.c_int * 5)()
    assert lib.dz_abi_struct_sizes(C.byref(sizes)) == 0
    mine = [C.sizeof(t) for t in (_lib.SincNetWeights, _lib.SegWeights, _lib.EmbWeights,
                                  _lib.EcapaWeights, _lib.ConvGemmDesc)]
    assert list(sizes) == mine and all(v > 0 for v in mine)
