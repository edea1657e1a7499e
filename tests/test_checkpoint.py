"""diart_amd/checkpoint.py: real checkpoints are PyTorch-Lightning pickles that carry objects of pyannote.audio
classes (the reference loads them through pyannote, /root/reference/src/diart/models.py:50, :59) or plain
torch.save files (speechbrain).  They must load on a box WITHOUT those packages, as tensors only, and a hostile
pickle must not get to run anything."""
import os
import pickle
import sys
import types

import pytest
import torch

from diart_amd import checkpoint
from diart_amd.models import _read_state
from diart_amd.synth import embedding_spec, segmentation_spec, synth_ecapa_state, synth_embedding_state, synth_segmentation_state


def _foreign_module(name):
    """A throw-away package `name` with classes like the ones pyannote pickles; removed again by the caller."""
    parts = name.split(".")
    for i in range(1, len(parts) + 1):
        sys.modules.setdefault(".".join(parts[:i]), types.ModuleType(".".join(parts[:i])))
    mod = sys.modules[name]

    class Specifications:
        def __init__(self, problem, resolution, duration, classes):
            self.problem, self.resolution, self.duration, self.classes = problem, resolution, duration, classes

    import enum

    class Problem(enum.Enum):
        MULTI_LABEL_CLASSIFICATION = 2

    class Bag(dict):                      # a dict subclass: SETITEMS on a stub
        pass

    for c in (Specifications, Problem, Bag):
        c.__module__, c.__qualname__ = name, c.__name__
        setattr(mod, c.__name__, c)
    return mod


def _forget(name):
    for k in [k for k in sys.modules if k == name.split(".")[0] or k.startswith(name.split(".")[0] + ".")]:
        del sys.modules[k]


@pytest.mark.parametrize("zipped", [True, False])
def test_lightning_checkpoint_with_classes_of_a_missing_package_loads_as_tensors(tmp_path, zipped):
    state = synth_segmentation_state(seed=5)
    mod = _foreign_module("pyannote_like.audio.core.task")
    try:
        bag = mod.Bag()
        bag["versions"] = {"torch": "1.7.1"}
        ckpt = {"epoch": 19, "global_step": 12345, "pytorch-lightning_version": "1.1.3",
                "state_dict": {"model." + k: v for k, v in state.items()},
                "hyper_parameters": {"sincnet": {"stride": 10}, "lstm": bag},
                "pyannote.audio": {"specifications": mod.Specifications(mod.Problem.MULTI_LABEL_CLASSIFICATION, 2, 5.0,
                                                                        ["speaker#1", "speaker#2", "speaker#3"]),
                                   "architecture": {"module": "pyannote.audio.models.segmentation", "class": "PyanNet"}},
                "optimizer_states": [{"state": {0: {"exp_avg": torch.randn(3, 4)}}, "param_groups": [{"lr": 1e-3}]}]}
        f = tmp_path / ("seg_zip.ckpt" if zipped else "seg_legacy.ckpt")
        torch.save(ckpt, f, _use_new_zipfile_serialization=zipped)
    finally:
        _forget("pyannote_like")
    with pytest.raises(Exception):                 # what models.py did until round 3
        torch.load(str(f), map_location="cpu", weights_only=False)
    obj, stubbed = checkpoint.load_object(f)
    assert any("Specifications" in s for s in stubbed) and any("Problem" in s for s in stubbed)
    assert obj["epoch"] == 19 and obj["hyper_parameters"]["sincnet"]["stride"] == 10
    got = _read_state(f)                            # the 'model.' prefix is stripped by the LOADER
    assert set(got) == set(state)
    for k, v in state.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and torch.equal(got[k], v), k
    spec = {k: s for k, s, _ in segmentation_spec(num_speakers=got["classifier.weight"].shape[0])}
    assert not [k for k in spec if k not in got] and not [k for k in spec if tuple(got[k].shape) != spec[k]]


@pytest.mark.parametrize("version,want", [("3.1.1", "nearest"), ("3.3.2", "nearest"), ("3.0.1", "linear"), ("2.1.1", "linear"),
                                          ("4.0.0.dev1", "nearest"), (None, "linear")])
def test_embedding_loader_takes_the_pooling_mode_from_the_checkpoint_version(tmp_path, version, want):
    """pyannote.audio >= 3.1 resamples StatsPool's weights with mode="nearest", older releases with "linear"
    (VERDICT r5 #8): the loader reads the version a Lightning checkpoint records and sets `weight_interp`; an explicit
    argument wins; a plain state dict means "linear" (the reference's pin is >= 2.1.1)."""
    from diart_amd.models import EmbeddingLoader
    from diart_amd.synth import synth_embedding_state
    state = synth_embedding_state(seed=2)
    ckpt = {"state_dict": {"model." + k: v for k, v in state.items()}, "pytorch-lightning_version": "1.6.5"}
    if version is not None:
        ckpt["pyannote.audio"] = {"versions": {"torch": "2.0.1", "pyannote.audio": version},
                                  "architecture": {"module": "pyannote.audio.models.embedding", "class": "XVectorSincNet"}}
    f = tmp_path / "emb.ckpt"
    torch.save(ckpt, f)
    assert checkpoint.pyannote_version(f) == (None if version is None else tuple(int(x) for x in version.split(".")[:2]))
    m = EmbeddingLoader(str(f), max_batch=3)()
    assert type(m).__name__ == "HipEmbedding" and m.weight_interp == want
    assert EmbeddingLoader(str(f), max_batch=3, weight_interp="linear")().weight_interp == "linear"
    assert EmbeddingLoader(state, max_batch=3)().weight_interp == "linear"
    import pickle
    again = pickle.loads(pickle.dumps(m))
    assert again.weight_interp == want


def test_plain_state_dicts_speechbrain_shape_and_safetensors(tmp_path):
    ecapa = synth_ecapa_state(seed=3)
    f = tmp_path / "embedding_model.ckpt"
    torch.save(ecapa, f)
    got = _read_state(f)
    assert set(got) == set(ecapa) and all(torch.equal(got[k], v) for k, v in ecapa.items())
    assert got["blocks.0.norm.norm.num_batches_tracked"].dtype == torch.int64          # non-float entries survive
    from safetensors.torch import save_file
    emb = {k: v.contiguous() for k, v in synth_embedding_state(seed=4).items()}
    g = tmp_path / "embedding.safetensors"
    save_file(emb, str(g))
    got = _read_state(g)
    assert set(got) == set(emb) and all(torch.equal(got[k], v) for k, v in emb.items())
    spec = {k: s for k, s, _ in embedding_spec()}
    assert not [k for k in spec if k not in got]


def test_views_shared_storages_and_other_dtypes(tmp_path):
    base = torch.arange(24, dtype=torch.float32).reshape(4, 6)
    sd = {"view": base[1:3, ::2], "t": base.t(), "half": torch.randn(5).half(), "bf": torch.randn(3, 2).bfloat16(),
          "i64": torch.tensor([1, 2, 3]), "flag": torch.tensor([True, False]), "empty": torch.zeros(0, 7),
          "scalar": torch.tensor(3.5), "param": torch.nn.Parameter(torch.randn(2, 2))}
    for zipped in (True, False):
        f = tmp_path / f"misc_{zipped}.pt"
        torch.save(sd, f, _use_new_zipfile_serialization=zipped)
        got = checkpoint.read_state(f)
        assert set(got) == set(sd)
        for k, v in sd.items():
            assert got[k].dtype == v.dtype and got[k].shape == v.shape and torch.equal(got[k], v.detach()), (zipped, k)


class _Hostile:
    def __reduce__(self):
        return (os.system, ("touch /tmp/dz_checkpoint_pwned",))


def test_a_hostile_pickle_does_not_run(tmp_path):
    marker = "/tmp/dz_checkpoint_pwned"
    if os.path.exists(marker):
        os.remove(marker)
    f = tmp_path / "evil.ckpt"
    torch.save({"state_dict": {"w": torch.ones(2)}, "callbacks": _Hostile()}, f)
    got = checkpoint.read_state(f)
    assert torch.equal(got["w"], torch.ones(2)) and not os.path.exists(marker)
    raw = tmp_path / "evil_raw.bin"                 # not a torch file at all: a bare pickle
    raw.write_bytes(pickle.dumps(_Hostile()))
    with pytest.raises(Exception):
        checkpoint.read_state(raw)
    assert not os.path.exists(marker)


def test_errors_are_told(tmp_path):
    with pytest.raises(FileNotFoundError):
        checkpoint.read_state(tmp_path / "nope.bin")
    f = tmp_path / "notstate.pt"
    torch.save([1, 2, 3], f)
    with pytest.raises(ValueError, match="state dict"):
        checkpoint.read_state(f)


def test_model_prefix_is_stripped_per_key_and_unknown_storages_are_named(tmp_path):
    """one extra top-level entry beside `model.*` must not keep the prefix on every key; a tensor on a
    storage class the reader does not know is an error that names the entry, not a silent drop."""
    state = synth_segmentation_state(seed=6)
    sd = {"model." + k: v for k, v in state.items()}
    sd["loss_weight"] = torch.tensor([0.5])
    f = tmp_path / "extra.ckpt"
    torch.save({"state_dict": sd}, f)
    got = _read_state(f)
    assert set(got) == set(state) | {"loss_weight"} and torch.equal(got["sincnet.conv1d.1.weight"], state["sincnet.conv1d.1.weight"])
    # a collision (both `x` and `model.x`) keeps the names as they are
    g = tmp_path / "collide.ckpt"
    torch.save({"state_dict": {"model.w": torch.ones(2), "w": torch.zeros(2)}}, g)
    assert set(_read_state(g)) == {"model.w", "w"}
    assert isinstance(checkpoint._rebuild_tensor(object(), 0, (2,), (1,)), checkpoint.UnknownStorage)
    # what an unknown storage class unpickles to reaches read_state as a marker, and is named
    orig = checkpoint.load_object
    try:
        checkpoint.load_object = lambda p: ({"state_dict": {"q.weight": checkpoint.UnknownStorage(), "w": torch.ones(1)}}, set())
        with pytest.raises(ValueError, match="q.weight"):
            checkpoint.read_state(f)
    finally:
        checkpoint.load_object = orig


def test_verify_real_tool_checks_checkpoints_without_a_gpu(tmp_path):
    """tools/verify_real.py, step 1 (the part that runs anywhere): Lightning-style checkpoints of the two
    architectures are found, read as tensors and matched against the architectures' key / shape lists; a
    checkpoint with a missing layer is refused by name."""
    import json
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    seg, emb = synth_segmentation_state(seed=8), synth_embedding_state(seed=9)
    torch.save({"state_dict": {"model." + k: v for k, v in seg.items()}, "epoch": 3}, tmp_path / "segmentation.ckpt")
    torch.save({k: v for k, v in emb.items()}, tmp_path / "embedding.bin")
    r = subprocess.run([sys.executable, str(root / "tools" / "verify_real.py"), "--ckpt", str(tmp_path), "--skip-gates"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rep = json.loads(r.stdout[r.stdout.index("{"):])
    assert rep["segmentation_keys"]["missing"] == [] and rep["embedding_keys"]["wrong_shape"] == []
    bad = dict(emb)
    del bad["embedding.weight"]
    torch.save(bad, tmp_path / "embedding.bin")
    r = subprocess.run([sys.executable, str(root / "tools" / "verify_real.py"), "--ckpt", str(tmp_path), "--skip-gates"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "embedding.weight" in r.stderr
