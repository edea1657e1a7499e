"""Reading model checkpoints WITHOUT running their pickles.

The reference gets its weights through ``pyannote.audio.Model.from_pretrained`` /
``PretrainedSpeakerEmbedding`` (/root/reference/src/diart/models.py:50, :59): a PyTorch-Lightning checkpoint whose
pickle also carries objects of ``pyannote.audio`` classes (``pyannote.audio.core.task.Specifications``, enums, ...),
or speechbrain's ``embedding_model.ckpt`` (a plain ``torch.save`` of a state dict).  ``torch.load(weights_only=False)``
needs those packages to be importable and executes whatever the file's pickle asks for — a file the user was told to
download; ``weights_only=True`` refuses the foreign classes.  This module extracts ONLY tensors:

* ``.safetensors`` files through ``safetensors``;
* ``torch.save`` archives, both the zip form (``<name>/data.pkl`` + ``<name>/data/<key>``) and the legacy stream
  (magic number, three header pickles, the object, then the storages), through a ``pickle.Unpickler`` whose
  ``find_class`` knows tensors, storages, dtypes, ``OrderedDict`` and a few builtins; EVERY other global becomes an
  inert stub class (constructing, calling or setting the state of a stub does nothing), so a checkpoint that
  pickles classes of modules that are not installed loads, and a hostile ``__reduce__`` is never called.

``read_state(path)`` -> ``{name: tensor}``: the checkpoint's ``state_dict`` entry if it has one (Lightning), the
object itself if it is a flat mapping of tensors; a ``model.`` prefix shared by every key is stripped.
"""
from __future__ import annotations

import io
import pickle
import struct
import zipfile
from collections import OrderedDict
from pathlib import Path
from typing import Any, Dict, Union

import torch

_MAGIC = 0x1950A86A20F9469CFC6C

_STORAGE_DTYPES = {
    "FloatStorage": torch.float32, "DoubleStorage": torch.float64, "HalfStorage": torch.float16,
    "BFloat16Storage": torch.bfloat16, "LongStorage": torch.int64, "IntStorage": torch.int32,
    "ShortStorage": torch.int16, "CharStorage": torch.int8, "ByteStorage": torch.uint8, "BoolStorage": torch.bool,
    "ComplexFloatStorage": torch.complex64, "ComplexDoubleStorage": torch.complex128,
}


class _StorageType:
    """Stands in for ``torch.FloatStorage`` & co inside persistent ids."""

    def __init__(self, name: str):
        self.name, self.dtype = name, _STORAGE_DTYPES[name]


class _Storage:
    """Bytes of one storage (filled at once for zip archives, after the pickle for the legacy stream)."""

    def __init__(self, dtype: torch.dtype, nbytes: int):
        self.dtype, self.buf = dtype, bytearray(nbytes)


class Stub:
    """What every global outside the allow-list unpickles to: accepts anything, does nothing."""

    def __new__(cls, *args, **kwargs):
        return object.__new__(cls)

    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, *args, **kwargs):
        return Stub()

    def __setstate__(self, state):
        pass

    def __reduce__(self):
        return (Stub, ())

    # containers that get items appended / set during unpickling (APPENDS / SETITEMS on a stubbed subclass)
    def append(self, item):
        pass

    def extend(self, items):
        pass

    def __setitem__(self, key, value):
        pass

    def add(self, item):
        pass


def _stub_class(module: str, name: str):
    return type(name, (Stub,), {"__module__": module, "_stub_of": f"{module}.{name}"})


def _element_size(dtype: torch.dtype) -> int:
    return torch.empty((), dtype=dtype).element_size()


class UnknownStorage(Stub):
    """A tensor whose storage class is outside ``_STORAGE_DTYPES`` (UntypedStorage, quantised, ...): kept as a
    marker so that ``read_state`` can name the entry instead of dropping it silently."""


def _rebuild_tensor(storage, storage_offset, size, stride, *unused):
    if not isinstance(storage, _Storage):
        return UnknownStorage()
    size, stride = tuple(int(s) for s in size), tuple(int(s) for s in stride)
    if len(storage.buf) == 0 or any(s == 0 for s in size):
        return torch.empty(size, dtype=storage.dtype)
    flat = torch.frombuffer(storage.buf, dtype=storage.dtype)          # shares the bytearray (filled later if legacy)
    need = int(storage_offset) + sum((n - 1) * st for n, st in zip(size, stride)) + 1
    if need > flat.numel() or int(storage_offset) < 0 or any(st < 0 for st in stride):
        raise ValueError(f"checkpoint tensor of size {size} / stride {stride} / offset {storage_offset} does not fit "
                         f"its storage of {flat.numel()} elements")
    return flat.as_strided(size, stride, int(storage_offset))


def _rebuild_parameter(data, *unused):
    return data


def _rebuild_from_type_v2(func, new_type, args, state):
    return func(*args) if func in (_rebuild_tensor, _rebuild_parameter) else Stub()


_SAFE_BUILTINS = {"set": set, "frozenset": frozenset, "dict": dict, "list": list, "tuple": tuple, "int": int,
                  "float": float, "complex": complex, "bool": bool, "str": str, "bytes": bytes,
                  "bytearray": bytearray, "slice": slice, "range": range}


class _Unpickler(pickle.Unpickler):
    def __init__(self, file, load_storage):
        super().__init__(file, encoding="utf-8")
        self._load_storage = load_storage
        self.stubbed: set = set()

    def find_class(self, module: str, name: str):
        if module in ("torch._utils",):
            if name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
                return _rebuild_tensor
            if name in ("_rebuild_parameter", "_rebuild_parameter_with_state"):
                return _rebuild_parameter
        if module == "torch._tensor" and name == "_rebuild_from_type_v2":
            return _rebuild_from_type_v2
        if module in ("torch", "torch.storage") and name in _STORAGE_DTYPES:
            return _StorageType(name)
        if module == "torch":
            obj = getattr(torch, name, None)
            if isinstance(obj, torch.dtype) or name == "Size":
                return obj
            if name == "device":
                return lambda *a, **k: "cpu"
        if module == "torch.nn.parameter" and name == "Parameter":
            return _rebuild_parameter
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        if module in ("builtins", "__builtin__") and name in _SAFE_BUILTINS:
            return _SAFE_BUILTINS[name]
        self.stubbed.add(f"{module}.{name}")
        return _stub_class(module, name)

    def persistent_load(self, pid):
        if isinstance(pid, tuple) and pid and _ascii(pid[0]) == "storage":
            return self._load_storage(pid[1:])
        return Stub()                      # "module" ids of very old containers and anything unknown


def _ascii(x) -> str:
    return x.decode("ascii") if isinstance(x, bytes) else x


def _load_zip(path: Path):
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
        pkl = [n for n in names if n.endswith("/data.pkl") or n == "data.pkl"]
        if not pkl:
            raise ValueError(f"{path}: a zip archive without data.pkl (a TorchScript / other archive?)")
        prefix = pkl[0][: -len("data.pkl")]
        order = prefix + "byteorder"
        if order in names and z.read(order).strip() not in (b"little", b""):
            raise ValueError(f"{path}: big-endian checkpoints are not supported")
        cache: Dict[str, _Storage] = {}

        def load_storage(args):
            storage_type, key, _location, numel = args[:4]
            key = _ascii(key)
            if not isinstance(storage_type, _StorageType):
                return Stub()
            if key not in cache:
                raw = z.read(f"{prefix}data/{key}")
                st = _Storage(storage_type.dtype, 0)
                st.buf = bytearray(raw)
                if len(raw) < int(numel) * _element_size(storage_type.dtype):
                    raise ValueError(f"{path}: storage {key} holds {len(raw)} bytes, {numel} elements expected")
                cache[key] = st
            return cache[key]

        up = _Unpickler(io.BytesIO(z.read(pkl[0])), load_storage)
        return up.load(), up.stubbed


def _load_legacy(path: Path):
    with open(path, "rb") as f:
        inert = lambda args: Stub()
        if _Unpickler(f, inert).load() != _MAGIC:
            raise ValueError(f"{path}: not a torch.save file")
        _Unpickler(f, inert).load()                        # protocol version
        _Unpickler(f, inert).load()                        # system info
        storages: Dict[Any, _Storage] = {}

        def load_storage(args):
            storage_type, root_key, _location, numel, view = args[:5]
            if not isinstance(storage_type, _StorageType):
                return Stub()
            if view is not None:
                raise ValueError(f"{path}: storage views of the legacy format are not supported")
            root_key = _ascii(root_key)
            if root_key not in storages:
                storages[root_key] = _Storage(storage_type.dtype, int(numel) * _element_size(storage_type.dtype))
            return storages[root_key]

        up = _Unpickler(f, load_storage)
        obj = up.load()
        keys = _Unpickler(f, inert).load()                 # the order in which the storages follow
        for key in keys:
            key = _ascii(key)
            (numel,) = struct.unpack("<q", f.read(8))
            st = storages.get(key)
            nbytes = numel * _element_size(st.dtype) if st is not None else None
            if st is None or nbytes != len(st.buf):
                raise ValueError(f"{path}: storage {key} does not match the object that refers to it")
            if f.readinto(st.buf) != nbytes:
                raise ValueError(f"{path}: truncated storage {key}")
        return obj, up.stubbed


def load_object(path: Union[str, Path]):
    """The unpickled top-level object of a ``torch.save`` file with every non-tensor class stubbed, and the set of
    globals that were stubbed.  Tensors are CPU tensors."""
    path = Path(path)
    if zipfile.is_zipfile(path):
        return _load_zip(path)
    return _load_legacy(path)


def _is_safetensors(path: Path) -> bool:
    if path.suffix == ".safetensors":
        return True
    try:
        with open(path, "rb") as f:
            head = f.read(9)
        return len(head) == 9 and head[8:9] == b"{" and struct.unpack("<Q", head[:8])[0] < (1 << 32)
    except OSError:
        return False


def pyannote_version(path: Union[str, Path]):
    """``(major, minor)`` of the ``pyannote.audio`` that wrote a PyTorch-Lightning checkpoint (its top-level
    ``"pyannote.audio"["versions"]["pyannote.audio"]`` entry), or None (safetensors, plain state dicts, no such entry).
    Decides how ``StatsPool`` resamples its pooling weights: ``mode="nearest"`` from 3.1 on, ``"linear"`` before."""
    path = Path(path)
    if not path.is_file() or _is_safetensors(path):
        return None
    try:
        obj, _ = load_object(path)
        ver = obj["pyannote.audio"]["versions"]["pyannote.audio"]
        nums = [int("".join(ch for ch in part if ch.isdigit()) or 0) for part in str(ver).split(".")[:2]]
        return (nums[0], nums[1] if len(nums) > 1 else 0)
    except Exception:      # noqa: BLE001 — any shape of "no version recorded"
        return None


def read_state(path: Union[str, Path]) -> Dict[str, torch.Tensor]:
    """``{parameter name: CPU tensor}`` of a checkpoint file (see the module docstring)."""
    path = Path(path)
    if not path.is_file():
        raise FileNotFoundError(f"{path}: no such checkpoint file")
    if _is_safetensors(path):
        from safetensors.torch import load_file
        state = dict(load_file(str(path), device="cpu"))
    else:
        obj, _ = load_object(path)
        if isinstance(obj, dict) and isinstance(obj.get("state_dict"), dict):
            obj = obj["state_dict"]                       # PyTorch-Lightning checkpoint (pyannote.audio)
        if not isinstance(obj, dict):
            raise ValueError(f"{path}: expected a state dict or a checkpoint with a 'state_dict' entry, "
                             f"found {type(obj).__name__}")
        state = {str(k): v for k, v in obj.items() if isinstance(v, torch.Tensor)}
        dropped = [str(k) for k, v in obj.items() if not isinstance(v, torch.Tensor)]
        unknown = [str(k) for k, v in obj.items() if isinstance(v, UnknownStorage)]
        if unknown:
            raise ValueError(f"{path}: tensor(s) {unknown[:5]} use a storage class this reader does not know "
                             f"(known: {sorted(_STORAGE_DTYPES)}); re-save the checkpoint as safetensors")
        if not state:
            raise ValueError(f"{path}: no tensors in the state dict (entries: {dropped[:5]} ...)")
    # a Lightning module keeps its network under `model.`: strip the prefix PER KEY (one extra top-level entry —
    # a loss weight, a metric buffer — must not keep the prefix on all the others), unless that would collide
    stripped = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in state.items()}
    if len(stripped) == len(state) and any(k.startswith("model.") for k in state):
        state = stripped
    # own the memory (frombuffer views keep whole storages alive) and drop aliasing between entries
    return {k: v.clone() for k, v in state.items()}
