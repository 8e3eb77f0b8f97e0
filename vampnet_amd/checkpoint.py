"""Reader for the reference's checkpoint files (audiotools `BaseModel.load`, called from interface.py:27-50 `_load_model`
and interface.py:70 `DAC.load`; semantics from the public audiotools sources, SURVEY.md App. C [UNVERIFIED-DEP]).

`BaseModel.load(path)` tries two formats, in this order:
  1. a **torch.package** archive holding the pickled module object as `<package_name>/<package_name>.pth` (+ optionally
     `<package_name>.metadata`) — what `save(path, package=True)` writes;
  2. `torch.load(path, "cpu")` of `{"state_dict": ..., "metadata": {"kwargs": <constructor kwargs>, ...}}`.
LoRA files (`lora.pth`, interface.py:45) are a bare `{name: tensor}` dict.

Safety: a checkpoint path can come from a request (`serve.py` model choice).  Format 2 is read with
`weights_only=True` (tensors + primitives only — the payload never needs more).  A torch.package archive EXECUTES the code
packed inside it, and a dict checkpoint whose metadata holds non-primitive objects needs full unpickling: both are refused
unless the caller opts in with `trusted=True` or VN_TRUST_CHECKPOINTS=1."""
import os
import zipfile
from pathlib import Path

import torch


def _trusted(flag):
    return bool(flag) or os.environ.get("VN_TRUST_CHECKPOINTS") == "1"


def is_torch_package(path) -> bool:
    """torch.save also writes zip files; a torch.package archive is the one WITHOUT `<root>/data.pkl` and with the
    `.data/extern_modules` record its exporter always writes."""
    if not zipfile.is_zipfile(path):
        return False
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
    if any(n.count("/") == 1 and n.endswith("/data.pkl") for n in names):
        return False
    return any(n.endswith(".data/extern_modules") for n in names)


def _attr_kwargs(module, keys):
    return {k: getattr(module, k) for k in keys if isinstance(getattr(module, k, None), (int, float, bool, str))}


def load_model_checkpoint(path, *, package_name: str = "VampNet", kwarg_keys=(), trusted: bool = False):
    """-> (state_dict, constructor kwargs).  `package_name` is the class name audiotools packs under ("VampNet", "DAC" /
    "LAC"); `kwarg_keys` are read from the module's attributes when a packaged model carries no metadata."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"checkpoint {path} does not exist")
    if is_torch_package(path):
        if not _trusted(trusted):
            raise PermissionError(f"{path} is a torch.package archive: loading it executes the code packed inside. "
                                  "Pass trusted=True or set VN_TRUST_CHECKPOINTS=1 for files you trust.")
        from torch import package
        imp = package.PackageImporter(str(path))
        names = [package_name] + [n for n in ("VampNet", "DAC", "LAC") if n != package_name]
        model, err = None, None
        for n in names:
            try:
                model = imp.load_pickle(n, f"{n}.pth", "cpu")
                package_name = n
                break
            except Exception as e:          # wrong package name: try the next
                err = e
        if model is None:
            raise ValueError(f"{path}: no {names} model inside the torch.package archive ({err})")
        try:
            meta = imp.load_pickle(package_name, f"{package_name}.metadata")
        except Exception:
            meta = getattr(model, "metadata", None) or {}
        kwargs = dict((meta or {}).get("kwargs", {})) or _attr_kwargs(model, kwarg_keys)
        return {k: v.detach().cpu() for k, v in model.state_dict().items()}, kwargs
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        if not _trusted(trusted):
            raise PermissionError(f"{path} cannot be read with weights_only=True ({type(e).__name__}: {e}). If it is a "
                                  "checkpoint you trust whose metadata holds non-tensor objects, pass trusted=True or set "
                                  "VN_TRUST_CHECKPOINTS=1.") from e
        ckpt = torch.load(path, map_location="cpu", weights_only=False)
    if not isinstance(ckpt, dict) or "state_dict" not in ckpt:
        raise ValueError(f"{path}: not an audiotools-format checkpoint (no 'state_dict')")
    meta = ckpt.get("metadata") or {}
    return ckpt["state_dict"], dict(meta.get("kwargs") or {})


def load_tensor_dict(path, *, trusted: bool = False) -> dict:
    """A bare {name: tensor} file (lora.pth = loralib.lora_state_dict, optimizer.pth, scheduler.pth)."""
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"{path} does not exist")
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        if not _trusted(trusted):
            raise PermissionError(f"{path} cannot be read with weights_only=True ({type(e).__name__}: {e}); pass trusted=True "
                                  "or set VN_TRUST_CHECKPOINTS=1 for files you trust") from e
        return torch.load(path, map_location="cpu", weights_only=False)


LORA_SCALING = 1.0 / 8.0      # loralib: lora_alpha (1) / r (8), transformer.py:22 LORA_R


def merge_lora_state_dict(sd: dict) -> dict:
    """What loralib's `Linear.eval()` computes once for every adapter pair: W_eff = W + (lora_B @ lora_A) * alpha / r
    (SURVEY.md App. C).  Returns a state_dict WITHOUT lora_A / lora_B keys whose weights are the merged ones — the form both
    the engine's packer and a plain `nn.Linear` model (the reference under the import shim) load."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".lora_A") or k.endswith(".lora_B"):
            continue
        if k.endswith(".weight"):
            stem = k[:-len(".weight")]
            a, b = sd.get(stem + ".lora_A"), sd.get(stem + ".lora_B")
            if a is not None and b is not None:
                v = v.float() + (b.float() @ a.float()) * LORA_SCALING
        out[k] = v
    return out
