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


def lora_scaling(sd: dict, lora_alpha: float = 1.0) -> float:
    """loralib.Linear.scaling = lora_alpha / r.  r is read from the checkpoint's adapters (lora_A is (r, in)); lora_alpha is not
    stored in a state_dict — loralib's constructor default is 1 and the reference never passes it (transformer.py:67-68, 109-114:
    `lora.Linear(..., r=LORA_R)`).  No adapters -> the reference's 1 / LORA_R."""
    ranks = {int(v.shape[0]) for k, v in sd.items() if k.endswith(".lora_A")}
    if not ranks:
        return LORA_SCALING
    if len(ranks) != 1:
        raise ValueError(f"LoRA adapters of different ranks {sorted(ranks)} in one checkpoint")
    return float(lora_alpha) / ranks.pop()


def merge_lora_state_dict(sd: dict, lora_alpha: float = 1.0) -> dict:
    """What loralib's `Linear.train(False)` computes once for every adapter pair (fan_in_fan_out=False, the default the reference
    uses): W_eff = W + (lora_B @ lora_A) * lora_alpha / r  (SURVEY.md App. C), with r taken from the adapters themselves.  Returns
    a state_dict WITHOUT lora_A / lora_B keys whose weights are the merged ones — the form both the engine's packer and a plain
    `nn.Linear` model (the reference under the import shim) load."""
    scaling = lora_scaling(sd, lora_alpha)
    out = {}
    for k, v in sd.items():
        if k.endswith(".lora_A") or k.endswith(".lora_B"):
            continue
        if k.endswith(".weight"):
            stem = k[:-len(".weight")]
            a, b = sd.get(stem + ".lora_A"), sd.get(stem + ".lora_B")
            if a is not None and b is not None:
                if tuple(b.shape) != (v.shape[0], a.shape[0]) or a.shape[1] != v.shape[1]:
                    raise ValueError(f"{stem}: lora_B {tuple(b.shape)} @ lora_A {tuple(a.shape)} does not match weight {tuple(v.shape)}")
                v = v.float() + (b.float() @ a.float()) * scaling
        out[k] = v
    return out


_VAMPNET_KEYS = ("n_heads", "n_layers", "n_codebooks", "n_conditioning_codebooks", "latent_dim", "embedding_dim", "vocab_size")
_VAMPNET_DEFAULTS = dict(n_heads=20, n_layers=16, n_codebooks=9, n_conditioning_codebooks=0, latent_dim=8, embedding_dim=1280,
                         vocab_size=1024)                        # VampNet.__init__ defaults, transformer.py:536-545
_LORA_PARENTS = ("self_attn.w_qs", "self_attn.w_vs", "self_attn.fc", "feed_forward.w_1", "feed_forward.w_2")


def validate_vampnet_state_dict(sd: dict, kwargs: dict = None) -> dict:
    """Shape check of a VampNet state_dict against its constructor kwargs (metadata.kwargs of the checkpoint) BEFORE anything is
    packed for the GPU: every tensor the engine consumes must be present with the shape the reference's modules give it
    (transformer.py:55-58 RMSNorm, :81-84 FeedForward, :109-121 attention, :536-616 VampNet; layers.py:134-163 CodebookEmbedding),
    adapters must be consistent pairs.  Raises ValueError naming the first mismatch; returns a summary
    {"n_tensors", "n_lora_pairs", "lora_rank", "kwargs"} for logs (scripts/parity_real_ckpt.py prints it)."""
    kw = dict(_VAMPNET_DEFAULTS, **{k: v for k, v in (kwargs or {}).items() if k in _VAMPNET_KEYS})
    H, L, C, Cc = int(kw["n_heads"]), int(kw["n_layers"]), int(kw["n_codebooks"]), int(kw["n_conditioning_codebooks"])
    ld, D, V = int(kw["latent_dim"]), int(kw["embedding_dim"]), int(kw["vocab_size"])
    if D != 64 * H:
        raise ValueError(f"embedding_dim {D} / n_heads {H}: the engine (and every published checkpoint) has 64-wide heads")

    def need(key, shape):
        if key not in sd:
            raise ValueError(f"VampNet checkpoint lacks {key} (kwargs {kw})")
        if tuple(sd[key].shape) != tuple(shape):
            raise ValueError(f"{key} is {tuple(sd[key].shape)}, kwargs {kw} imply {tuple(shape)}")

    need("embedding.special.MASK", (C, ld))
    need("embedding.out_proj.weight", (D, C * ld, 1))
    need("embedding.out_proj.bias", (D,))
    need("transformer.layers.0.self_attn.relative_attention_bias.weight", (32, H))
    need("transformer.norm.weight", (D,))
    Cp = C - Cc
    if "classifier.layers.0.weight_v" in sd:
        need("classifier.layers.0.weight_v", (V * Cp, D, 1))
        need("classifier.layers.0.weight_g", (V * Cp, 1, 1))
    else:
        need("classifier.layers.0.weight", (V * Cp, D, 1))
    need("classifier.layers.0.bias", (V * Cp,))
    ranks = set()
    for l in range(L):
        p = f"transformer.layers.{l}."
        need(p + "norm_1.weight", (D,))
        need(p + "norm_3.weight", (D,))
        for name, shape in (("self_attn.w_qs", (D, D)), ("self_attn.w_ks", (D, D)), ("self_attn.w_vs", (D, D)),
                            ("self_attn.fc", (D, D)), ("feed_forward.w_1", (4 * D, D)), ("feed_forward.w_2", (D, 2 * D))):
            need(p + name + ".weight", shape)
            a, b = sd.get(p + name + ".lora_A"), sd.get(p + name + ".lora_B")
            if (a is None) != (b is None):
                raise ValueError(f"{p}{name}: only one of lora_A / lora_B present")
            if a is not None:
                if name not in _LORA_PARENTS:
                    raise ValueError(f"{p}{name} carries adapters; the reference only adapts {_LORA_PARENTS}")
                if a.shape[1] != shape[1] or tuple(b.shape) != (shape[0], a.shape[0]):
                    raise ValueError(f"{p}{name}: lora_A {tuple(a.shape)} / lora_B {tuple(b.shape)} do not fit weight {shape}")
                ranks.add(int(a.shape[0]))
    if f"transformer.layers.{L}.norm_1.weight" in sd:
        raise ValueError(f"checkpoint holds more than n_layers={L} transformer layers")
    if len(ranks) > 1:
        raise ValueError(f"adapters of different ranks {sorted(ranks)}")
    return {"n_tensors": len(sd), "n_lora_pairs": sum(k.endswith(".lora_A") for k in sd), "lora_rank": ranks.pop() if ranks else None,
            "kwargs": kw}
