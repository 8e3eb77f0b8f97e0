"""Wire compatibility of the `vamp` endpoint adapter (vampnet_amd/serve.py) with the reference's app and unloop client:
argument order, renames, the seed / top_p / model-choice rules of `_vamp_internal` (app.py:120-263).  CPU only — the
interface is a recording stand-in; the end-to-end GPU run is tests/test_gpu_codec.py::test_vamp_service_end_to_end."""
import inspect
import os
import re

import numpy as np
import pytest
import torch

from vampnet_amd import serve
from vampnet_amd.codec import AudioSignal

REF_APP = "/root/reference/app.py"
REF_CLIENT = "/root/reference/unloop/client.py"


class _Recorder:
    """Records what the service asks of Interface; returns shaped dummies."""
    device = "cpu"

    def __init__(self):
        self.calls = []

    def _preprocess(self, sig):
        self.calls.append(("_preprocess", sig.sample_rate, tuple(sig.samples.shape)))
        return sig

    def reload(self, coarse_ckpt=None, c2f_ckpt=None):
        self.calls.append(("reload", coarse_ckpt, c2f_ckpt))

    def encode(self, sig):
        self.calls.append(("encode",))
        return torch.zeros(1, 14, 20, dtype=torch.int64)

    def build_mask(self, z, **kw):
        self.calls.append(("build_mask", kw))
        self.rand_after_seed = float(torch.rand(1))            # the global generator state the mask draws would see
        return torch.ones_like(z)

    def set_chunk_size(self, s):
        self.calls.append(("set_chunk_size", s))

    def vamp(self, codes, mask, **kw):
        self.calls.append(("vamp", kw))
        B = kw["batch_size"]
        return codes.expand(B, -1, -1).clone(), mask.expand(B, -1, -1).clone()

    def decode(self, z):
        self.calls.append(("decode", tuple(z.shape)))
        g = torch.Generator().manual_seed(0)
        return AudioSignal(0.05 * torch.randn(z.shape[0], 1, 44100, generator=g), 44100)


def _request(**over):
    t = np.arange(44100) / 44100.0
    pcm = (0.3 * np.sin(2 * np.pi * 220 * t) * 32767).astype(np.int16)
    r = dict(input_audio=(44100, pcm), sampletemp=0.9, top_p=0.0, periodic_p=5, dropout=0.1, stretch_factor=1,
             onset_mask_width=0, typical_filtering=True, typical_mass=0.2, typical_min_tokens=32, seed=11,
             model_choice="default", n_mask_codebooks=4, pitch_shift_amt=0, sample_cutoff=0.9, sampling_steps=16,
             beat_mask_ms=0, num_feedback_steps=2)
    r.update(over)
    return r


def test_positional_order_is_the_reference_inputs_list():
    assert len(serve.VAMP_ARG_ORDER) == 18 and len(serve.UNLOOP_OSC_ORDER) == 18
    assert serve.VAMP_ARG_ORDER[0] == "input_audio" and serve.VAMP_ARG_ORDER[-1] == "num_feedback_steps"
    rec = _Recorder()
    svc = serve.VampService(rec)
    r = _request()
    a = svc.api_vamp(*[r[k] for k in serve.VAMP_ARG_ORDER])
    assert len(a) == 2 and a[0][0] == 44100 and a[0][1].shape == (44100,)
    v = svc.vamp(**r)
    assert len(v) == 3                                         # UI variant: two audios + the mask (image in the reference)
    with pytest.raises(TypeError):
        svc.api_vamp(*([0] * 19))
    with pytest.raises(TypeError):
        svc.api_vamp(**dict(r, bogus=1))
    with pytest.raises(TypeError):
        svc.api_vamp(**{k: r[k] for k in serve.VAMP_ARG_ORDER[:-1]})


def test_request_mapping_follows_vamp_internal():
    rec = _Recorder()
    svc = serve.VampService(rec, models={"opera": ("/m/opera/coarse.pth", "/m/opera/c2f.pth")})
    r = _request(model_choice="opera")
    svc.api_vamp(**r)
    names = [c[0] for c in rec.calls]
    assert names == ["_preprocess", "reload", "encode", "build_mask", "set_chunk_size", "vamp", "decode"]   # app.py:176-246
    assert rec.calls[1] == ("reload", "/m/opera/coarse.pth", "/m/opera/c2f.pth")
    bm = rec.calls[3][1]
    assert bm["periodic_prompt"] == 5 and bm["_dropout"] == 0.1 and bm["upper_codebook_mask"] == 4 \
        and bm["onset_mask_width"] == 0 and bm["sig"] is not None
    assert rec.calls[4] == ("set_chunk_size", 10.0)
    kw = rec.calls[5][1]
    assert kw == dict(batch_size=2, feedback_steps=2, _sampling_steps=16, time_stretch_factor=1, return_mask=True,
                      temperature=0.9, typical_filtering=True, typical_mass=0.2, typical_min_tokens=32, top_p=None,
                      seed=11, sample_cutoff=0.9)                              # top_p <= 0 -> None (app.py:221-226)
    # seed > 0 seeds torch / numpy / random before the mask is built (app.py:164-167)
    torch.manual_seed(11)
    assert rec.rand_after_seed == float(torch.rand(1))
    assert svc.last_seed == 11
    # top_p > 0 passes through
    rec.calls.clear()
    svc.api_vamp(**_request(top_p=0.8))
    assert rec.calls[-2][1]["top_p"] == 0.8
    # seed <= 0: a fresh 32-bit seed is drawn and used for both the mask draws and generate()
    svc.api_vamp(**_request(seed=0))
    assert 0 <= svc.last_seed < 2 ** 32 and rec.calls[-2][1]["seed"] == svc.last_seed
    with pytest.raises(KeyError):
        svc.api_vamp(**_request(model_choice="no-such-model"))
    for bad in (dict(pitch_shift_amt=2), dict(beat_mask_ms=100)):
        with pytest.raises(NotImplementedError):
            svc.api_vamp(**_request(**bad))
    # with a tracker the beat mask is AND-ed in and the upper codebooks re-masked (app.py:206-217)
    rec.beat_tracker = object()
    rec.make_beat_mask = lambda sig, after_beat_s: (rec.calls.append(("make_beat_mask", after_beat_s)),
                                                    torch.zeros(1, 14, 20, dtype=torch.long))[1]
    rec.calls.clear()
    svc.api_vamp(**_request(beat_mask_ms=100, n_mask_codebooks=4))
    assert ("make_beat_mask", 0.1) in rec.calls
    assert bool((svc.last_mask[:, :4] == 0).all()) and bool((svc.last_mask[:, 4:] == 1).all())
    with pytest.raises(ValueError):
        svc.api_vamp(**_request(input_audio=None))


def test_output_is_renormalised_to_the_input_loudness():
    rec = _Recorder()
    svc = serve.VampService(rec)
    r = _request()
    (sr, a0), (_, a1) = svc.api_vamp(**r)
    want = float(serve._to_signal(r["input_audio"]).loudness()[0])
    for a in (a0, a1):
        assert abs(float(AudioSignal(a, sr).loudness()[0]) - want) < 1e-3        # app.py:247 sig.normalize(loudness)
    # integer PCM is divided by the dtype's max (app.py:173); stereo (samples, channels) is mixed down
    t = np.arange(22050) / 22050.0
    st = np.stack([np.sin(2 * np.pi * 300 * t), np.zeros_like(t)], axis=1)
    sig = serve._to_signal((22050, (st * 32767).astype(np.int16)))
    assert sig.samples.shape == (1, 2, 22050) and abs(float(sig.samples.abs().max()) - 1.0) < 1e-4
    assert sig.to_mono().samples.shape == (1, 1, 22050)


def test_unloop_osc_fields_map_to_the_endpoint_keywords(tmp_path):
    osc = ["q7", "max", str(tmp_path / "loop.wav"), "opera", 7, 0.0, 42, 500.0, 1, 0.15, 64, 3, 0, 24, 1.0, 0.0, 0.0, 1]
    req = serve.unloop_to_request(osc)
    assert set(req) == set(serve.VAMP_ARG_ORDER)
    assert req["sampletemp"] == 1.0 and req["n_mask_codebooks"] == 3 and req["typical_filtering"] is True \
        and req["stretch_factor"] == 1 and req["pitch_shift_amt"] == 0 and req["sample_cutoff"] == 1.0 \
        and req["beat_mask_ms"] == 0 and isinstance(req["beat_mask_ms"], int) and req["seed"] == 42
    with pytest.raises(ValueError):
        serve.unloop_to_request(osc[:-1])
    # end to end: the recording is re-labelled 48 kHz and cropped to the loop length before it is sent
    AudioSignal(0.2 * torch.randn(1, 1, 44100, generator=torch.Generator().manual_seed(1)), 44100).write(tmp_path / "loop.wav")
    rec = _Recorder()
    svc = serve.VampService(rec, models={"opera": ("c.pth", None)})
    out = svc.unloop(osc)
    assert len(out) == 2
    assert rec.calls[0] == ("_preprocess", 48000, (1, 1, 24000))


@pytest.mark.skipif(not os.path.exists(REF_APP), reason="reference tree not mounted")
def test_orders_match_the_reference_sources():
    """Parse the reference's own lists: app.py `_inputs = [...]`, the api_vamp signature, and the OSC unpacking."""
    src = open(REF_APP).read()
    block = src[src.index("_inputs = ["):]
    block = block[:block.index("]")]
    names = re.findall(r"^\s*([a-z_]+),?\s*$", block, flags=re.M)
    assert tuple(names) == serve.VAMP_ARG_ORDER
    sig = src[src.index("def api_vamp("):]
    sig = sig[:sig.index("):")]
    assert tuple(re.findall(r"([a-z_]+)\s*(?:,|$)", sig.split("(", 1)[1])) == serve.VAMP_ARG_ORDER
    csrc = open(REF_CLIENT).read()
    csrc = csrc[csrc.index("def vampnet_process("):]
    csrc = csrc[:csrc.index("job.result()")]
    got = re.findall(r"^\s*([a-z_]+)\s*=\s*(?:Path\()?args\[(\d+)\]", csrc, flags=re.M)
    order = [n for n, _ in sorted(got, key=lambda p: int(p[1]))]
    assert tuple(order) == serve.UNLOOP_OSC_ORDER
    # every keyword the client submits is an endpoint argument
    sub = csrc[csrc.index("client.submit("):]
    sub = sub[:sub.index("api_name")]
    assert set(re.findall(r"^\s*([a-z_]+)=", sub, flags=re.M)) == set(serve.VAMP_ARG_ORDER)
