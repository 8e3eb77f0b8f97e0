"""-m gpu: DAC codec layers (Interface.encode / decode) vs oracle/dac_oracle.py.
PARITY UNPINNED upstream (no `lac` source/weights): the oracle restates the published DAC design."""
import math

import numpy as np
import pytest
import torch

from oracle import dac_oracle as D

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from vampnet_amd.engine import Engine
    return Engine("cuda:0")


def _audio(B, L, seed=0):
    g = np.random.default_rng(seed)
    t = np.arange(L) / 44100.0
    x = 0.2 * np.sin(2 * np.pi * 220 * t) + 0.1 * np.sin(2 * np.pi * 3000 * t) + 0.05 * g.standard_normal((B, 1, L))
    return torch.from_numpy(x.astype(np.float32))


@pytest.mark.parametrize("cfg,B,frames", [(D.DAC_TINY_CFG, 2, 50), (D.DAC_TINY_CFG, 1, 7), (D.DAC_DEFAULT_CFG, 2, 12)])
def test_encode_decode_vs_oracle(eng, cfg, B, frames):
    from vampnet_amd.codec import DacCodec
    sd = D.synth_dac_state_dict(cfg, 0)
    codec = DacCodec(sd, cfg, engine=eng)
    hop = D.hop_length(cfg)
    assert codec.hop_length == hop
    audio = _audio(B, hop * frames - 3)
    padded, length = codec.preprocess(audio, 44100)
    assert padded.shape[-1] == hop * frames and length == hop * frames - 3
    assert torch.equal(padded, D.preprocess(cfg, audio))
    # ---- encoder + RVQ
    with torch.inference_mode():
        z_ref = D.encoder(sd, cfg, padded)                                   # (B, L, T)
        zq_ref, codes_ref = D.rvq_encode(sd, cfg, z_ref)
    out = codec.encode(padded, 44100)
    z = out["z"].cpu().permute(0, 2, 1)
    scale = z_ref.abs().max().item()
    err = (z - z_ref).abs().max().item()
    print(f"encoder latents: max |d| = {err:.3e} (scale {scale:.2f})")
    assert err <= 2e-5 * max(scale, 1.0)
    codes = out["codes"].cpu()
    assert codes.shape == codes_ref.shape == (B, cfg["n_codebooks"], frames)
    agree = (codes == codes_ref).float().mean().item()
    print(f"codes agreement {agree:.4f}")
    # a flipped code re-routes every later residual level of that frame: require the FIRST level to agree except on
    # near-ties, and overall agreement to stay high
    assert (codes[:, 0] == codes_ref[:, 0]).float().mean().item() >= 0.98
    assert agree >= 0.9
    # ---- decoder on the ORACLE's codes (decouples it from encoder near-ties)
    with torch.inference_mode():
        audio_ref = D.decode(sd, cfg, codes_ref)
    got = codec.decode_codes(codes_ref).cpu()
    assert got.shape == audio_ref.shape == (B, 1, hop * frames)
    rms = (got - audio_ref).pow(2).mean().sqrt().item()
    print(f"decoded audio: rms err = {rms:.3e}, max err = {(got - audio_ref).abs().max().item():.3e}")
    assert rms <= 1e-5                                                       # north star: within 1e-4 RMS


def test_interface_encode_decode_roundtrip(eng):
    """Interface.encode / decode over the codec object (interface.py:203-224): shapes, MASK -> 0, AudioSignal out."""
    from oracle import weights as W
    from tests.gpu_common import model_kwargs
    from vampnet_amd.codec import AudioSignal, DacCodec
    from vampnet_amd.interface import Interface
    cfg = dict(D.DAC_TINY_CFG, n_codebooks=14)
    sd = D.synth_dac_state_dict(cfg, 1)
    codec = DacCodec(sd, cfg, engine=eng)
    itf = Interface.from_state_dicts(codec, W.synth_state_dict(W.TINY_COARSE_DIMS, 0), model_kwargs(W.TINY_COARSE_DIMS),
                                     W.synth_state_dict(W.TINY_C2F_DIMS, 1), model_kwargs(W.TINY_C2F_DIMS), max_batch=2,
                                     coarse_chunk_size_s=0.005, coarse2fine_chunk_size_s=0.003)   # hop 8: 28 / 17 tokens
    sig = AudioSignal(_audio(1, 8 * 40 + 5)[0], 44100)
    z = itf.encode(sig)
    assert z.shape == (1, 14, 41) and z.dtype == torch.int64 and z.min() >= 0 and z.max() < 1024
    zm = z.clone()
    zm[:, :, 3] = 1024                                                        # MASK tokens are replaced by 0 (transformer.py:669)
    a = itf.decode(zm)
    z0 = z.clone()
    z0[:, :, 3] = 0
    b = itf.decode(z0)
    assert isinstance(a, AudioSignal) and a.sample_rate == 44100 and a.samples.shape == (1, 1, 41 * 8)
    assert torch.equal(a.samples, b.samples)
    # the codebooks the sampling loop embeds are the codec's own (layers.py:145)
    assert torch.equal(itf._codebooks[3], sd["quantizer.quantizers.3.codebook.weight"])
    mask = itf.build_mask(z)
    out = itf.vamp(z, mask, batch_size=2, seed=0, _sampling_steps=2)
    assert out.shape == (2, 14, 41)
    assert itf.decode(out).samples.shape == (2, 1, 41 * 8)


def test_vamp_service_end_to_end(eng):
    """The reference's `/vamp` endpoint body (app.py:120-263) over the HIP Interface: int16 PCM in, two loudness-matched
    audios out, and the tokens behind them equal to driving the Interface by hand with the same seed."""
    import numpy as np
    from oracle import weights as W
    from tests.gpu_common import model_kwargs
    from vampnet_amd import serve
    from vampnet_amd.codec import AudioSignal, DacCodec
    from vampnet_amd.engine import seed_all
    from vampnet_amd.interface import Interface
    cfg = dict(D.DAC_TINY_CFG, n_codebooks=14)
    codec = DacCodec(D.synth_dac_state_dict(cfg, 1), cfg, engine=eng)
    itf = Interface.from_state_dicts(codec, W.synth_state_dict(W.TINY_COARSE_DIMS, 0), model_kwargs(W.TINY_COARSE_DIMS),
                                     W.synth_state_dict(W.TINY_C2F_DIMS, 1), model_kwargs(W.TINY_C2F_DIMS), max_batch=2,
                                     coarse_chunk_size_s=0.05, coarse2fine_chunk_size_s=0.02)     # hop 8: 276 / 111 tokens
    n = 4413                                                                                     # 0.1 s at 44.1 kHz, not a hop multiple
    t = np.arange(n) / 44100.0
    pcm = ((0.3 * np.sin(2 * np.pi * 220 * t) + 0.1 * np.sin(2 * np.pi * 3000 * t)) * 32767).astype(np.int16)
    svc = serve.VampService(itf, chunk_size_s=0.05)
    req = dict(input_audio=(44100, pcm), sampletemp=1.0, top_p=0.0, periodic_p=3, dropout=0.0, stretch_factor=1,
               onset_mask_width=0, typical_filtering=True, typical_mass=0.15, typical_min_tokens=64, seed=5,
               model_choice="default", n_mask_codebooks=3, pitch_shift_amt=0, sample_cutoff=1.0, sampling_steps=3,
               beat_mask_ms=0, num_feedback_steps=1)
    (sr, a0), (_, a1) = svc.api_vamp(*[req[k] for k in serve.VAMP_ARG_ORDER])
    assert sr == 44100 and a0.shape == a1.shape == (n - n % 8 + (8 if n % 8 else 0),) and a0.dtype == np.float32
    want = float(serve._to_signal(req["input_audio"]).loudness()[0])
    for a in (a0, a1):
        assert np.isfinite(a).all() and abs(float(AudioSignal(a, sr).loudness()[0]) - want) < 1e-3
    assert not np.array_equal(a0, a1)                                   # two samples of the same prompt
    # by hand, in the order of _vamp_internal
    seed_all(5)
    sig = itf._preprocess(serve._to_signal(req["input_audio"]).to_mono())
    codes = itf.encode(sig)
    mask = itf.build_mask(codes, sig=sig, periodic_prompt=3, onset_mask_width=0, _dropout=0.0, upper_codebook_mask=3)
    assert torch.equal(mask, svc.last_mask)
    z = itf.vamp(codes, mask, batch_size=2, feedback_steps=1, _sampling_steps=3, temperature=1.0, top_p=None, seed=5,
                 sample_cutoff=1.0)
    hand = itf.decode(z).normalize(torch.tensor([want, want]))
    assert torch.equal(torch.from_numpy(a0), hand.samples[0, 0].cpu()) and torch.equal(torch.from_numpy(a1), hand.samples[1, 0].cpu())
