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
