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


@pytest.mark.parametrize("precision", ["f16x2", "bf16x3", "f32"])
@pytest.mark.parametrize("cfg,B,frames", [(D.DAC_TINY_CFG, 2, 50), (D.DAC_TINY_CFG, 1, 7), (D.DAC_DEFAULT_CFG, 2, 12)])
def test_encode_decode_vs_oracle(eng, cfg, B, frames, precision):
    """all codec pipes at the SAME bars: "f32" = every convolution on the fp32-input MFMA kernel; "bf16x3" = the MFMA-bound
    convolutions (>= 96 output channels, K >= 256) as six bf16-MFMA products of exact operand splits (gemm_x3.hip's implicit-GEMM
    mode), "f16x2" (default) = the same convolutions as three fp16-MFMA products of two-plane splits; the rest on the fp32
    kernels, activations handed over as split planes / fp32 as each consumer reads them"""
    from vampnet_amd.codec import DacCodec
    sd = D.synth_dac_state_dict(cfg, 0)
    codec = DacCodec(sd, cfg, engine=eng, precision=precision)
    n_x3 = sum(1 for blk in codec.enc["blocks"] for r in blk["res"] for c in (r["c7"], r["c1"]) if "w16" in c)
    n_x3 += sum(1 for blk in codec.dec["blocks"] for r in blk["res"] for c in (r["c7"], r["c1"]) if "w16" in c)
    if precision == "f32":
        assert n_x3 == 0 and "w16" not in codec.dec["in"]
    elif cfg is D.DAC_DEFAULT_CFG:
        assert n_x3 > 0 and "w16" in codec.dec["in"]
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


CONV_CASES = [  # (B, T_in, C_in, C_out, taps, in_stride, dil, pad, note)
    (2, 300, 128, 128, 7, 1, 1, 3, "k7"),
    (2, 300, 128, 128, 7, 1, 9, 27, "k7 dilation 9: 27 zero rows either side"),
    (1, 257, 256, 96, 7, 1, 3, 9, "96 output channels (a 128-wide tile with masked columns)"),
    (2, 240, 128, 256, 8, 4, 1, 2, "strided down-sampling k = 2 s, s = 4"),
    (3, 100, 512, 192, 1, 1, 1, 0, "k = 1 tail"),
    (1, 33, 1024, 1024, 3, 1, 1, 1, "final encoder conv"),
    (2, 301, 192, 192, 7, 1, 3, 9, "192 channels, k7 dilation 3: the channels-on-rows form (CONVT), ragged positions"),
    (3, 130, 96, 96, 7, 1, 9, 27, "96 channels, k7 dilation 9: CONVT on the 96-row k-split tile"),
    (1, 64, 384, 192, 2, 1, 1, 0, "two taps, 384 -> 192 (the shape of a transposed convolution's phase)"),
]


@pytest.mark.parametrize("B,T,cin,cout,taps,stride,dil,pad,note", CONV_CASES)
def test_conv1d_bf16x3_vs_torch_and_f32_kernel(eng, B, T, cin, cout, taps, stride, dil, pad, note):
    """vn_conv1d_bf16x3 (implicit GEMM on the bf16 matrix cores, exact 3-way splits) against torch's conv1d in float64 and against
    vn_conv1d_f32 on the same inputs: same fp32-grade tolerance; raw, snake-fp32 and snake-planes outputs, bias + residual."""
    import ctypes as C
    from vampnet_amd.codec import DacCodec
    g = torch.Generator().manual_seed(hash((B, T, cin, cout, taps)) % 1000)
    x = torch.randn(B, T, cin, generator=g)
    w = torch.randn(cout, taps, cin, generator=g) / math.sqrt(taps * cin)
    bias, alpha = torch.randn(cout, generator=g), torch.rand(cout, generator=g) + 0.5
    T_out = (T + 2 * pad - dil * (taps - 1) - 1) // stride + 1
    resid = torch.randn(B, T_out, cout, generator=g)
    ref = torch.nn.functional.conv1d(x.double().permute(0, 2, 1), w.double().permute(0, 2, 1), bias.double(), stride=stride,
                                     padding=pad, dilation=dil).permute(0, 2, 1) + resid.double()
    ref_s = ref + torch.sin(alpha.double() * ref) ** 2 / (alpha.double() + 1e-9)
    lib, dev = eng.lib, "cuda"
    xd, wd, bd, ad, rd = (t.to(dev).contiguous() for t in (x, w, bias, alpha, resid))
    x16 = eng.split3(xd.reshape(-1, cin))
    codec = DacCodec.__new__(DacCodec)
    codec.engine, codec.lib, codec.device, codec.precision = eng, lib, eng.device, "bf16x3"
    w16 = codec._tile_planes(wd.reshape(cout, -1))
    y, y2 = torch.empty(B, T_out, cout, device=dev), torch.empty(B, T_out, cout, device=dev)
    y216 = torch.empty(3, B * T_out, cout, device=dev, dtype=torch.bfloat16)
    eng.check(lib.vn_conv1d_bf16x3(eng.handle, x16.data_ptr(), B * T * cin, w16.data_ptr(), bd.data_ptr(), rd.data_ptr(), ad.data_ptr(),
                                   y.data_ptr(), y2.data_ptr(), y216.data_ptr(), B * T_out * cout, B, T, T_out, T_out, cin, cout, taps,
                                   stride, dil, pad, 1, 0, 0, eng.stream()), "vn_conv1d_bf16x3")
    yf, y2f = torch.empty_like(y), torch.empty_like(y2)
    y216f = torch.empty_like(y216)
    eng.check(lib.vn_conv1d_f32(eng.handle, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr(), ad.data_ptr(), yf.data_ptr(),
                                y2f.data_ptr(), y216f.data_ptr(), B * T_out * cout, B, T, T_out, T_out, cin, cout, taps, stride, dil, pad,
                                1, 0, 0, eng.stream()), "vn_conv1d_f32")
    torch.cuda.synchronize()
    absdot = torch.nn.functional.conv1d(x.abs().double().permute(0, 2, 1), w.abs().double().permute(0, 2, 1), bias.abs().double(),
                                        stride=stride, padding=pad, dilation=dil).permute(0, 2, 1) + resid.abs().double()
    tol = (2e-6 * absdot + 1e-6)
    for name, got in (("bf16x3", y), ("f32", yf)):
        err = (got.cpu().double() - ref).abs()
        print(f"{note}: {name} raw max err {err.max().item():.3e}")
        assert bool((err <= tol).all()), name
    for name, got in (("bf16x3", y2), ("f32", y2f)):
        assert (got.cpu().double() - ref_s).abs().max().item() <= 2e-5, name
    for name, planes, dense in (("bf16x3", y216, y2), ("f32", y216f, y2f)):
        p = planes.float()
        assert torch.equal((p[0] + p[1] + p[2]).reshape(B, T_out, cout), dense), name + ": the planes must sum to the fp32 snake output exactly"
    # ---- the same convolution on f16x2 operands (three fp16-MFMA products): same tolerance; its planes, and the fp16 planes the fp32
    # kernel writes for a consumer on that pipe (negative plane stride), reproduce the dense snake output to 2^-22
    x2, w2 = eng.split2h(xd.reshape(-1, cin)), eng.split2h(wd.reshape(cout, -1).contiguous(), tiled=True)
    yh, y2h = torch.empty_like(y), torch.empty_like(y2)
    y2h16 = torch.empty(2, B * T_out, cout, device=dev, dtype=torch.float16)
    eng.check(lib.vn_conv1d_f16x2(eng.handle, x2.data_ptr(), B * T * cin, w2.data_ptr(), bd.data_ptr(), rd.data_ptr(), ad.data_ptr(),
                                  yh.data_ptr(), y2h.data_ptr(), y2h16.data_ptr(), B * T_out * cout, B, T, T_out, T_out, cin, cout, taps,
                                  stride, dil, pad, 1, 0, 0, eng.stream()), "vn_conv1d_f16x2")
    y2f16h = torch.empty_like(y2h16)
    eng.check(lib.vn_conv1d_f32(eng.handle, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr(), ad.data_ptr(), None, y2f.data_ptr(),
                                y2f16h.data_ptr(), -(B * T_out * cout), B, T, T_out, T_out, cin, cout, taps, stride, dil, pad, 1, 0, 0,
                                eng.stream()), "vn_conv1d_f32")
    torch.cuda.synchronize()
    err = (yh.cpu().double() - ref).abs()
    print(f"{note}: f16x2 raw max err {err.max().item():.3e}")
    assert bool((err <= tol).all())
    assert (y2h.cpu().double() - ref_s).abs().max().item() <= 2e-5
    for name, planes, dense in (("f16x2", y2h16, y2h), ("f32 -> fp16 planes", y2f16h, y2f)):
        back = planes[0].double() + planes[1].double() / 2048.0
        d = (back.reshape(B, T_out, cout) - dense.double()).abs()
        assert bool((d <= dense.double().abs() * 2.0 ** -22 + 2.0 ** -24).all()), name


@pytest.mark.parametrize("B,T,cin,cout,taps,dil", [(2, 301, 192, 192, 7, 3), (1, 1000, 384, 192, 2, 1), (3, 130, 96, 96, 7, 9), (2, 257, 256, 96, 7, 1)])
def test_conv_channels_on_rows_equals_the_column_tile_form(B, T, cin, cout, taps, dil):
    """Round 6: 96- / 192-channel convolutions run with the output channels on the tile's ROW axis (gemm_x3.hip CONVT: Y^T = W X^T, the tap
    gather on the W side, a transposed epilogue) — exactly one tile of rows instead of 128-wide column tiles that are three quarters full.
    Against the column-tile form of the same entry (a context created with VN_X3_CONVT=0): 192 channels BITWISE (the same products in the
    same order: one accumulator per output element, k-tiles in sequence), 96 channels to fp32 re-association noise (its 96-row tile splits
    the k-steps of a k-tile between two wave groups and adds the halves); raw, snake fp32 and snake planes, bias + residual."""
    import os
    from vampnet_amd.codec import DacCodec
    from vampnet_amd.engine import Engine
    g = torch.Generator().manual_seed(cout + taps)
    pad = dil * (taps - 1) // 2
    x = torch.randn(B, T, cin, generator=g)
    w = torch.randn(cout, taps, cin, generator=g) / math.sqrt(taps * cin)
    bias, alpha = torch.randn(cout, generator=g), torch.rand(cout, generator=g) + 0.5
    T_out = T + 2 * pad - dil * (taps - 1)
    resid = torch.randn(B, T_out, cout, generator=g)
    outs = []
    for flag in ("1", "0"):
        old = os.environ.get("VN_X3_CONVT")
        os.environ["VN_X3_CONVT"] = flag
        try:
            e = Engine("cuda:0")                       # the switch is read when the context is created
        finally:
            if old is None:
                os.environ.pop("VN_X3_CONVT", None)
            else:
                os.environ["VN_X3_CONVT"] = old
        xd, wd, bd, ad, rd = (t.cuda().contiguous() for t in (x, w, bias, alpha, resid))
        codec = DacCodec.__new__(DacCodec)
        codec.engine, codec.lib, codec.device, codec.precision = e, e.lib, e.device, "bf16x3"
        x16, w16 = e.split3(xd.reshape(-1, cin)), codec._tile_planes(wd.reshape(cout, -1))
        y, y2 = torch.full((B, T_out, cout), float("nan"), device="cuda"), torch.full((B, T_out, cout), float("nan"), device="cuda")
        y216 = torch.zeros(3, B * T_out, cout, device="cuda", dtype=torch.bfloat16)
        e.check(e.lib.vn_conv1d_bf16x3(e.handle, x16.data_ptr(), B * T * cin, w16.data_ptr(), bd.data_ptr(), rd.data_ptr(), ad.data_ptr(),
                                       y.data_ptr(), y2.data_ptr(), y216.data_ptr(), B * T_out * cout, B, T, T_out, T_out, cin, cout, taps,
                                       1, dil, pad, 1, 0, 0, e.stream()), "vn_conv1d_bf16x3")
        torch.cuda.synchronize()
        outs.append((y.cpu(), y2.cpu(), y216.cpu()))
    (ya, y2a, pa), (yb, y2b, pb) = outs
    assert bool(torch.isfinite(ya).all()) and bool(torch.isfinite(y2a).all())
    if cout == 192:
        assert torch.equal(ya, yb) and torch.equal(y2a, y2b) and torch.equal(pa, pb)
    else:
        scale = yb.abs().max().item()
        assert (ya - yb).abs().max().item() <= 2e-6 * scale and (y2a - y2b).abs().max().item() <= 4e-6 * max(1.0, y2b.abs().max().item())
        p = pa.float()
        assert torch.equal((p[0] + p[1] + p[2]).reshape(B, T_out, cout), y2a)


def test_conv_transpose_phases_on_bf16x3(eng):
    """WNConvTranspose1d(k = 2 s, stride s) as s two-tap phase convolutions (dil = -1, out_stride = s) through the bf16x3 entry,
    against torch's conv_transpose1d in float64."""
    from vampnet_amd.codec import DacCodec
    g = torch.Generator().manual_seed(3)
    B, T, cin, cout, st = 2, 40, 256, 128, 4
    x = torch.randn(B, T, cin, generator=g)
    w = torch.randn(cin, cout, 2 * st, generator=g) / math.sqrt(2 * cin)          # ConvTranspose1d weight layout
    bias = torch.randn(cout, generator=g)
    pad = math.ceil(st / 2)
    T_out = (T - 1) * st - 2 * pad + 2 * st
    ref = torch.nn.functional.conv_transpose1d(x.double().permute(0, 2, 1), w.double(), bias.double(), stride=st, padding=pad).permute(0, 2, 1)
    codec = DacCodec.__new__(DacCodec)
    codec.engine, codec.lib, codec.device, codec.precision = eng, eng.lib, eng.device, "bf16x3"
    x16 = eng.split3(x.cuda().reshape(-1, cin))
    y = torch.zeros(B, T_out, cout, device="cuda")
    bd = bias.cuda()
    for r in range(st):
        ph = torch.stack([w[:, :, r], w[:, :, r + st]], dim=0).permute(2, 0, 1).contiguous().cuda()      # [Cout][2][Cin]
        w16 = codec._tile_planes(ph.reshape(cout, -1))
        eng.check(eng.lib.vn_conv1d_bf16x3(eng.handle, x16.data_ptr(), B * T * cin, w16.data_ptr(), bd.data_ptr(), None, None,
                                           y.data_ptr(), None, None, 0, B, T, T + 1, T_out, cin, cout, 2, 1, -1, 0, st, r - pad, 0,
                                           eng.stream()), "vn_conv1d_bf16x3")
    err = (y.cpu().double() - ref).abs().max().item()
    print(f"transposed conv via {st} phases on the bf16x3 pipe: max err {err:.3e}")
    assert err <= 2e-5


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2", "f32"])
@pytest.mark.parametrize("cfg,B,frames", [(D.DAC_TINY_CFG, 3, 21), (D.DAC_DEFAULT_CFG, 2, 9)])
def test_one_call_per_direction_equals_the_layer_calls(eng, cfg, B, frames, precision):
    """vn_dac_encode / vn_dac_decode (include/vampnet_hip.h; ONE C call walking the recorded layer loop over one planned arena) give
    bitwise what the per-layer entry points give when the host issues them one by one (VN_CODEC_EAGER=1: the round-3 path): the same
    launches in the same order — also when the program is run twice (the arena's recycled bytes carry nothing over) and for a second
    shape next to the first (programs are cached per shape)."""
    from vampnet_amd.codec import DacCodec
    sd = D.synth_dac_state_dict(cfg, 0)
    prog = DacCodec(sd, cfg, engine=eng, precision=precision)
    eager = DacCodec(sd, cfg, engine=eng, precision=precision)
    eager.use_program = False
    assert prog.use_program
    hop = D.hop_length(cfg)
    for b, fr in ((B, frames), (1, frames + 2)):
        audio = _audio(b, hop * fr, seed=fr)
        a, e = prog.encode(audio, 44100), eager.encode(audio, 44100)
        assert torch.equal(a["codes"], e["codes"]) and torch.equal(a["z"], e["z"])
        again = prog.encode(audio, 44100)
        assert torch.equal(again["codes"], e["codes"])
        codes = e["codes"]
        wa, we = prog.decode_codes(codes), eager.decode_codes(codes)
        assert wa.shape == we.shape == (b, 1, hop * fr) and torch.equal(wa, we)
        assert torch.equal(prog.decode_codes(codes), we)
    assert len(prog._programs) == 4 and not eager._programs
    p = prog._programs[(1, B, frames, precision)]
    print(f"[{precision}] decode program B={B} T={frames}: {p['n_ops']} launches behind one call, arena {p['arena_bytes'] / 2**20:.1f} MiB")


@pytest.mark.parametrize("precision", ["bf16x3", "f16x2", "f32"])
@pytest.mark.parametrize("cfg,B,frames", [(D.DAC_TINY_CFG, 3, 21), (D.DAC_DEFAULT_CFG, 2, 9)])
def test_program_planned_in_c_equals_the_python_recorder(eng, cfg, B, frames, precision):
    """vn_codec_create_from_weights (csrc/codec_plan.hip: weights re-laid, layer loop recorded and arena planned IN C from a flat blob
    of state_dict tensors — what a host without Python calls) builds the program the Python recorder builds: vn_dac_encode /
    vn_dac_decode on it give bitwise the codes / the audio of DacCodec, in every precision, also when run twice; the tensor table
    covers exactly the tensors the codec reads."""
    import ctypes as C
    from vampnet_amd.codec import DacCodec, codec_cfg_struct, pack_codec_blob
    lib = eng.lib
    sd = D.synth_dac_state_dict(cfg, 0)
    ref = DacCodec(sd, cfg, engine=eng, precision=precision)
    cs = codec_cfg_struct(cfg)
    blob = pack_codec_blob(lib, cs, sd).cuda()
    n = C.c_int()
    assert lib.vn_codec_tensor_count(C.byref(cs), C.byref(n)) == 0 and n.value > 50
    off, cnt = C.c_int64(), C.c_int64()
    assert lib.vn_codec_tensor_offset(C.byref(cs), b"decoder.model.0.weight", C.byref(off), C.byref(cnt)) == 0
    assert cnt.value == sd["decoder.model.0.weight_v"].numel()
    assert lib.vn_codec_tensor_offset(C.byref(cs), b"no.such.tensor", C.byref(off), C.byref(cnt)) != 0
    hop = D.hop_length(cfg)
    pcode = {"f32": 0, "bf16x3": 2, "f16x2": 3}[precision]
    audio = _audio(B, hop * frames, seed=frames).cuda().reshape(B, hop * frames).contiguous()
    want = ref.encode(audio.reshape(B, 1, -1), 44100)["codes"]
    enc, dec = C.c_void_p(), C.c_void_p()
    eng.check(lib.vn_codec_create_from_weights(eng.handle, C.byref(cs), blob.data_ptr(), 0, B, hop * frames, pcode, C.byref(enc)), "create enc")
    eng.check(lib.vn_codec_create_from_weights(eng.handle, C.byref(cs), blob.data_ptr(), 1, B, frames, pcode, C.byref(dec)), "create dec")
    try:
        for _ in range(2):
            codes = torch.empty(B, cfg["n_codebooks"], frames, dtype=torch.int64, device="cuda")
            eng.check(lib.vn_dac_encode(enc, audio.data_ptr(), codes.data_ptr(), eng.stream()), "vn_dac_encode")
            assert torch.equal(codes, want)
            wave = torch.empty(B, hop * frames, device="cuda")
            eng.check(lib.vn_dac_decode(dec, want.data_ptr(), wave.data_ptr(), eng.stream()), "vn_dac_decode")
            assert torch.equal(wave.reshape(B, 1, -1), ref.decode_codes(want))
        # a program checks its direction; bad arguments are refused with a message
        assert lib.vn_dac_decode(enc, want.data_ptr(), wave.data_ptr(), eng.stream()) != 0
        bad = C.c_void_p()
        assert lib.vn_codec_create_from_weights(eng.handle, C.byref(cs), blob.data_ptr(), 0, B, hop * frames + 1, pcode, C.byref(bad)) != 0
        assert b"hop" in lib.vn_last_error(eng.handle)
        assert lib.vn_codec_create_from_weights(eng.handle, C.byref(cs), blob.data_ptr(), 0, B, hop, 1, C.byref(bad)) != 0
    finally:
        lib.vn_codec_destroy(enc)
        lib.vn_codec_destroy(dec)


def test_preprocess_on_the_device_equals_the_host_twin(eng):
    """Interface._preprocess (interface.py:206-217) on the device (csrc/preprocess.hip: BS.1770-4 loudness with the K-weighting IIR
    filters evaluated in parallel over 100 ms chunks through their state-space form, gain, peak limit, pad) against the all-host
    twin (scipy lfilter in float64): loudness to 1e-6 LU, samples to 1e-6 relative — a normal clip, a quiet one, one that the gain
    pushes over full scale (peak limit), silence (stays silent), a clip shorter than one 400 ms block, stereo and a length that is
    not a multiple of anything.  Both are restatements of audiotools' chain: parity with the reference is UNPINNED."""
    import ctypes as C
    from vampnet_amd.codec import AudioSignal, DacCodec, integrated_loudness
    cfg = D.DAC_TINY_CFG
    codec = DacCodec(D.synth_dac_state_dict(cfg, 0), cfg, engine=eng, precision="f32")
    sr = codec.sample_rate
    g = np.random.default_rng(3)
    T = 3 * sr + 1234
    t = np.arange(T) / sr
    clips = [0.2 * np.sin(2 * np.pi * 220 * t) + 0.05 * g.standard_normal(T),          # normal
             1e-3 * g.standard_normal(T),                                               # quiet: a large gain
             0.02 * np.sin(2 * np.pi * 90 * t) + 0.9 * (np.abs(t - 1.5) < 2e-3),        # a click: the gain drives it over full scale
             np.zeros(T),                                                               # silence
             0.3 * g.standard_normal(T) * (t > 2.0)]                                    # gated: two thirds of the blocks are silent
    x = torch.from_numpy(np.stack(clips).astype(np.float32))[:, None, :]
    x = torch.cat([x, 0.5 * x.flip(0)], dim=1)                                           # stereo: to_mono first
    for sig in (AudioSignal(x, sr), AudioSignal(x[:2, :, :sr // 5], sr), AudioSignal(x[:1, :1, :7], sr)):
        dev = codec.preprocess_signal(sig)
        host = codec.preprocess_signal_host(sig)
        assert dev.samples.is_cuda and dev.samples.shape == host.samples.shape
        assert dev.samples.shape[-1] % codec.hop_length == 0
        d, h = dev.samples.cpu(), host.samples
        scale = h.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-30)
        err = ((d - h).abs() / scale).max().item()
        print(f"T = {sig.samples.shape[-1]}: max relative sample error {err:.2e}; peaks {h.abs().amax(dim=(1, 2)).tolist()}")
        assert err <= 1e-6
        assert float(h.abs().max()) <= 1.0 + 1e-6
    # the loudness figure itself
    mono = x.mean(dim=1)
    B, Tn = mono.shape
    nbytes = C.c_int64()
    eng.check(eng.lib.vn_preprocess_workspace(B, Tn, sr, C.byref(nbytes)), "ws")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device="cuda")
    y = torch.empty(B, Tn, device="cuda")
    lufs = torch.empty(B, device="cuda")
    kw = codec._kw_cache
    mono_d = mono.cuda().contiguous()      # a temporary here is a use after free: its block is released before the call is even made
    eng.check(eng.lib.vn_preprocess_f32(eng.handle, mono_d.data_ptr(), y.data_ptr(), B, Tn, Tn, sr, -24.0, kw[0], kw[1],
                                        ws.data_ptr(), lufs.data_ptr(), eng.stream()), "vn_preprocess_f32")
    want = [integrated_loudness(mono[b:b + 1].numpy(), sr) for b in range(B)]
    print("LUFS device", lufs.cpu().tolist(), "host", want)
    assert max(abs(a - b) for a, b in zip(lufs.cpu().tolist(), want)) <= 2e-5          # fp32 output of a float64 figure
    assert eng.lib.vn_preprocess_f32(eng.handle, mono_d.data_ptr(), y.data_ptr(), B, Tn, Tn, 11025, -24.0, kw[0], kw[1], ws.data_ptr(),
                                     None, eng.stream()) != 0


def test_codec_program_abi_errors(eng):
    """the program entry points return a status + message on bad input (no crash, no exception across the ABI)"""
    import ctypes as C
    from vampnet_amd import _lib
    lib = eng.lib
    ops = (_lib.vn_codec_op * 1)()
    ops[0].kind = 99
    h = C.c_void_p()
    assert lib.vn_codec_create(eng.handle, ops, 1, 0, C.byref(h)) != 0 and b"unknown kind" in lib.vn_last_error(eng.handle)
    assert lib.vn_codec_create(eng.handle, ops, 0, 0, C.byref(h)) != 0
    assert lib.vn_codec_create(eng.handle, ops, 1, 7, C.byref(h)) != 0
    ops[0].kind = 6                                              # an rvq_decode with null pointers: the layer entry refuses it
    assert lib.vn_codec_create(eng.handle, ops, 1, 1, C.byref(h)) == 0
    x = torch.zeros(8, device="cuda")
    assert lib.vn_dac_encode(h, x.data_ptr(), x.data_ptr(), None) != 0 and b"decodes" in lib.vn_last_error(eng.handle)
    assert lib.vn_dac_decode(h, x.data_ptr(), x.data_ptr(), None) != 0 and b"codec program: op 0" in lib.vn_last_error(eng.handle)
    lib.vn_codec_destroy(h)
