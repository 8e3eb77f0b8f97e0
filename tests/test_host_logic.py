"""CPU tests of the product's host logic (no GPU): mask builders, RNG replay, schedule, weight packing,
C-ABI surface, Interface orchestration (with an oracle-backed stand-in for the device model)."""
import ctypes as C
import os
import re

import math
import numpy as np
import pytest
import torch

from oracle import vampnet_oracle as O, weights as W
from vampnet_amd import _lib, masks
from vampnet_amd.engine import draw_noise_host, pack_weights, VampNetModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    """The shared library loads on a GPU-less host and exports exactly what include/vampnet_hip.h (the boundary) and
    include/vampnet_hip_debug.h (tuning / test hooks, every vn_debug_* of them per OBJECT — a vn_ctx, a vn_model or a vn_train; the
    vn_guard_* allocator harness is per PROCESS, as an allocator is) declare."""
    hdr = open(os.path.join(ROOT, "include", "vampnet_hip.h")).read()
    dbg = open(os.path.join(ROOT, "include", "vampnet_hip_debug.h")).read()
    assert not re.search(r"\bvn_debug_\w+\s*\(", hdr), "debug hooks belong in vampnet_hip_debug.h"
    dbg_code = re.sub(r"/\*.*?\*/", "", dbg, flags=re.S)
    for name, args in re.findall(r"\bint\s+(vn_debug_\w+)\s*\(([^;{]*?)\)\s*;", dbg_code, flags=re.S):
        first = args.split(",")[0]
        assert any(k in first.replace(" *", "*") for k in ("vn_ctx*", "vn_model*", "vn_train*")), f"{name} must act on an object, not the process"
    hdr = hdr + dbg
    declared = set(re.findall(r"\b(vn_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name)
    assert b"gfx950" in lib.vn_version()
    # the ctypes prototypes carry as many arguments as the C declarations (a short list would silently pass garbage)
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decls = re.findall(r"\b(?:int|void|const char\*)\s+\*?\s*(vn_\w+)\s*\(([^;{]*?)\)\s*;", code, flags=re.S)
    assert {n for n, _ in decls} == declared
    for name, args in decls:
        n_args = 0 if args.strip() in ("", "void") else len(args.split(","))
        assert n_args == len(_lib.SYMBOLS[name][1]), (name, n_args, len(_lib.SYMBOLS[name][1]))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.VnError):
        _lib.load()


def test_engine_refuses_cpu():
    from vampnet_amd.engine import Engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.VnError):
        Engine("cuda:0")


def test_weight_layout_and_packing():
    lib = _lib.load()
    d = W.TINY_COARSE_DIMS
    dims = _lib.vn_dims(d["n_layers"], d["n_heads"], d["d_model"], d["n_codebooks"], d["n_cond"], d["vocab"],
                        d["latent_dim"], 32, 128, 1e-6, 2, 64)
    n = C.c_int64()
    assert lib.vn_weights_size(C.byref(dims), C.byref(n)) == 0
    spans = []
    for tid in range(13):
        for layer in (range(d["n_layers"]) if tid >= _lib.W_NORM1 else [0]):
            off, cnt = C.c_int64(), C.c_int64()
            assert lib.vn_weights_offset(C.byref(dims), tid, layer, C.byref(off), C.byref(cnt)) == 0
            assert off.value % 64 == 0
            spans.append((off.value, off.value + cnt.value))
    spans.sort()
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= n.value
    sd, cb = W.synth_state_dict(d, 0), W.synth_codebooks()
    blob = pack_weights(lib, dims, sd, cb)
    D, V, Cp = d["d_model"], d["vocab"], d["n_codebooks"] - d["n_cond"]
    off, cnt = C.c_int64(), C.c_int64()
    # classifier rows re-ordered (p c) -> (c p), weight-norm folded
    lib.vn_weights_offset(C.byref(dims), _lib.W_CLS_W, 0, C.byref(off), C.byref(cnt))
    wc = blob[off.value:off.value + cnt.value].view(Cp, V, D)
    ref = O.classifier_weight(sd).squeeze(-1)
    assert torch.equal(wc[2, 17], ref[17 * Cp + 2])
    # W1 interleave: packed row 64g+i = value row 32g+i ; 64g+32+i = gate row 2D+32g+i
    lib.vn_weights_offset(C.byref(dims), _lib.W_W1, 1, C.byref(off), C.byref(cnt))
    w1p = blob[off.value:off.value + cnt.value].view(4 * D, D)
    w1 = sd["transformer.layers.1.feed_forward.w_1.weight"]
    assert torch.equal(w1p[64 * 3 + 5], w1[32 * 3 + 5]) and torch.equal(w1p[64 * 3 + 32 + 5], w1[2 * D + 32 * 3 + 5])
    # LoRA merge: W + (B A) / 8
    sd2 = dict(sd)
    A, Bm = torch.randn(8, D), torch.randn(D, 8)
    sd2["transformer.layers.0.self_attn.fc.lora_A"], sd2["transformer.layers.0.self_attn.fc.lora_B"] = A, Bm
    blob2 = pack_weights(lib, dims, sd2, cb)
    lib.vn_weights_offset(C.byref(dims), _lib.W_WO, 0, C.byref(off), C.byref(cnt))
    got = blob2[off.value:off.value + cnt.value].view(D, D)
    assert torch.allclose(got, sd["transformer.layers.0.self_attn.fc.weight"] + (Bm @ A) / 8.0)
    bad = _lib.vn_dims(2, 4, 128, 4, 0, 1024, 8, 32, 128, 1e-6, 1, 8)       # d_head 32
    assert lib.vn_weights_size(C.byref(bad), C.byref(n)) != 0


# ------------------------------------------------------------------------------------------ masks
@pytest.mark.parametrize("kw", [dict(), dict(periodic_prompt=5, upper_codebook_mask=2, _dropout=0.1),
                                dict(rand_mask_intensity=0.8, prefix_s=0.2, suffix_s=0.1, periodic_prompt=0),
                                dict(periodic_prompt=13, periodic_prompt_width=3, ncc=1),
                                dict(periodic_prompt=4, periodic_prompt_width=5, upper_codebook_mask=14)])
@pytest.mark.parametrize("B,T", [(1, 575), (2, 120), (3, 7)])
def test_build_mask_matches_oracle_and_rng_stream(kw, B, T):
    z = W.synth_codes(B, 14, T, seed=4)
    for seed in (0, 1, 2):
        torch.manual_seed(seed)
        ref = O.build_mask(z, **kw)
        tail_ref = torch.rand(3)
        torch.manual_seed(seed)
        got = masks.build_mask(z, rand_mask_intensity=kw.get("rand_mask_intensity", 1.0),
                               n_prefix=O.s2t(kw.get("prefix_s", 0.0)), n_suffix=O.s2t(kw.get("suffix_s", 0.0)),
                               periodic_prompt=kw.get("periodic_prompt", 7),
                               periodic_prompt_width=kw.get("periodic_prompt_width", 1),
                               dropout=kw.get("_dropout", 0.0), upper_codebook_mask=kw.get("upper_codebook_mask", 3),
                               ncc=kw.get("ncc", 0))
        assert torch.equal(ref, got)
        assert torch.equal(tail_ref, torch.rand(3))          # generator left in the same state


def test_apply_mask_and_asserts():
    z = W.synth_codes(2, 4, 9)
    m = (torch.arange(9) % 2).expand(2, 4, 9).long()
    out, _ = masks.apply_mask(z, m, 1024)
    ref, _ = O.apply_mask(z, m, 1024)
    assert torch.equal(out, ref)
    with pytest.raises(AssertionError):
        masks.apply_mask(z, m.int(), 1024)
    with pytest.raises(AssertionError):
        masks.apply_mask(z, m * 2, 1024)
    with pytest.raises(AssertionError):
        masks.apply_mask(z, m[:, :2], 1024)


# ------------------------------------------------------------------------------------------ RNG replay + schedule
@pytest.mark.parametrize("cutoff", [1.0, 0.5, -1.0])
def test_noise_ledger_matches_reference_stream(cutoff):
    dims, B, T, steps = W.TINY_COARSE_DIMS, 3, 21, 4
    sd, cb = W.synth_state_dict(dims, 0), W.synth_codebooks()
    z = W.synth_codes(B, 4, T, seed=1)
    mask = torch.ones_like(z)
    trace = []
    torch.manual_seed(5)
    O.generate(sd, dims, cb, z, mask, sampling_steps=steps, sample_cutoff=cutoff, trace=trace)
    tail = torch.rand(2)
    N, V = T * 4, 1024
    torch.manual_seed(5)
    exp, unif = draw_noise_host(B, N, V, steps, cutoff)
    assert torch.equal(tail, torch.rand(2))
    for i, t in enumerate(trace):
        if t["exp"] is not None:
            assert torch.equal(exp[i], t["exp"])
        else:
            assert not exp[i].any()
        assert torch.equal(unif[i], t["unif"])
    # shard view: rows of items [1, 3) of the same global stream
    torch.manual_seed(5)
    exp_s, unif_s = draw_noise_host(B, N, V, steps, cutoff, b0=1, nb=2)
    assert torch.equal(exp_s, exp[:, N:3 * N]) and torch.equal(unif_s, unif[:, 1:3])


def test_mask_schedule_matches_reference_arithmetic():
    for steps in (1, 2, 6, 12, 36):
        for n0 in (1, 43, 2049, 2300, 16392, 64 * 2049):
            got = VampNetModel.mask_schedule(steps, n0)
            for i in range(steps):
                r = torch.tensor((i + 1) / steps).repeat(2)
                want = torch.floor(O.gamma(r) * torch.tensor(n0)).long()[0].item()
                assert got[i] == want
            assert got[-1] == 0


def test_bs1770_loudness_reference_tone():
    """ITU-R BS.1770: a 0 dBFS 997 Hz sine reads -3.01 LKFS (per channel)."""
    from vampnet_amd.codec import integrated_loudness
    sr = 44100
    t = np.arange(sr * 5) / sr
    x = np.sin(2 * np.pi * 997 * t)[None]
    assert abs(integrated_loudness(x, sr) - (-3.01)) < 0.1
    assert abs(integrated_loudness(0.1 * x, sr) - (-23.01)) < 0.1
    assert integrated_loudness(np.zeros((1, sr)), sr) == -70.0


# ----------------------------------------------------------------------------- torch CPU RNG stream (rng="torch_device")
def _mt_raw(seed, n):
    """at::mt19937(seed) raw outputs: numpy's MT19937 with the legacy init_genrand seeding is the same engine."""
    import numpy as np
    bg = np.random.MT19937()
    bg._legacy_seeding(seed)
    return bg.random_raw(n).astype(np.uint64)


def test_torch_rng_stream_formulas():
    """The two facts csrc/torch_rng.hip relies on, pinned against THIS torch build: exponential_ on a float32 CPU tensor is
    float(-log1p(-u53)) of consecutive (hi, lo) mt19937 word pairs, uniform_(a, b) is the 24-bit mantissa formula, both
    strictly sequential in the generator's output — for a tensor as large as one sampling step (2300 x 1024)."""
    import numpy as np
    n = 2300 * 1024
    raw = _mt_raw(123, 2 * n + 2300)
    torch.manual_seed(123)
    x = torch.empty(2300, 1024).exponential_(1).reshape(-1).numpy()
    r64 = (raw[0:2 * n:2] << np.uint64(32)) | raw[1:2 * n:2]
    u = (r64 & np.uint64((1 << 53) - 1)).astype(np.float64) * 2.0 ** -53
    assert np.array_equal(x, (-np.log1p(-u)).astype(np.float32))
    y = torch.zeros(2300).uniform_(1e-20, 1).numpy()
    r = raw[2 * n:2 * n + 2300]
    emu = ((r & np.uint64((1 << 24) - 1)).astype(np.float32) * np.float32(2.0 ** -24)) * np.float32(1 - 1e-20) + np.float32(1e-20)
    assert np.array_equal(y, emu)


def test_torch_rng_state_parse_and_patch():
    """torch.get_rng_state() layout used to hand the generator over to the device and back."""
    import numpy as np
    from vampnet_amd.torch_rng import parse_torch_rng_state, patch_torch_rng_state

    def temper(y):
        y = np.uint32(y)
        y ^= y >> np.uint32(11)
        y ^= (y << np.uint32(7)) & np.uint32(0x9d2c5680)
        y ^= (y << np.uint32(15)) & np.uint32(0xefc60000)
        y ^= y >> np.uint32(18)
        return int(y)

    torch.manual_seed(77)
    state, pos = parse_torch_rng_state(torch.get_rng_state())
    assert pos == 624 and state[0] == 77                              # freshly seeded: block exhausted, init_genrand(77)
    _ = torch.empty(1000).exponential_()                              # 2000 words = 3 regenerations + 128
    blob = torch.get_rng_state()
    state, pos = parse_torch_rng_state(blob)
    assert pos == 128
    raw = _mt_raw(77, 2010)
    assert [temper(state[pos + i]) for i in range(5)] == [int(v) for v in raw[2000:2005]]
    # patch: move the position forward by 3 words == what drawing 3 more 32-bit randoms does
    torch.set_rng_state(patch_torch_rng_state(blob, state, pos + 3))
    nxt = torch.empty(2).uniform_(0, 1).numpy()                       # one word per element: (w & (2^24 - 1)) * 2^-24
    want = [np.float32(int(raw[2003 + i]) & ((1 << 24) - 1)) * np.float32(2.0 ** -24) for i in range(2)]
    assert nxt.tolist() == [float(w) for w in want]
    # and a patched "block exhausted" position regenerates on the next draw
    torch.set_rng_state(patch_torch_rng_state(blob, state, 624))
    st2, pos2 = parse_torch_rng_state(torch.get_rng_state())
    assert pos2 == 624 and np.array_equal(st2, state)


# ----------------------------------------------------------------------------- mt19937 jump-ahead (parallel torch_device noise)
def _mt_blocks(state, n_blocks):
    """numpy restatement of at::mt19937::next_state(): successive 624-word blocks of UNTEMPERED words."""
    import numpy as np

    def tw(u, v):
        y = (u & np.uint32(0x80000000)) | (v & np.uint32(0x7fffffff))
        return (y >> np.uint32(1)) ^ np.where(v & np.uint32(1), np.uint32(0x9908b0df), np.uint32(0))

    out, o = [], state
    for _ in range(n_blocks):
        w = np.empty_like(o)
        w[0:227] = o[397:624] ^ tw(o[0:227], o[1:228])
        w[227:454] = w[0:227] ^ tw(o[227:454], o[228:455])
        w[454:623] = w[227:396] ^ tw(o[454:623], o[455:624])
        w[623] = w[396] ^ tw(o[623:624], w[0:1])[0]
        out.append(w)
        o = w
    return out


def test_mt19937_characteristic_polynomial():
    """PHI_EXPONENTS (vampnet_amd/mt_jump.py) is the minimal polynomial of the generator: Berlekamp-Massey on 2 x 19937 + 200
    output bits of at::mt19937 finds an LFSR of length 19937 whose reversed connection polynomial is exactly it."""
    import numpy as np
    from vampnet_amd import mt_jump as J
    bg = np.random.MT19937()
    bg._legacy_seeding(5489)
    st = bg.state["state"]["key"].astype(np.uint32)
    words = np.concatenate(_mt_blocks(st, (2 * 19937 + 200) // 624 + 1))
    c, length = J.berlekamp_massey_gf2((words[:2 * 19937 + 200] & 1).tolist())
    assert length == J.MT_DEGREE
    phi = int(bin(c)[2:].zfill(J.MT_DEGREE + 1)[::-1], 2)
    assert phi == J.PHI and len(J.PHI_EXPONENTS) == 135


@pytest.mark.parametrize("steps", [1, 623, 624, 100_003, 2 * 2300 * 1024])
def test_mt19937_jump_polynomial(steps):
    """x_{k+J} = XOR over the set bits i of x^J mod phi of x_{k+i}: 624 sliding XORs over the next 20 560 words give the state
    J steps ahead, and the generator continues from it."""
    import numpy as np
    from vampnet_amd import mt_jump as J
    bg = np.random.MT19937()
    bg._legacy_seeding(321)
    st = bg.state["state"]["key"].astype(np.uint32)
    need = steps + 3 * 624
    stream = np.concatenate(_mt_blocks(st, need // 624 + 35))
    g = J.jump_poly(steps)
    assert J.poly_mul(g, J.jump_poly(5)) == J.jump_poly(steps + 5)
    words = J.jump_poly_words(steps)
    idx = np.nonzero(np.unpackbits(words.view(np.uint8), bitorder="little"))[0]
    assert idx.max() < J.MT_DEGREE
    jumped = np.array([np.bitwise_xor.reduce(stream[k + idx]) for k in range(624)], dtype=np.uint32)
    assert np.array_equal(jumped, stream[steps:steps + 624])
    assert np.array_equal(_mt_blocks(jumped, 1)[0], stream[steps + 624:steps + 1248])


def test_bf16x3_split_numerics_on_the_host():
    """The numerics claim behind precision="bf16x3" (DESIGN.md §3/§4), checked with torch on the CPU: (1) the three-way
    bf16 split is EXACT (two exact fp32 remainders); (2) the six kept plane products, each exact in fp32 and accumulated in
    fp32, are as close to the float64 product as an fp32-accumulating fp32 GEMM; (3) keeping fewer terms is not."""
    g = torch.Generator().manual_seed(0)

    def bf(x):
        return x.to(torch.bfloat16).to(torch.float32)

    def split3(x):
        p0 = bf(x)
        r1 = x - p0
        p1 = bf(r1)
        return p0, p1, bf(r1 - p1)

    x = torch.cat([torch.randn(4096, generator=g), torch.randn(4096, generator=g) * 1e-20, torch.randn(4096, generator=g) * 1e20])
    p = split3(x)
    assert torch.equal(p[0] + p[1] + p[2], x)                       # exact, incl. tiny / huge magnitudes
    assert torch.equal(p[2], bf(p[2])) and torch.equal((x - p[0]) - p[1], p[2])
    M, N, K = 257, 384, 1280
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    ref = A.double() @ W.double().t()
    e_f32 = ((A @ W.t()).double() - ref).abs().max().item()
    a, w = split3(A), split3(W)

    def terms(pairs):
        acc = torch.zeros(M, N)
        for i, j in pairs:
            acc = acc + a[i] @ w[j].t()
        return (acc.double() - ref).abs().max().item()

    six = terms([(0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)])
    three = terms([(0, 1), (1, 0), (0, 0)])
    assert six <= 1.5 * e_f32 + 1e-7, (six, e_f32)                  # fp32 class (measured: 1.4e-6 vs 2.9e-6 at K = 1280)
    assert three > 4 * six                                          # the 2^-16 terms matter: ~2e-5


def test_f16x2_split_numerics_on_the_host():
    """The numerics claim behind precision="f16x2" (DESIGN.md §3/§4), checked with torch on the CPU: (1) h0 = fp16(x),
    h1 = fp16((x - h0) 2^11) reproduce x to 2^-22 |x| over fp16's normal range (round-to-nearest at both levels) and to 2^-25 in
    absolute terms below it; (2) the three kept products — a0 b0 into one fp32 accumulator, a0 b1 + a1 b0 into a second one that
    joins times 2^-11 — are as close to the float64 product as an fp32-accumulating fp32 GEMM of the unsplit operands, at every
    operand magnitude fp16 can hold (no per-tensor scaling); (3) WITHOUT the 2^11 on the second plane small operands lose it to
    fp16's subnormal range; (4) two products are not enough."""
    g = torch.Generator().manual_seed(0)

    def split2h(x, scale=2048.0):
        x = x.clamp(-65504.0, 65504.0)
        h0 = x.to(torch.float16).to(torch.float32)
        h1 = ((x - h0) * scale).to(torch.float16).to(torch.float32)
        return h0, h1

    mag = torch.exp(torch.empty(1 << 14).uniform_(math.log(1e-9), math.log(6e4), generator=g))
    x = mag * torch.where(torch.rand(1 << 14, generator=g) < 0.5, -1.0, 1.0)
    h0, h1 = split2h(x)
    err = (h0.double() + h1.double() / 2048.0 - x.double()).abs()
    assert torch.all(err <= torch.maximum(x.double().abs() * 2.0 ** -22, torch.tensor(2.0 ** -24, dtype=torch.float64)))
    assert split2h(torch.tensor([1e6, -3e38]))[0].tolist() == [65504.0, -65504.0]            # saturation, not inf
    M, N, K = 257, 384, 1280
    W = torch.randn(N, K, generator=g) / K ** 0.5

    def rms_rel(c, ref):
        return ((c.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()

    for a_scale, w_scale in [(1.0, 1.0), (0.02, 1.0), (300.0, 1.0), (1.0, 0.05)]:
        A = torch.randn(M, K, generator=g) * a_scale
        Wm = W * w_scale
        ref = A.double() @ Wm.double().t()
        e_f32 = rms_rel(A @ Wm.t(), ref)
        (a0, a1), (w0, w1) = split2h(A), split2h(Wm)
        three = rms_rel(a0 @ w0.t() + (a0 @ w1.t() + a1 @ w0.t()) * (1.0 / 2048.0), ref)
        two = rms_rel(a0 @ w0.t() + (a0 @ w1.t()) * (1.0 / 2048.0), ref)
        assert three <= 1.25 * e_f32 + 2e-8, (a_scale, w_scale, three, e_f32)       # measured: 2.4e-7 vs 2.5e-7 (the fp32 accumulation itself)
        assert two > 100 * three
    (a0, a1), (w0, w1) = split2h(torch.randn(M, K, generator=g), 1.0), split2h(W * 0.05, 1.0)       # second plane NOT scaled
    A = a0 + 0.0
    plain = rms_rel(a0 @ w0.t() + a0 @ w1.t() + a1 @ w0.t(), (a0 + a1).double() @ (W * 0.05).double().t())
    assert plain > 2e-6                                             # weights ~ 1e-3: their remainders fall into fp16 subnormals


def test_f16x2_attention_numerics_on_the_host():
    """The numerics claim behind the attention of precision="f16x2" (attention_x3.hip, NP = 2), emulated with torch on the CPU: q / 8, k
    and 16 v as fp16 two-plane splits WITHOUT the 2^11 on the second plane (one accumulator), softmax weights formed as
    2^(s log2 e + c) with c = 4 - m log2 e (their factor 16 in the exponent), split the same way; three products per matrix product.
    Against float64 the output error stays in the class of plain fp32 arithmetic — for random scores, for near-uniform attention
    (small outputs), for peaky attention and with the reference max deferred by up to 6."""
    g = torch.Generator().manual_seed(0)
    LOG2E = 1.44269502162933349609375

    def split2u(x):
        x = x.clamp(-65504.0, 65504.0)
        h0 = x.to(torch.float16).to(torch.float32)
        return h0.double(), (x - h0).to(torch.float16).to(torch.float64)

    T = 575
    bias = 0.5 * torch.randn(T, T, generator=g)
    cases = {"random": (1.0, 1.0, bias), "near-uniform": (0.05, 0.05, bias * 0), "peaky": (3.0, 3.0, bias)}
    for name, (sq, sk, b) in cases.items():
        q, k, v = sq * torch.randn(T, 64, generator=g), sk * torch.randn(T, 64, generator=g), torch.randn(T, 64, generator=g)
        ref = torch.softmax(q.double() @ k.double().t() / 8 + b.double(), -1) @ v.double()
        f32 = (torch.softmax(q @ k.t() / 8 + b, -1) @ v).double()
        (q0, q1), (k0, k1), (v0, v1) = split2u(q * 0.125), split2u(k), split2u(v * 16.0)
        S = (q0 @ k0.t() + q0 @ k1.t() + q1 @ k0.t() + b.double()).float()
        for defer in (0.0, 6.0):                                   # reference max up to 6 below the row max: weights up to e^6
            m = S.max(-1, keepdim=True).values - defer
            c = torch.addcmul(torch.tensor(4.0), m, torch.tensor(-LOG2E))          # one fp32 fma per row (emulated: mul + add in fp32)
            P = torch.exp2(torch.addcmul(c, S, torch.tensor(LOG2E)))               # 16 exp(s - m), fp32
            assert float(P.max()) < 6.5e3
            p0, p1 = split2u(P)
            O = p0 @ v0 + p0 @ v1 + p1 @ v0
            out = O / (P.double().sum(-1, keepdim=True) * 16.0)
            err, err32 = (out - ref).abs().max().item(), (f32 - ref).abs().max().item()
            assert err <= 2.0 * err32 + 3e-7 * ref.abs().max().item(), (name, defer, err, err32)


def _click_track(sr, seconds, times, seed=0):
    rng = np.random.default_rng(seed)
    y = 1e-4 * rng.standard_normal(int(sr * seconds)).astype(np.float32)
    for t in times:
        i, n = int(t * sr), 4000
        y[i:i + n] += (0.5 * rng.standard_normal(n) * np.exp(-np.arange(n) / 800.0)).astype(np.float32)
    return y


def test_onset_detector_building_blocks():
    """vampnet_amd/onsets.py restates librosa.onset.onset_detect [UNVERIFIED-DEP, librosa absent]: pin the pieces that have
    closed forms — slaney mel scale anchors, filterbank shape / normalisation, peak_pick and backtrack semantics."""
    from vampnet_amd import onsets as ON
    assert abs(float(ON._hz_to_mel(1000.0)) - 15.0) < 1e-12 and abs(float(ON._mel_to_hz(15.0)) - 1000.0) < 1e-9
    assert abs(float(ON._mel_to_hz(ON._hz_to_mel(6400.0))) - 6400.0) < 1e-6
    assert abs(float(ON._hz_to_mel(6400.0)) - 42.0) < 1e-9              # 15 + 27 log-steps of ln(6.4)/27
    fb = ON.mel_filterbank(44100)
    assert fb.shape == (128, 1025) and (fb >= 0).all() and (fb.sum(1) > 0).all()
    peaks = fb.argmax(1)
    assert (np.diff(peaks) >= 0).all() and peaks[0] > 0                 # triangles march up the spectrum
    # peak_pick: local max over [n - pre_max, n + post_max), delta above the local mean, greedy wait
    x = np.array([0, 0, 1.0, 0, 0, 0.5, 0.6, 0, 0, 0, 0.9, 0.9, 0, 0], dtype=np.float32)
    got = ON.peak_pick(x, pre_max=1, post_max=1, pre_avg=2, post_avg=3, delta=0.07, wait=1)
    assert got.tolist() == [2, 5, 10]        # post_max = 1: the window only looks BACK, so a rise fires on its first frame that
                                             # clears the mean (5), 6 and 11 fall inside `wait`
    # backtrack: nearest local minimum at or before the event; frame 0 always counts as one
    env = np.array([3, 2, 1, 2, 5, 4, 3, 3.5, 6, 1, 7], dtype=np.float32)
    assert ON.onset_backtrack(np.array([1, 4, 8, 10]), env).tolist() == [0, 2, 6, 9]
    assert ON.onset_detect(np.zeros(44100, dtype=np.float32), 44100, 768).size == 0


def test_onset_detect_finds_clicks_and_builds_the_mask():
    from vampnet_amd import masks as M, onsets as ON
    sr, hop = 44100, 768
    times = [0.5, 1.3, 2.0, 3.7, 5.05, 7.2, 9.0]
    y = _click_track(sr, 10.0, times)
    raw = ON.onset_detect(y, sr, hop, backtrack=False)
    want = np.array([t * sr / hop for t in times])
    assert len(raw) == len(times) and np.abs(raw - want).max() <= 1.0   # peak within a frame of the true attack
    bt = ON.onset_detect(y, sr, hop, backtrack=True)
    assert len(bt) == len(times) and ((raw - bt) >= 0).all() and ((raw - bt) <= 4).all()   # rolled back to the preceding dip
    z = torch.zeros(2, 14, 575, dtype=torch.long)
    m = M.onset_mask_from_samples(torch.from_numpy(y)[None, None], sr, z, hop, width=3)
    assert m.shape == z.shape and m.dtype == torch.long
    cols = torch.nonzero(m[0, 0] == 0).flatten().tolist()
    exp = sorted({c for i in bt.tolist() for c in range(i - 3, i + 3)})
    assert cols == exp and torch.equal(m[0, 0], m[1, 13])               # [idx - w, idx + w) on every codebook / item
    # reference quirk kept: an onset closer than `width` to the start gives a negative slice start -> nothing un-masked
    y2 = _click_track(sr, 2.0, [0.02])
    first = int(ON.onset_detect(y2, sr, hop)[0])
    assert first < 5
    m2 = M.onset_mask_from_samples(torch.from_numpy(y2)[None, None], sr, torch.zeros(1, 4, 115, dtype=torch.long), hop, width=5)
    assert int((m2 == 0).sum()) == 0
    # composition inside build_mask: AND with the other masks, upper codebooks re-masked afterwards
    torch.manual_seed(0)
    full = M.build_mask(z, periodic_prompt=0, onset_mask=m, upper_codebook_mask=3)
    assert torch.equal(full[:, :3], m[:, :3]) and bool((full[:, 3:] == 1).all())


def test_jump_polynomial_disk_cache(tmp_path, monkeypatch):
    """jump_poly_words keeps its 2 496-byte results on disk (VN_CACHE_DIR): a second process does not recompute, a damaged or
    unwritable cache only costs the recomputation."""
    from vampnet_amd import mt_jump as J
    monkeypatch.setenv("VN_CACHE_DIR", str(tmp_path))
    steps = 777_777
    want = np.frombuffer(J.jump_poly(steps).to_bytes(624 * 4, "little"), dtype=np.uint32)
    a = J.jump_poly_words(steps)
    f = tmp_path / f"mtjump_{steps}.bin"
    assert np.array_equal(a, want) and f.exists() and f.stat().st_size == 2496
    monkeypatch.setattr(J, "jump_poly", lambda s: (_ for _ in ()).throw(AssertionError("recomputed")))
    assert np.array_equal(J.jump_poly_words(steps), want)          # served from the file
    monkeypatch.undo()
    monkeypatch.setenv("VN_CACHE_DIR", str(tmp_path))
    f.write_bytes(b"short")                                         # damaged entry: recomputed and repaired
    assert np.array_equal(J.jump_poly_words(steps), want) and f.stat().st_size == 2496
    monkeypatch.setenv("VN_CACHE_DIR", "/proc/definitely/not/writable")
    assert np.array_equal(J.jump_poly_words(steps), want)


def test_python_sources_have_no_unbound_names():
    """A forgotten import in a `-m gpu` test file only shows up as a collection error on the GPU box: scan every Python source
    for names that are read but bound nowhere in their module (imports, defs, assignments, arguments, builtins)."""
    import ast
    import builtins
    import glob

    def unbound(path):
        tree = ast.parse(open(path).read(), path)
        bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
        for n in ast.walk(tree):
            if isinstance(n, (ast.Import, ast.ImportFrom)):
                bound.update((a.asname or a.name).split(".")[0] for a in n.names)
            elif isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                bound.add(n.name)
            elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
                bound.add(n.id)
            elif isinstance(n, ast.ExceptHandler) and n.name:
                bound.add(n.name)
            elif isinstance(n, ast.arg):
                bound.add(n.arg)
        return [(n.lineno, n.id) for n in ast.walk(tree)
                if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound]

    files = [f for pat in ("*.py", "vampnet_amd/*.py", "tests/*.py", "oracle/*.py", "scripts/*.py", "examples/*.py")
             for f in glob.glob(os.path.join(ROOT, pat))]
    assert len(files) > 30
    bad = {os.path.relpath(f, ROOT): u for f in files if (u := unbound(f))}
    assert not bad, bad


def test_split_plane_gemm_lds_addressing():
    """Model of the LDS layout of csrc/gemm_x3.hip (3 bf16 planes, verified on the GPU): replay the per-lane LDS-DMA destinations and source swizzle, then check that every
    ds_read_b128 fragment read fetches the (operand, plane, row, k) it feeds to the MFMA and that each of its four 16-lane
    groups touches 16 distinct 16-byte slots of the 256-byte bank row (conflict-free)."""
    def check(cfg, planes=3):
        RI, CJ, WR, WC = {1: (1, 2, 4, 2), 2: (2, 2, 4, 2), 3: (3, 1, 2, 4)}[cfg]       # x3_geo<CFG>
        BM = 32 * RI * WR
        APLANE = BM * 16            # floats
        BPLANE = 128 * 16
        STAGE = planes * APLANE + planes * BPLANE
        NQ = (planes * BM + planes * 128) // 16
        NPW = (NQ + 7) // 8
        lds = {}                    # half index -> (op, plane, row, k)
        issued = 0
        for wave in range(8):
            for j in range(NPW):
                q = 8 * j + wave if cfg == 3 else wave * NPW + j     # CFG 3: 60 instructions = 8 for waves 0-3, 7 for waves 4-7
                if cfg == 3 and j == NPW - 1 and wave >= 4:
                    continue
                assert q < NQ
                issued += 1
                for lane in range(64):
                    drow, dslot = lane >> 2, (lane & 3) ^ ((lane >> 4) & 3)
                    if q < planes * (BM // 16):
                        plane, row = q // (BM // 16), (q % (BM // 16)) * 16 + drow
                        op = "A"
                    else:
                        qb = q - planes * (BM // 16)
                        plane, row = qb >> 3, (qb & 7) * 16 + drow
                        op = "W"
                    base = (q * 256 + lane * 4) * 2
                    for e in range(8):
                        assert base + e not in lds
                        lds[base + e] = (op, plane, row, dslot * 8 + e)
        assert issued == NQ and len(lds) == STAGE * 2, (issued, len(lds), STAGE * 2)
        # bank conflicts of the ds_read_b128 groups
        groups = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
        groups += [[l + 32 for l in g] for g in groups]
        covered_a, covered_w = set(), set()
        for wave in range(8):
            wm, wn = wave // WC, wave % WC
            for s in range(2):
                for q in range(planes):
                    for i in range(RI):
                        addrs = {}
                        for lane in range(64):
                            l31, h, sw = lane & 31, lane >> 5, (lane >> 2) & 3
                            aRow = (wm * 32 * RI + l31) * 16
                            off = ((2 * s + h) ^ sw) * 4
                            fo = q * APLANE + aRow + i * 32 * 16 + off
                            addrs[lane] = fo * 4
                            covered_a.add(wm * 32 * RI + i * 32 + l31)
                            for e in range(8):
                                assert lds[fo * 2 + e] == ("A", q, wm * 32 * RI + i * 32 + l31, (2 * s + h) * 8 + e), (cfg, wave, lane)
                        for g in groups:
                            slots = {(addrs[l] // 16) % 16 for l in g}
                            assert len(slots) == 16, ("bank conflict A", cfg, wave, s, q, i)
                    for j in range(CJ):
                        addrs = {}
                        for lane in range(64):
                            l31, h, sw = lane & 31, lane >> 5, (lane >> 2) & 3
                            bRow = (wn * 32 * CJ + l31) * 16
                            off = ((2 * s + h) ^ sw) * 4
                            fo = planes * APLANE + q * BPLANE + bRow + j * 32 * 16 + off
                            addrs[lane] = fo * 4
                            covered_w.add(wn * 32 * CJ + j * 32 + l31)
                            for e in range(8):
                                assert lds[fo * 2 + e] == ("W", q, wn * 32 * CJ + j * 32 + l31, (2 * s + h) * 8 + e), (cfg, wave, lane)
                        for g in groups:
                            assert len({(addrs[l] // 16) % 16 for l in g}) == 16, ("bank conflict W",)
        assert covered_a == set(range(BM)) and covered_w == set(range(128))      # the eight wave tiles tile BM x 128
        return STAGE * 4, NPW


    assert check(1) == (48 * 1024, 6)          # gemm_x3.hip, 128 x 128 (waves 4 x 2)
    assert check(2) == (72 * 1024, 9)          # gemm_x3.hip, 256 x 128 (waves 4 x 2)
    assert check(3) == (60 * 1024, 8)          # gemm_x3.hip, 192 x 128 (waves 2 x 4; 60 DMA instructions per stage)


def test_attention_x3_lds_addressing():
    """Model of attention_x3.hip's LDS images (K plane tile: 32 rows x 128 B, V^T plane tile: 64 rows x 64 B): replay the per-lane
    LDS-DMA destinations and source-side swizzles, then check that every ds_read_b128 of the two MFMA A operands fetches the
    (row, 8-element slot) it feeds to the matrix core — K rows through the swap-bits-2-3 key map — and that each of its four
    16-lane groups touches 16 distinct 16-byte slots of the 256-byte bank row (conflict-free)."""
    def swap23(r):
        return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)

    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    lds = {}
    for q in range(4):                                   # K: instruction q = 8 rows x 128 B
        for lane in range(64):
            row, phys = 8 * q + (lane >> 3), lane & 7
            lds[q * 1024 + lane * 16] = (row, phys ^ ((row >> 1) & 7))
    assert sorted(lds.values()) == [(r, s) for r in range(32) for s in range(8)]
    keys_of = {}
    for s in range(4):                                   # 16-wide d step s; lane (l31, hh) needs row swap23(l31), slot 2 s + hh
        for g in groups:
            slots = set()
            for l in g:
                l31, hh = l & 31, l >> 5
                kr = swap23(l31)
                byte = kr * 128 + ((2 * s + hh) ^ ((kr >> 1) & 7)) * 16
                assert lds[byte] == (kr, 2 * s + hh)
                slots.add((byte // 16) % 16)
                keys_of[l31] = kr
            assert len(slots) == 16
    # the key map: accumulator register r of lane half hh holds MFMA row (r & 3) + 8 (r >> 2) + 4 hh = key 16 (r >> 3) + 8 hh + (r & 7)
    for hh in range(2):
        for r in range(16):
            assert keys_of[(r & 3) + 8 * (r >> 2) + 4 * hh] == 16 * (r >> 3) + 8 * hh + (r & 7)
    lds = {}
    for q in range(4):                                   # V^T: instruction q = 16 rows x 64 B
        for lane in range(64):
            row, phys = 16 * q + (lane >> 2), lane & 3
            lds[q * 1024 + lane * 16] = (row, phys ^ ((row >> 2) & 3))
    assert sorted(lds.values()) == [(r, s) for r in range(64) for s in range(4)]
    for s in range(2):                                   # 16-key step s: lane (d = 32 dt + l31, hh) needs keys 16 s + 8 hh .. + 7 = slot 2 s + hh
        for dt in range(2):
            for g in groups:
                slots = set()
                for l in g:
                    l31, hh = l & 31, l >> 5
                    d = 32 * dt + l31
                    byte = d * 64 + ((2 * s + hh) ^ ((l31 >> 2) & 3)) * 16
                    assert lds[byte] == (d, 2 * s + hh)
                    slots.add((byte // 16) % 16)
                assert len(slots) == 16


def test_build_mask_device_word_ledger_and_kernel_arithmetic():
    """The device build_mask (csrc/elementwise.hip: vn_build_mask_kernel, masks.build_mask_device) restated in numpy over the raw
    mt19937 words — the same ledger (masks.mask_word_ledger), the same 24-bit bernoulli, `% period` roll, `% T` dropout columns,
    nearest-centre window test — against the host twin of vampnet/mask.py (itself pinned bitwise to the reference), including
    the generator position after the call.  The GPU test holds the kernel to the same masks."""
    import numpy as np
    from vampnet_amd import masks

    def model(raw, B, C, T, rand_mask_intensity=1.0, n_prefix=0, n_suffix=0, periodic_prompt=7, periodic_prompt_width=1,
              onset_mask=None, dropout=0.0, upper_codebook_mask=3, ncc=0):
        period, width = periodic_prompt, periodic_prompt_width
        roll_word, drop_word, n_drop, n_words = masks.mask_word_ledger(B, C, T, period, width, dropout)
        n_lin = B * C * T
        u = (raw[:n_lin] & np.uint64(0xFFFFFF)).astype(np.float32) * np.float32(2.0 ** -24)
        m = (u < np.float32(rand_mask_intensity)).astype(np.int64).reshape(B, C, T)
        t = np.arange(T)
        m[:, :, (t < n_prefix) | (t >= T - n_suffix)] = 0
        if period > 0:
            off = int(raw[roll_word] % np.uint64(period))
            ts = (t - off % T) % T
            hw, c0 = width // 2, (ts // period) * period
            c1 = c0 + period
            m[:, :, ((ts - c0) <= hw) | ((c1 < T) & ((c1 - ts) <= hw))] = 0
        if onset_mask is not None:
            m[onset_mask.numpy() == 0] = 0
        for j in range(n_drop):
            m[:, :, int(raw[drop_word + j] % np.uint64(T))] = 1
        m[:, :(0 if ncc is None else masks._slice_start(ncc, C)), :] = 0
        m[:, masks._slice_start(upper_codebook_mask, C):, :] = 1
        return m, n_words

    cases = [dict(), dict(rand_mask_intensity=0.7, periodic_prompt=5, periodic_prompt_width=3),
             dict(n_prefix=44, n_suffix=69, periodic_prompt=13, periodic_prompt_width=5, upper_codebook_mask=6, ncc=2),
             dict(periodic_prompt=0, dropout=0.3, upper_codebook_mask=14),
             dict(rand_mask_intensity=0.35, periodic_prompt=3, periodic_prompt_width=8, dropout=0.05, ncc=1, upper_codebook_mask=9),
             dict(periodic_prompt=700, upper_codebook_mask=0), dict(upper_codebook_mask=-2, ncc=-13), dict(n_prefix=600), dict(ncc=None)]
    for ci, kw in enumerate(cases):
        for (B, T) in [(1, 575), (3, 173), (2, 64)]:
            z = torch.zeros(B, 14, T, dtype=torch.long)
            onset = None
            if ci in (1, 4):
                onset = (torch.rand(1, 1, T, generator=torch.Generator().manual_seed(ci)) < 0.8).long().expand(B, 14, T)
            torch.manual_seed(50 + ci)
            raw = _mt_raw(50 + ci, B * 14 * T + 4 * B * T + 700)
            ref = masks.build_mask(z, onset_mask=onset, **kw)
            got, n_words = model(raw, B, 14, T, onset_mask=onset, **kw)
            assert np.array_equal(got, ref.numpy()), (ci, B, T)
            nxt = torch.empty(1).uniform_(0, 1).item()                 # the generator stands n_words further
            assert nxt == float(np.float32(int(raw[n_words]) & 0xFFFFFF) * np.float32(2.0 ** -24)), (ci, B, T)


def test_tiled_plane_layout_formula():
    """The tiled operand layout of gemm_x3.hip — vn_tiled_off in csrc/vn_common.h, written by vn_store_planes4 / vn_tile_planes_kernel,
    read by the GEMM's LDS-DMA — against Engine.tile3 (the permutation the GPU tests build tiled operands with): element (row, k)
    of plane q sits at (((row >> 4) (K >> 5) + (k >> 5)) 3 + q) 512 + (row & 15) 32 + (k & 31); a DMA piece (16 rows x 32 k of one
    plane) is 512 consecutive elements = 1 KiB, and a k-tile step is +3 pieces."""
    from vampnet_amd.engine import Engine

    def tiled_off(row, k, K):
        return (((row >> 4) * (K >> 5) + (k >> 5)) * 3) * 512 + (row & 15) * 32 + (k & 31)

    for R, K in [(48, 64), (33, 96), (16, 32), (100, 256)]:
        planes = torch.arange(3 * R * K, dtype=torch.int32).reshape(3, R, K)
        flat = Engine.tile3(planes).reshape(-1)
        R16 = (R + 15) // 16 * 16
        assert flat.numel() == 3 * R16 * K
        for q in range(3):
            for row in sorted({0, 1, 15, min(16, R - 1), R - 1}):
                for k in sorted({0, 1, 31, min(32, K - 1), K - 1}):
                    assert int(flat[tiled_off(row, k, K) + 512 * q]) == int(planes[q, row, k]), (R, K, q, row, k)
        # one piece = the 16 x 32 block of one plane, contiguous
        piece = flat[tiled_off(16 if R > 16 else 0, 32 if K > 32 else 0, K) + 512:][:512].reshape(16, 32)
        r0, k0 = (16 if R > 16 else 0), (32 if K > 32 else 0)
        want = torch.zeros(16, 32, dtype=torch.int32)
        rows = min(16, R - r0)
        want[:rows] = planes[1, r0:r0 + rows, k0:k0 + 32]
        assert torch.equal(piece, want)


def test_tiled_plane_layout_formula_f16x2():
    """the same layout with TWO planes per piece group (vn_tiled_off_np(.., 2): f16x2 operands), against the permutation the GPU
    tests compare Engine.split2h(tiled=True) with"""
    def tiled_off(row, k, K):
        return (((row >> 4) * (K >> 5) + (k >> 5)) * 2) * 512 + (row & 15) * 32 + (k & 31)

    for R, K in [(48, 64), (16, 32), (96, 256)]:
        planes = torch.arange(2 * R * K, dtype=torch.int32).reshape(2, R, K)
        flat = planes.reshape(2, R // 16, 16, K // 32, 32).permute(1, 3, 0, 2, 4).contiguous().reshape(-1)
        for q in range(2):
            for row in sorted({0, 1, 15, min(16, R - 1), R - 1}):
                for k in sorted({0, 1, 31, min(32, K - 1), K - 1}):
                    assert int(flat[tiled_off(row, k, K) + 512 * q]) == int(planes[q, row, k]), (R, K, q, row, k)


def test_mask_rng_self_check_formulas_hold_on_this_torch():
    """The word arithmetic DeviceTorchRng.self_check("mask") compares against torch on the GPU box — bernoulli(p tensor) = 24 bits
    of one word * 2^-24 < p, randint = one word % range, strictly sequential — holds for THIS torch build's CPU generator
    (the stream here comes from numpy's MT19937 loaded with torch's state)."""
    import numpy as np
    from vampnet_amd.torch_rng import parse_torch_rng_state
    torch.manual_seed(5)
    torch.rand(100)
    state, pos = parse_torch_rng_state(torch.get_rng_state())
    bg = np.random.MT19937()
    st = bg.state
    st["state"]["key"] = state.astype(np.uint32)
    st["state"]["pos"] = pos
    bg.state = st
    w = bg.random_raw(60).astype(np.uint32)
    g = torch.Generator()
    g.set_state(torch.get_rng_state())
    p = torch.tensor([0.37, 0.5, 0.93, 1.0] * 12)
    want_b = torch.bernoulli(p, generator=g)
    ranges = (7, 575, 173, 13, 1000, 3) * 2
    want_r = torch.cat([torch.randint(0, r, (1,), generator=g) for r in ranges])
    u24 = (w[:48] & np.uint32(0xFFFFFF)).astype(np.float32) * np.float32(2.0 ** -24)
    assert torch.equal(torch.from_numpy((u24 < p.numpy()).astype(np.float32)), want_b)
    assert torch.equal(torch.from_numpy(w[48:].astype(np.int64) % np.array(ranges, dtype=np.int64)), want_r)


def test_attention_x3_key_split_dma_addressing():
    """Model of the key-split kernel's LDS-DMA (attention_x3.hip: vn_attention_x3_split_kernel): ONE wave stages whole plane tiles
    with four buffer loads per plane whose instruction offset 0 / 1024 / 2048 / 3072 is added on BOTH sides (MUBUF to LDS:
    LDS address = M0 base + inst_offset + lane * 16; memory address = descriptor + per-lane offset + scalar offset + inst_offset) and
    two per-lane K offsets (even / odd piece: the swizzle key (row >> 1) & 7 flips bit 2 with the piece).  The resulting LDS image
    must be the one the shared-tile kernel builds with one piece per wave — the fragment reads are common to both kernels."""
    # shared-tile kernel: wave w, lane -> (LDS byte, source byte inside the tile)
    want_k, want_v = {}, {}
    for w in range(4):
        for lane in range(64):
            krow = 8 * w + (lane >> 3)
            want_k[w * 1024 + lane * 16] = krow * 128 + ((lane & 7) ^ ((krow >> 1) & 7)) * 16
            vrow = 16 * w + (lane >> 2)
            want_v[w * 1024 + lane * 16] = vrow * 64 + ((lane & 3) ^ ((vrow >> 2) & 3)) * 16
    got_k, got_v = {}, {}
    for lane in range(64):
        kr0, vr0 = lane >> 3, lane >> 2
        kvoff_e = (kr0 * 64 + ((lane & 7) ^ ((kr0 >> 1) & 7)) * 8) * 2
        kvoff_o = (kr0 * 64 + ((lane & 7) ^ (((kr0 >> 1) + 4) & 7)) * 8) * 2
        vvoff = (vr0 * 32 + ((lane & 3) ^ ((vr0 >> 2) & 3)) * 8) * 2
        for w in range(4):
            imm = 1024 * w
            got_k[imm + lane * 16] = (kvoff_o if w & 1 else kvoff_e) + imm          # M0 = the plane tile's base for all four pieces
            got_v[imm + lane * 16] = vvoff + imm
    assert got_k == want_k and got_v == want_v
    assert sorted(got_k.values()) == [16 * i for i in range(256)] and sorted(got_v.values()) == [16 * i for i in range(256)]


def test_codec_routing_rule():
    """DacCodec._on_x3: which convolutions of the published 44.1 kHz DAC configuration run on the bf16x3 pipe — the table behind
    DESIGN.md section 3 (>= 128 output channels and K x tile efficiency >= 512; round 6: the 96- / 192-channel layers with taps x C_in >= 512 in
    the channels-on-rows form, bf16x3 operands only), and nothing at all in the f32 pipe."""
    from vampnet_amd.codec import DacCodec
    c = DacCodec.__new__(DacCodec)
    c.precision = "bf16x3"
    conv = lambda cout, cin, k: dict(cout=cout, cin=cin, k=k)
    table = {  # (C_out, C_in, taps): on the bf16x3 pipe ?
        (64, 64, 7): False, (128, 64, 4): False, (128, 128, 7): True, (128, 128, 1): False, (256, 128, 8): True, (256, 256, 7): True,
        (256, 256, 1): False, (512, 256, 16): True, (512, 512, 7): True, (512, 512, 1): True, (1024, 512, 24): True, (1024, 1024, 3): True,
        (1536, 1024, 7): True, (768, 1536, 2): True, (768, 768, 7): True, (768, 768, 1): True, (384, 768, 2): True, (384, 384, 7): True,
        (384, 384, 1): False, (192, 384, 2): True, (192, 192, 7): True, (192, 192, 1): False, (96, 192, 2): False, (96, 96, 7): True,
        (96, 96, 1): False}
    for (cout, cin, k), want in table.items():
        assert c._on_x3(conv(cout, cin, k)) == want, (cout, cin, k)
        assert c._fmt(conv(cout, cin, k)) == ("x3" if want else "f32")
    c.precision = "f16x2"                                     # the channels-on-rows form exists for bf16x3 operands only
    assert not c._on_x3(conv(96, 96, 7)) and c._on_x3(conv(192, 192, 7)) and c._on_x3(conv(384, 384, 7))
    c.precision = "f32"
    assert not any(c._on_x3(conv(*key)) for key in table)


def test_codec_program_arena_planner():
    """vampnet_amd/codec.py _Recorder.plan: the arena offsets of a recorded codec program.  Two buffers may share bytes only if
    their live ranges [first use, last use] are disjoint — an op's inputs are never handed to its outputs — pinned buffers are
    never recycled, and the arena is smaller than the sum of the buffers (it exists to recycle)."""
    import random
    import torch
    from vampnet_amd.codec import _FAKE_BASE, _PTR_IN, _PTR_OUT, _Recorder
    rnd = random.Random(5)
    for trial in range(20):
        rec = _Recorder()
        inp = rec.new((4, 100), torch.float32, io="in")
        out = rec.new((4, 7), torch.int64, io="out")
        assert inp.data_ptr() == _PTR_IN and out.data_ptr() == _PTR_OUT
        live = [inp]
        pinned = None
        for k in range(40):
            reads = rnd.sample(live, min(len(live), rnd.randint(1, 3)))
            w = rec.new((rnd.randint(1, 9), rnd.randint(1, 300)), rnd.choice([torch.float32, torch.bfloat16, torch.float16]))
            v = w.reshape(-1)                                   # a view keeps the buffer's identity
            assert v.data_ptr() == w.data_ptr() and v.shape == (w.numel(),)
            rec.emit(0, [("p", r.data_ptr()) for r in reads] + [("p", None), ("i", 3), ("p", v.data_ptr())])
            live.append(w)
            if len(live) > 5:
                live.pop(rnd.randrange(len(live)))
            if k == 17:
                pinned = w
                rec.bufs[w.bid]["pinned"] = True
        rec.emit(0, [("p", live[-1].data_ptr()), ("p", out.data_ptr())])
        off, total = rec.plan()
        used = [i for i, b in enumerate(rec.bufs) if b["first"] is not None and b["io"] is None]
        assert set(off) == set(used)
        n_ops = len(rec.ops)
        span = lambda i: (rec.bufs[i]["first"], n_ops if rec.bufs[i]["pinned"] else rec.bufs[i]["last"])
        for a in used:
            assert off[a] % 256 == 0 and off[a] + rec.bufs[a]["nbytes"] <= total
            for b in used:
                if a < b:
                    (fa, la), (fb, lb) = span(a), span(b)
                    overlap_time = not (la < fb or lb < fa)
                    overlap_mem = not (off[a] + rec.bufs[a]["nbytes"] <= off[b] or off[b] + rec.bufs[b]["nbytes"] <= off[a])
                    assert not (overlap_time and overlap_mem), (trial, a, b)
        assert total < sum(-(-rec.bufs[i]["nbytes"] // 256) * 256 for i in used)
        assert span(pinned.bid)[1] == n_ops


def test_f16x2_generate_wrapper_repeats_a_saturated_call_from_the_same_generator_state():
    """VampNetModel.generate in precision "f16x2" (host logic, no GPU): the saturation ledger is read after the call; if a word is
    set the result is discarded, the model moves to bf16x3 with a PrecisionFallbackWarning, torch's generator is put back where the
    call started and the call is repeated with the same arguments; a clean ledger returns the first result untouched."""
    import warnings
    import torch
    from vampnet_amd.engine import PrecisionFallbackWarning, VampNetModel

    class _Eng:
        def __init__(self, flags):
            self.flags = list(flags)

        def saturation(self, clear=True):
            return self.flags.pop(0) if self.flags else (0, 0, 0, 0)

    class _Model(VampNetModel):
        def __init__(self, flags):                       # no device: only what the wrapper touches
            self.engine, self.precision, self.calls = _Eng(flags), "f16x2", []

        def set_precision(self, precision):
            self.precision = precision

        def _generate(self, **kw):
            draw = float(torch.rand(1))                  # what a seed-less call would draw its noise / device seed from
            self.calls.append((self.precision, draw, kw["_sampling_steps"], kw["temperature"]))
            return (self.precision, draw)

    torch.manual_seed(123)
    want = float(torch.rand(1))
    after_one_call = torch.get_rng_state()
    # saturated: repeated on bf16x3 from the same generator state, generator left where ONE call leaves it
    m = _Model([(0, 1, 0, 0)])
    torch.manual_seed(123)
    with warnings.catch_warnings(record=True) as wl:
        warnings.simplefilter("always")
        out = m.generate(_sampling_steps=5, temperature=0.7)
    assert [w.category for w in wl] == [PrecisionFallbackWarning] and "attention 1" in str(wl[0].message)
    assert out == ("bf16x3", want) and m.precision == "bf16x3"
    assert m.calls == [("f16x2", want, 5, 0.7), ("bf16x3", want, 5, 0.7)]
    assert torch.equal(torch.get_rng_state(), after_one_call)
    assert m.generate(_sampling_steps=2, temperature=1.0)[0] == "bf16x3" and len(m.calls) == 3      # stays there, no ledger read
    # clean ledger: one call, no warning, precision kept
    m = _Model([(0, 0, 0, 0)])
    torch.manual_seed(123)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out = m.generate(_sampling_steps=5, temperature=0.7)
    assert out == ("f16x2", want) and m.precision == "f16x2" and len(m.calls) == 1


# ----------------------------------------------------------------------------- round 5: host pieces of the new C entries
def test_codec_tensor_table_matches_the_state_dict_names():
    """vn_codec_tensor_name / _offset / vn_codec_weights_size (csrc/codec_plan.hip: what a host without Python fills before
    vn_codec_create_from_weights) enumerate exactly the tensors of a DAC-family state_dict — names, element counts, non-overlapping
    256-byte aligned slots, the level-stacked quantizer regions — and pack_codec_blob places every folded tensor where the table says."""
    from oracle import dac_oracle as D
    from vampnet_amd.codec import codec_cfg_struct, pack_codec_blob, _fold
    lib = _lib.load()
    for cfg in (D.DAC_TINY_CFG, D.DAC_DEFAULT_CFG):
        sd = D.synth_dac_state_dict(cfg, 0)
        cs = codec_cfg_struct(cfg)
        n, cnt = C.c_int64(), C.c_int()
        assert lib.vn_codec_weights_size(C.byref(cs), C.byref(n)) == 0 and lib.vn_codec_tensor_count(C.byref(cs), C.byref(cnt)) == 0
        name = C.create_string_buffer(256)
        off, num = C.c_int64(), C.c_int64()
        seen, spans = set(), []
        for i in range(cnt.value):
            assert lib.vn_codec_tensor_name(C.byref(cs), i, name, 256, C.byref(off), C.byref(num)) == 0
            key = name.value.decode()
            t = _fold(sd, key[:-7]) if key.endswith(".weight") and key not in sd else sd[key]
            assert t.numel() == num.value, key
            o2, n2 = C.c_int64(), C.c_int64()
            assert lib.vn_codec_tensor_offset(C.byref(cs), key.encode(), C.byref(o2), C.byref(n2)) == 0 and (o2.value, n2.value) == (off.value, num.value)
            seen.add(key)
            spans.append((off.value, off.value + num.value))
        folded = {k[:-9] + ".weight" if k.endswith(".weight_v") else k for k in sd if not k.endswith(".weight_g")}
        assert seen == folded, seen ^ folded
        spans.sort()
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= n.value
        lat = cfg["latent_dim"] or cfg["encoder_dim"] * 2 ** len(cfg["encoder_rates"])
        o0, o1 = C.c_int64(), C.c_int64()
        lib.vn_codec_tensor_offset(C.byref(cs), b"quantizer.quantizers.0.in_proj.bias", C.byref(o0), C.byref(num))
        lib.vn_codec_tensor_offset(C.byref(cs), b"quantizer.quantizers.1.in_proj.bias", C.byref(o1), C.byref(num))
        assert o1.value - o0.value == cfg["codebook_dim"]                       # stacked level after level: the RVQ kernels' layout
        blob = pack_codec_blob(lib, cs, sd)
        lib.vn_codec_tensor_offset(C.byref(cs), b"decoder.model.0.weight", C.byref(off), C.byref(num))
        assert torch.equal(blob[off.value:off.value + num.value], _fold(sd, "decoder.model.0").reshape(-1))
        assert num.value == cfg["decoder_dim"] * lat * 7
    bad = codec_cfg_struct(D.DAC_TINY_CFG)
    bad.n_rates = 0
    assert lib.vn_codec_weights_size(C.byref(bad), C.byref(n)) != 0


def test_kweighting_state_space_power_reproduces_lfilter():
    """csrc/preprocess.hip evaluates the K-weighting IIR cascade chunk by chunk: S_(j+1) = A^CH S_j + e_j.  The host side of that
    (codec.kweight_state_power) against scipy.signal.lfilter over the whole signal, in float64."""
    from scipy.signal import lfilter
    from vampnet_amd.codec import _k_weighting, kweight_state_power
    sr, CH = 44100, 441
    x = np.random.default_rng(0).standard_normal(5 * CH)
    want = x.copy()
    for b, a in _k_weighting(sr):
        want = lfilter(b, a, want)
    (b1, a1), (b2, a2) = _k_weighting(sr)

    def run(s, xs):
        s, out = list(s), []
        for xv in xs:
            y1 = b1[0] * xv + s[0]
            s[0], s[1] = b1[1] * xv - a1[1] * y1 + s[1], b1[2] * xv - a1[2] * y1
            y2 = b2[0] * y1 + s[2]
            s[2], s[3] = b2[1] * y1 - a2[1] * y2 + s[3], b2[2] * y1 - a2[2] * y2
            out.append(y2)
        return np.array(out), np.array(s)
    P = kweight_state_power(sr, CH)
    S, got = np.zeros(4), []
    for j in range(5):
        chunk = x[j * CH:(j + 1) * CH]
        e = run(np.zeros(4), chunk)[1]
        got.append(run(S, chunk)[0])
        S = P @ S + e
    assert np.abs(np.concatenate(got) - want).max() < 1e-10


def test_pack_lora_vector_layout():
    """engine.pack_lora_vector: loralib tensors -> the adapter vector vn_model_apply_lora reads (At [in][8] then B [out][8] per layer
    and LoRA'd linear, w_1's B rows in the packed value / gate order of VN_W_W1); absent adapters stay zero."""
    from vampnet_amd.engine import LORA_KEYS, pack_lora_vector
    from vampnet_amd._lib import vn_dims
    lib = _lib.load()
    d = W.TINY_COARSE_DIMS
    dims = vn_dims(d["n_layers"], d["n_heads"], d["d_model"], d["n_codebooks"], d["n_cond"], 1024, 8, 32, 128, 1e-6, 2, 64)
    D_ = d["d_model"]
    g = torch.Generator().manual_seed(1)
    sd = {"transformer.layers.1.feed_forward.w_1.lora_A": torch.randn(8, D_, generator=g),
          "transformer.layers.1.feed_forward.w_1.lora_B": torch.randn(4 * D_, 8, generator=g),
          "transformer.layers.0.self_attn.w_vs.lora_A": torch.randn(8, D_, generator=g),
          "transformer.layers.0.self_attn.w_vs.lora_B": torch.randn(D_, 8, generator=g)}
    vec = pack_lora_vector(lib, dims, sd)
    off, cnt = C.c_int64(), C.c_int64()

    def slot(l, w, ab):
        assert lib.vn_lora_param_offset(C.byref(dims), l, w, ab, C.byref(off), C.byref(cnt)) == 0
        return vec[off.value:off.value + cnt.value]
    w1 = LORA_KEYS.index("feed_forward.w_1")
    assert torch.equal(slot(1, w1, 0).view(D_, 8), sd["transformer.layers.1.feed_forward.w_1.lora_A"].t())
    b = sd["transformer.layers.1.feed_forward.w_1.lora_B"]
    packed = torch.stack([b[:2 * D_].view(2 * D_ // 32, 32, 8), b[2 * D_:].view(2 * D_ // 32, 32, 8)], dim=1).reshape(4 * D_, 8)
    assert torch.equal(slot(1, w1, 1).view(4 * D_, 8), packed)
    wv = LORA_KEYS.index("self_attn.w_vs")
    assert torch.equal(slot(0, wv, 1).view(D_, 8), sd["transformer.layers.0.self_attn.w_vs.lora_B"])
    used = sum(v.numel() for v in sd.values())
    assert int((vec != 0).sum()) == used                                          # every other adapter slot is zero: w + 0 on merge
    with pytest.raises(_lib.VnError):
        pack_lora_vector(lib, dims, {"transformer.layers.0.self_attn.fc.lora_A": torch.zeros(4, D_),
                                     "transformer.layers.0.self_attn.fc.lora_B": torch.zeros(D_, 4)})


def test_interface_resident_model_lru(tmp_path):
    """Interface's resident-model cache (what makes reload() to an earlier checkpoint a lookup): keyed by path + mtime + size, least
    recently used entries dropped beyond max_resident"""
    from vampnet_amd.interface import Interface
    itf = object.__new__(Interface)
    paths = []
    for i in range(6):
        p = tmp_path / f"m{i}.pth"
        p.write_bytes(b"x" * (i + 1))
        paths.append(p)
    os.environ["VN_RESIDENT_MODELS"] = "3"
    try:
        for i, p in enumerate(paths[:3]):
            itf._resident_put("coarse", p, f"model{i}")
    finally:
        del os.environ["VN_RESIDENT_MODELS"]
    assert itf._resident_get("coarse", paths[0]) == "model0"                      # refreshes m0
    itf._resident_put("coarse", paths[3], "model3")                               # evicts the least recently used: m1
    assert itf._resident_get("coarse", paths[1]) is None
    assert itf._resident_get("coarse", paths[0]) == "model0" and itf._resident_get("coarse", paths[3]) == "model3"
    assert itf._resident_get("c2f", paths[0]) is None                             # roles are separate
    paths[0].write_bytes(b"changed")                                              # a rewritten file is another model
    assert itf._resident_get("coarse", paths[0]) is None


@pytest.mark.parametrize("T,nb,md", [(37, 32, 128), (64, 32, 128), (575, 32, 128), (1024, 32, 128), (300, 16, 64), (130, 64, 512)])
def test_attention_backward_bias_tables_cover_every_mixed_bucket_tile(T, nb, md):
    """Host side of the split-plane attention backward (csrc/attention_train_x3.hip): the bucket LUT the kernels index with
    key - query + T - 1 is the oracle's relative_position_bucket, and near_r — the half-width of the per-wave bias-gradient tables —
    covers every 32 x 32 wave tile that is NOT summed in registers: a tile is "far" when its whole offset range lies on one side of the
    diagonal inside ONE bucket; for every other tile all offsets of real (query, key) pairs must satisfy |key - query| <= near_r."""
    lib = _lib.load()
    lut = np.zeros(2 * T - 1, dtype=np.int32)
    nr = C.c_int()
    assert lib.vn_attention_bwd_table_span(T, nb, md, lut.ctypes.data_as(C.c_void_p), C.byref(nr)) == 0
    rel = torch.arange(-(T - 1), T)
    assert np.array_equal(lut, O.relative_position_bucket(rel, nb, md).numpy().astype(np.int32))
    near_r = nr.value
    assert 0 < near_r <= T - 1
    for q0 in range(0, T, 32):
        for k0 in range(0, T, 32):
            lo, hi = k0 - (q0 + 31), k0 + 31 - q0
            lo_c, hi_c = max(lo, -(T - 1)), min(hi, T - 1)
            far = (lo > 0 or hi < 0) and lut[lo_c + T - 1] == lut[hi_c + T - 1]
            if far:
                assert len(set(lut[lo_c + T - 1:hi_c + T])) == 1        # monotone on each side: the ends decide the whole range
                continue
            qs = np.arange(q0, min(q0 + 32, T))[:, None]
            ks = np.arange(k0, min(k0 + 32, T))[None, :]
            assert np.abs(ks - qs).max() <= near_r, (q0, k0, near_r)
