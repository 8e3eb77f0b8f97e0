"""-m gpu: each HIP kernel through the C ABI vs the CPU oracle (fp32, seeded)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import vampnet_oracle as O, weights as W

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from vampnet_amd.engine import Engine
    return Engine("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


@pytest.mark.parametrize("rows,D", [(4600, 1280), (575, 1280), (7, 256), (3, 1000), (1, 1280)])
def test_rmsnorm(eng, rows, D):
    x, w = _rand((rows, D), 1, 3.0), 1 + _rand((D,), 2, 0.1)
    got = eng.rmsnorm(x.cuda(), w.cuda()).cpu()
    ref = O.rmsnorm(x, w)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("rows,D,ns", [(4600, 1280, 2), (575, 1280, 4), (601, 256, 2), (3, 256, 3), (1, 1280, 2)])
def test_splitk_reduce_rmsnorm_fused_vs_two_kernels(eng, rows, D, ns):
    """vn_splitk_reduce_rmsnorm_kernel (the reduce pass of a split RESIDUAL GEMM that also runs the next RMSNorm) against the
    reduce kernel followed by the norm kernel: x and the three y planes bitwise; and against torch within fp32 rounding."""
    part, x0, w = _rand((ns, rows, D), 11, 0.7), _rand((rows, D), 12, 2.0), 1 + _rand((D,), 13, 0.1)
    outs = []
    part_d, w_d = part.cuda(), w.cuda()             # keep the device copies alive across the launches
    for fused in (1, 0):
        x = x0.cuda().clone()
        y16 = torch.zeros(3, rows, D, dtype=torch.bfloat16, device="cuda")
        eng.check(eng.lib.vn_debug_splitk_reduce_rmsnorm(eng.handle, part_d.data_ptr(), ns, x.data_ptr(), w_d.data_ptr(),
                                                         y16.data_ptr(), rows * D, rows, D, 1e-6, fused, eng.stream()), "reduce_rmsnorm")
        outs.append((x.cpu(), y16.cpu()))
    (xf, yf), (xs, ys) = outs
    dx = (xf - xs).abs().max().item()
    dy = (yf.float().sum(0) - ys.float().sum(0)).abs().max().item()
    print(f"fused vs two kernels: max |dx| = {dx:.3e}, max |dy| = {dy:.3e}")
    assert torch.equal(xf, xs)
    assert torch.equal(yf.view(torch.int16), ys.view(torch.int16))
    xr = x0.clone()
    acc = part[0].clone()
    for sp in range(1, ns):
        acc += part[sp]
    xr = acc + xr                                  # the reduce's order: images first, then the residual
    assert torch.equal(xf, xr)
    yr = O.rmsnorm(xr, w)
    np.testing.assert_allclose(yf.float().sum(0).numpy(), yr.numpy(), rtol=2e-6, atol=1e-6)
    if D % 32 == 0:                                    # the tiled plane layout of the model path: the same planes, re-laid
        for fused in (1, 0):
            x = x0.cuda().clone()
            yt = torch.zeros(3 * ((rows + 15) // 16 * 16) * D, dtype=torch.bfloat16, device="cuda")
            eng.check(eng.lib.vn_debug_splitk_reduce_rmsnorm(eng.handle, part_d.data_ptr(), ns, x.data_ptr(), w_d.data_ptr(),
                                                             yt.data_ptr(), -1, rows, D, 1e-6, fused, eng.stream()), "reduce_rmsnorm tiled")
            want = eng.tile3(yf.cuda()).reshape(-1).view(torch.int16)
            got = yt.view(torch.int16)
            live = eng.tile3(torch.ones(3, rows, D, dtype=torch.bfloat16, device="cuda")).reshape(-1) != 0     # padded rows are not written
            assert torch.equal(got[live], want[live])
    assert torch.equal(yf.double().sum(0).float(), yf.float().sum(0))       # planes: exact three-way split of an fp32 value


GEMM_SHAPES = [(575, 1280, 1280), (4600, 3840, 1280), (4600, 1280, 2560), (1384, 10240, 1280),
               (50, 256, 256), (1, 128, 64), (129, 192, 96), (64, 64, 32), (700, 4096, 256)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_store_bias_residual(eng, M, N, K):
    from vampnet_amd import _lib
    a, w, b = _rand((M, K), 3), _rand((N, K), 4, 1.0 / np.sqrt(K)), _rand((N,), 5)
    ref64 = a.double() @ w.double().t()
    absdot = a.abs().double() @ w.abs().double().t()
    tol = (2e-6 * absdot + 1e-6).numpy()       # fp32 accumulation-order class (guide: ~1e-7 * sum|a b| at K<=1024)
    A, Wd = a.cuda(), w.cuda()
    got = eng.gemm(A, Wd).cpu().double()
    assert np.all(np.abs((got - ref64).numpy()) <= tol)
    got = eng.gemm(A, Wd, bias=b.cuda(), epilogue=_lib.EPI_BIAS).cpu().double()
    assert np.all(np.abs((got - (ref64 + b.double())).numpy()) <= tol)
    c0 = _rand((M, N), 6)
    out = c0.cuda().clone()
    eng.gemm(A, Wd, epilogue=_lib.EPI_RESIDUAL, out=out)
    assert np.all(np.abs((out.cpu().double() - (ref64 + c0.double())).numpy()) <= tol)


def test_gemm_identity_asymmetric(eng):
    """A = I catches a transposed C write (guide §3: always check with asymmetric B)."""
    K = N = 256
    a = torch.eye(K)[:200]
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 1e-3
    got = eng.gemm(a.cuda(), w.cuda()).cpu()
    assert torch.equal(got, w.t()[:200])


@pytest.mark.parametrize("M,D", [(575, 1280), (4600, 1280), (33, 256)])
def test_gemm_geglu(eng, M, D):
    from vampnet_amd import _lib
    x, w1 = _rand((M, D), 7), _rand((4 * D, D), 8, 1.0 / np.sqrt(D))
    ref = O.gated_gelu(torch.nn.functional.linear(x, w1))
    val, gate = w1[:2 * D].reshape(2 * D // 32, 32, D), w1[2 * D:].reshape(2 * D // 32, 32, D)
    w1p = torch.stack([val, gate], dim=1).reshape(4 * D, D)
    got = eng.gemm(x.cuda(), w1p.cuda(), epilogue=_lib.EPI_GEGLU).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


def _attention_ref(q, k, v, table):
    B, H, T, _ = q.shape
    bias = O.compute_bias(table, T)                      # (H,1,T,T)
    qh, kh, vh = (t.permute(1, 0, 2, 3) for t in (q, k, v))
    attn = torch.einsum("hblk,hbtk->hblt", qh, kh) / np.sqrt(64)
    attn = torch.softmax(attn + bias, dim=3)
    out = torch.einsum("hblt,hbtv->hblv", attn, vh)
    return out.permute(1, 2, 0, 3).reshape(B, T, H * 64)


# attention_x3.hip has two work decompositions: 128-query blocks that share their K / V^T tiles ("x3/shared") and 32-query blocks
# whose KS waves walk disjoint key tiles and merge through LDS ("x3/ks1", "x3/ks2", "x3/ks4"), and 64-query blocks of eight waves = two
# query sub-blocks x four key parts sharing their tiles ("x3/pair": the one- / two-sequence shape); "bf16x3" = the launcher's own choice
ATTN_FORMS = {"f32": ("f32", None), "bf16x3": ("bf16x3", -1), "x3/shared": ("bf16x3", 0), "x3/ks1": ("bf16x3", 1),
              "x3/ks2": ("bf16x3", 2), "x3/ks4": ("bf16x3", 4), "x3/pair": ("bf16x3", 8),
              "f16x2": ("f16x2", -1), "h2/shared": ("f16x2", 0), "h2/ks1": ("f16x2", 1), "h2/ks2": ("f16x2", 2), "h2/ks4": ("f16x2", 4),
              "h2/pair": ("f16x2", 8)}


def _attention(eng, form, q, k, v, table):
    precision, split = ATTN_FORMS[form]
    if split is None:
        return eng.attention(q, k, v, table, precision=precision)
    eng.check(eng.lib.vn_debug_attention_x3_config(eng.handle, split, 0, -1, None), "vn_debug_attention_x3_config")
    try:
        return eng.attention(q, k, v, table, precision=precision)
    finally:
        eng.check(eng.lib.vn_debug_attention_x3_config(eng.handle, -1, 0, -1, None), "vn_debug_attention_x3_config")


@pytest.mark.parametrize("form", list(ATTN_FORMS))
@pytest.mark.parametrize("B,H,T", [(1, 20, 575), (2, 20, 173), (3, 4, 64), (2, 2, 1), (1, 3, 65), (1, 2, 130), (2, 1, 600),
                                   (1, 2, 32), (1, 1, 33), (2, 3, 128), (1, 2, 129), (1, 1, 97), (3, 2, 31)])
def test_attention(eng, B, H, T, form):
    """the fp32-input MFMA kernel and every decomposition of the bf16x3 kernel (six bf16-MFMA products of exact splits) at the
    SAME tolerance; the T values put the end of the sequence at every position of a 32-key tile and of a 32 / 64 / 128-query
    block, and T = 1 / 31 / 32 leave key-split waves without a tile (their empty partial result must merge as zero weight)"""
    q, k, v = _rand((B, H, T, 64), 10), _rand((B, H, T, 64), 11), _rand((B, H, T, 64), 12)
    table = _rand((32, H), 13)
    ref = _attention_ref(q, k, v, table)
    got = _attention(eng, form, q.cuda(), k.cuda(), v.cuda(), table.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=2e-5, atol=3e-6)


def test_attention_x3_decompositions_agree(eng):
    """shared tiles vs key-split: the same products in a different summation order (per-wave partial sums merged at the end,
    a per-wave softmax reference) -> equal to fp32 re-association noise, and each form is run-to-run bitwise reproducible
    (the merge order is fixed)."""
    B, H, T = 2, 20, 575
    q, k, v = (_rand((B, H, T, 64), s).cuda() for s in (40, 41, 42))
    table = _rand((32, H), 43).cuda()
    outs = {}
    for fam in ("x3", "h2"):
        for form in (f"{fam}/shared", f"{fam}/ks1", f"{fam}/ks2", f"{fam}/ks4", f"{fam}/pair"):
            outs[form] = _attention(eng, form, q, k, v, table)
            assert torch.equal(outs[form], _attention(eng, form, q, k, v, table)), form
        for form in (f"{fam}/ks1", f"{fam}/ks2", f"{fam}/ks4", f"{fam}/pair"):
            d = (outs[form] - outs[f"{fam}/shared"]).abs().max().item()
            print(f"{form} vs shared tiles: max |d| = {d:.3e}")
            assert d < 2e-6
    d = (outs["h2/shared"] - outs["x3/shared"]).abs().max().item()
    print(f"f16x2 vs bf16x3 operands, shared tiles: max |d| = {d:.3e}")
    assert d < 4e-6


def test_attention_bf16x3_forced_rescale_and_large_scores(eng):
    """online-softmax edge cases of attention_x3.hip against float64: a key late in the sequence that dominates one query
    (running max jumps at the last tiles: every earlier partial sum is rescaled by ~e^-40), scores of large magnitude, and
    transpose-detecting structure (q, k, v asymmetric)."""
    B, H, T = 1, 2, 200
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(B, H, T, 64, generator=g) for _ in range(3))
    k[0, 0, 190] = q[0, 0, 7] * 5.0                       # q7 . k190 / 8 ~ 40 above everything else
    k[0, 1, 3] = q[0, 1, 150] * 4.0
    q[0, 1, 20] *= 6.0
    # a score ramp for one query: +~2 per 32-key tile, ~12 over the sequence — every tile's growth stays under the deferred-max
    # threshold of attention_x3.hip (P > 1 for a while), the accumulated growth does not (the reference must move eventually)
    k[0, 0, :, :] += q[0, 0, 9][None, :] * (torch.arange(T, dtype=torch.float32) / T * 1.5)[:, None]
    table = torch.randn(32, H, generator=g)
    s = torch.einsum("bhld,bhtd->bhlt", q.double(), k.double()) / 8.0 + O.compute_bias(table, T).permute(1, 0, 2, 3).double()
    ref = torch.einsum("bhlt,bhtd->bhld", torch.softmax(s, -1), v.double()).permute(0, 2, 1, 3).reshape(B, T, H * 64)
    for form in ATTN_FORMS:
        got = _attention(eng, form, q.cuda(), k.cuda(), v.cuda(), table.cuda()).cpu().double()
        err = (got - ref).abs().max().item()
        print(f"{form}: max |err| vs float64 = {err:.3e}")
        assert err < 2e-5


@pytest.mark.parametrize("form", ["f32", "x3/shared", "x3/ks2", "h2/shared", "h2/ks2"])
def test_attention_bias_buckets_exact(eng, form):
    """q = 0 -> scores are the bias alone; v = one-hot(position) -> output row = softmax(bias) itself,
    which pins every bucket boundary (rel = -574..574) against the oracle table."""
    T, H = 575, 2
    q = torch.zeros(1, H, T, 64)
    k = _rand((1, H, T, 64), 14)
    table = _rand((32, H), 15, 3.0)
    ref_soft = torch.softmax(O.compute_bias(table, T)[:, 0], dim=-1)      # (H, T, T)
    for blk in range(0, T, 64):
        v = torch.zeros(1, H, T, 64)
        n = min(64, T - blk)
        v[0, :, blk:blk + n, :n] = torch.eye(n)
        got = _attention(eng, form, q.cuda(), k.cuda(), v.cuda(), table.cuda()).cpu().reshape(T, H, 64)
        want = ref_soft[:, :, blk:blk + n].permute(1, 0, 2)
        np.testing.assert_allclose(got[:, :, :n].numpy(), want.numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("M,N,K", [(4600, 3840, 1280), (575, 1280, 2560), (130, 256, 64), (1384, 10240, 1280)])
def test_gemm_bf16_fast_mode(eng, M, N, K):
    """bf16-operand GEMM (fast mode): exact products of bf16 inputs, fp32 accumulation -> compare with the fp64 product
    of the SAME bf16-rounded operands (tolerance = fp32 accumulation order only)."""
    from vampnet_amd import _lib
    a16 = _rand((M, K), 3).to(torch.bfloat16)
    w16 = (_rand((N, K), 4) / np.sqrt(K)).to(torch.bfloat16)
    ref = a16.double() @ w16.double().t()
    absdot = a16.abs().double() @ w16.abs().double().t()
    tol = (2e-6 * absdot + 1e-6).numpy()
    got = eng.gemm_bf16(a16.cuda(), w16.cuda()).cpu().double()
    assert np.all(np.abs((got - ref).numpy()) <= tol)
    c0 = _rand((M, N), 6)
    out = c0.cuda().clone()
    eng.gemm_bf16(a16.cuda(), w16.cuda(), epilogue=_lib.EPI_RESIDUAL, out=out)
    assert np.all(np.abs((out.cpu().double() - (ref + c0.double())).numpy()) <= tol)


# ----------------------------------------------------------------------------- training kernels
@pytest.mark.parametrize("entry", ["vn_attention_train_f32", "vn_attention_train_bf16x3"])
@pytest.mark.parametrize("B,H,T,p", [(1, 2, 575, 0.1), (2, 3, 37, 0.0), (1, 1, 64, 0.1), (2, 2, 130, 0.1), (3, 2, 291, 0.1)])
def test_attention_train_fwd_bwd_vs_autograd(eng, B, H, T, p, entry):
    """vn_attention_train_fwd / _bwd_dq / _bwd_dkv (fp32-input MFMA) and their split-plane counterparts (attention_x3.hip TRAIN forward,
    attention_train_x3.hip backward) against torch CPU autograd of transformer.py:234-254 with the engine's own dropout keep-mask
    injected.  Tolerances: out 2e-6 abs, gradients 2e-5 of their max-abs."""
    import ctypes as C
    from oracle import vampnet_oracle as O
    lib = eng.lib
    g = torch.Generator().manual_seed(B * 1000 + T)
    q, k, v = (torch.randn(B, H, T, 64, generator=g) for _ in range(3))
    dout = torch.randn(B, T, H * 64, generator=g)
    rel = torch.randn(32, H, generator=g) * 0.5
    dev = lambda t: t.cuda().contiguous()
    qd, kd, vd, dd, rd = dev(q), dev(k), dev(v), dev(dout), dev(rel)
    out = torch.empty(B, T, H * 64, device="cuda")
    lse = torch.empty(B, H, T, device="cuda")
    dqkv = torch.zeros(B * T, 3 * H * 64, device="cuda")
    dbias = torch.zeros(32, H, device="cuda")
    seed = 77
    eng.check(getattr(lib, entry)(eng.handle, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), rd.data_ptr(),
                                  out.data_ptr(), lse.data_ptr(), dd.data_ptr(), dqkv.data_ptr(), dbias.data_ptr(),
                                  B, H, T, 32, 128, p, seed, eng.stream()), entry)
    keep = torch.ones(B * H * T, T)
    if p > 0:
        m8 = torch.empty(B * H * T, T, dtype=torch.uint8, device="cuda")
        eng.check(lib.vn_dropout_keep_mask(eng.handle, seed, 1, 0, 0, p, 0, B * H * T, T, m8.data_ptr(), eng.stream()),
                     "vn_dropout_keep_mask")
        keep = m8.cpu().float()
    keep = keep.view(B, H, T, T)
    qa, ka, va, ra = (t.clone().requires_grad_(True) for t in (q, k, v, rel))
    bias = O.compute_bias(ra, T).permute(1, 0, 2, 3)               # (H, 1, T, T) -> (1, H, T, T)
    s = torch.einsum("bhld,bhtd->bhlt", qa, ka) / 8.0 + bias
    pr = torch.softmax(s, dim=-1) * keep * (1.0 / (1.0 - p))
    o = torch.einsum("bhlt,bhtd->bhld", pr, va).permute(0, 2, 1, 3).reshape(B, T, H * 64)
    o.backward(dout)
    assert (out.cpu() - o.detach()).abs().max().item() < 2e-6 * max(1.0, o.abs().max().item())
    lse_ref = torch.logsumexp(s.detach(), dim=-1)
    assert (lse.cpu() - lse_ref).abs().max().item() < 1e-5
    got = dqkv.cpu().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)   # [3][B][H][T][64]
    for name, a, ref in (("dq", got[0], qa.grad), ("dk", got[1], ka.grad), ("dv", got[2], va.grad)):
        err = (a - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, (name, err)
    err = (dbias.cpu() - ra.grad).abs().max().item() / ra.grad.abs().max().item()
    assert err < 5e-5, ("dbias", err)


@pytest.mark.parametrize("R,C", [(4600, 1280), (37, 256), (130, 4096), (64, 64)])
def test_transpose_zero_pads(eng, R, C):
    ldd = (R + 31) // 32 * 32
    src = torch.randn(R, C, device="cuda")
    dst = torch.full((C, ldd), float("nan"), device="cuda")
    eng.check(eng.lib.vn_transpose_f32(eng.handle, src.data_ptr(), dst.data_ptr(), R, C, ldd, eng.stream()),
                 "vn_transpose_f32")
    assert torch.equal(dst[:, :R], src.t())
    assert bool((dst[:, R:] == 0).all())


@pytest.mark.parametrize("B,T,cin,cout,k,dil", [(2, 30001, 96, 96, 7, 3),      # 128x96 tile (4x1 waves), ragged rows
                                                 (2, 30001, 96, 96, 1, 1),      # k = 1 + residual (ResidualUnit tail)
                                                 (1, 5000, 192, 96, 7, 1),      # few tiles: 64-wide fallback for C_out = 96
                                                 (2, 9000, 64, 192, 3, 1)])
def test_conv1d_vs_torch(eng, B, T, cin, cout, k, dil):
    """vn_conv1d_f32 (channels-last implicit GEMM, fused bias + residual + next-layer Snake) against torch conv1d on the
    CPU; covers every N-tile variant incl. the exact 96-wide tile of the audio-rate decoder layers.  Tolerance 2e-5 of
    the output scale (fp32, different summation order)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    x = torch.randn(B, T, cin, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    bias = torch.randn(cout, generator=g) * 0.1
    alpha = torch.rand(cout, generator=g) + 0.5
    resid = torch.randn(B, T, cout, generator=g) if k == 1 else None
    pad = dil * (k - 1) // 2
    ref = F.conv1d(x.permute(0, 2, 1), w, bias, dilation=dil, padding=pad).permute(0, 2, 1)
    if resid is not None:
        ref = ref + resid
    ref2 = ref + (1.0 / (alpha + 1e-9)) * torch.sin(alpha * ref) ** 2
    xd, wd = x.cuda(), w.permute(0, 2, 1).contiguous().cuda()          # weights packed [C_out][taps][C_in]
    y = torch.empty(B, T, cout, device="cuda")
    y2 = torch.empty(B, T, cout, device="cuda")
    rd = resid.cuda() if resid is not None else None
    bd, ad = bias.cuda(), alpha.cuda()                                  # keep the device copies alive across the launch
    eng.check(eng.lib.vn_conv1d_f32(eng.handle, xd.data_ptr(), wd.data_ptr(), bd.data_ptr(),
                                    rd.data_ptr() if rd is not None else None, ad.data_ptr(), y.data_ptr(),
                                    y2.data_ptr(), None, 0, B, T, T, T, cin, cout, k, 1, dil, pad, 1, 0, 0, eng.stream()),
              "vn_conv1d_f32")
    scale = ref.abs().max().item()
    assert (y.cpu() - ref).abs().max().item() < 2e-5 * scale
    assert (y2.cpu() - ref2).abs().max().item() < 2e-5 * ref2.abs().max().item()


@pytest.mark.parametrize("B,H,T", [(1, 2, 575), (2, 3, 173), (1, 1, 37)])
def test_attention_bf16_fast_mode(eng, B, H, T):
    """vn_attention_bf16 (bf16 MFMA products, fp32 softmax) vs the fp32 formula: bf16-class error only (inputs and
    probabilities carry 8 mantissa bits): max |err| <= 3e-2 of the output scale, mean <= 4e-3."""
    from oracle import vampnet_oracle as O
    g = torch.Generator().manual_seed(T)
    q, k, v = (torch.randn(B, H, T, 64, generator=g) for _ in range(3))
    rel = torch.randn(32, H, generator=g) * 0.5
    qd, kd, vd, rd = q.cuda(), k.cuda(), v.cuda(), rel.cuda()
    out = torch.empty(B, T, H * 64, device="cuda", dtype=torch.bfloat16)
    eng.check(eng.lib.vn_attention_bf16(eng.handle, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), rd.data_ptr(), out.data_ptr(),
                                        B, H, T, 32, 128, eng.stream()), "vn_attention_bf16")
    bias = O.compute_bias(rel, T).permute(1, 0, 2, 3)
    s = torch.einsum("bhld,bhtd->bhlt", q, k) / 8.0 + bias
    ref = torch.einsum("bhlt,bhtd->bhld", torch.softmax(s, -1), v).permute(0, 2, 1, 3).reshape(B, T, H * 64)
    err = (out.float().cpu() - ref).abs()
    scale = ref.abs().max().item()
    print(f"bf16 attention T={T}: max err {err.max().item():.3e}, mean {err.mean().item():.3e}, scale {scale:.2f}")
    assert err.max().item() <= 3e-2 * scale and err.mean().item() <= 4e-3 * scale


def test_torch_rng_on_device(eng):
    """csrc/torch_rng.hip continues torch's CPU mt19937 stream on the GPU: exponential_ / uniform_ tensors of one sampling step
    (2300 x 1024, then 2300) are bit-identical to what torch draws on the host, across block boundaries and across two
    consecutive hand-overs; skipping words equals drawing and discarding; torch's generator ends in the same state."""
    from vampnet_amd.torch_rng import DeviceTorchRng
    rng = DeviceTorchRng(eng)                                # direct use: everything on the current stream
    torch.manual_seed(4242)
    _ = torch.rand(37)                                       # start mid-block
    blob = torch.get_rng_state()
    ref_e = torch.empty(2300, 1024).exponential_(1)
    ref_u = torch.zeros(2300).uniform_(1e-20, 1)
    ref_e2 = torch.empty(5, 1024).exponential_(1)
    ref_tail = torch.rand(8)
    end_state = torch.get_rng_state()

    torch.set_rng_state(blob)
    rng.load_from_torch()
    e = rng.exponential_(torch.empty(2300, 1024, device="cuda"))
    u = rng.uniform_(torch.empty(2300, device="cuda"), 1e-20, 1.0)
    rng.store_to_torch()                                     # hand back ...
    rng.load_from_torch()                                    # ... and over again
    e2 = rng.exponential_(torch.empty(5, 1024, device="cuda"))
    rng.store_to_torch()
    n_bad = int((e.cpu() != ref_e).sum())
    assert n_bad == 0, f"{n_bad} of {ref_e.numel()} exponentials differ"
    assert torch.equal(u.cpu(), ref_u) and torch.equal(e2.cpu(), ref_e2)
    assert torch.equal(torch.rand(8), ref_tail)              # host generator continues exactly where the reference's would
    assert torch.equal(torch.get_rng_state()[:5016], end_state[:5016])
    # skip == draw and discard
    torch.set_rng_state(blob)
    rng.load_from_torch()
    rng.skip(2 * 2300 * 1024)
    u2 = rng.uniform_(torch.empty(2300, device="cuda"), 1e-20, 1.0)
    assert torch.equal(u2.cpu(), ref_u)


@pytest.mark.parametrize("B,N,b0,nb,pattern", [(2, 1150, 0, 2, (1, 1, 1)),          # 4.7 M words per step: three chunks of 2 M
                                                 (3, 300, 1, 1, (1, 0, 1, 1)),        # a rank's rows of a sharded batch, a non-sampling step
                                                 (1, 37, 0, 1, (1, 1)),               # tiny: one short chunk per step
                                                 (2, 64, 2, 0, (1, 0, 1))])           # a rank WITHOUT items (world > batch): only the generator moves
def test_torch_rng_whole_call_plan_equals_torch(eng, B, N, b0, nb, pattern):
    """torch_rng.draw_units — every draw of a whole generate() call planned from ONE generator state (two-level jump-ahead, all chunks
    walked at once) — gives, step for step, the tensors torch draws on the host in the reference's order (exponential_ over (B N, V) when
    the step samples, then uniform_ over (B, N)), for the rows [b0, b0 + nb) of a sharded batch, and leaves torch's generator where
    the reference's would be."""
    from vampnet_amd.torch_rng import DeviceTorchRng, draw_units
    V = 1024
    rng = DeviceTorchRng(eng)
    torch.manual_seed(1234)
    _ = torch.rand(5)
    blob = torch.get_rng_state()
    want = []
    for s_ in pattern:
        e = torch.empty(B * N, V).exponential_(1) if s_ else None
        u = torch.zeros(B, N).uniform_(1e-20, 1)
        want.append((e, u))
    tail = torch.rand(6)
    torch.set_rng_state(blob)
    rng.load_from_torch()
    exp = torch.zeros(len(pattern), nb * N, V, device="cuda")
    unif = torch.empty(len(pattern), nb, N, device="cuda")
    draw_units(rng, [(exp[i] if s_ else None, unif[i]) for i, s_ in enumerate(pattern)], B, N, V, b0, nb)
    rng.store_to_torch()
    for i, (e, u) in enumerate(want):
        if e is not None:
            assert torch.equal(exp[i].cpu(), e[b0 * N:(b0 + nb) * N]), f"exponentials of step {i}"
        assert torch.equal(unif[i].cpu(), u[b0:b0 + nb]), f"uniforms of step {i}"
    assert torch.equal(torch.rand(6), tail)


@pytest.mark.parametrize("rows,lead_rows,total_rows,chunk", [(600, 0, 600, 1 << 16), (500, 70, 640, 1 << 16), (2300, 0, 2300, None)])
def test_torch_rng_jump_ahead_path(eng, rows, lead_rows, total_rows, chunk):
    """The parallel (jump-ahead) production of a block of torch's exponential_ stream equals the serial one and torch itself:
    chunk start states from x^J mod phi, chunks walked concurrently, generator left at the block end (checked by the uniform_
    draw that follows and by the state handed back to torch)."""
    from vampnet_amd.torch_rng import DeviceTorchRng
    V = 1024
    rng = DeviceTorchRng(eng)
    torch.manual_seed(99)
    _ = torch.rand(11)
    blob = torch.get_rng_state()
    ref_all = torch.empty(total_rows, V).exponential_(1)
    ref_u = torch.zeros(300).uniform_(1e-20, 1)
    end_state = torch.get_rng_state()
    torch.set_rng_state(blob)
    rng.load_from_torch()
    out = torch.empty(rows, V, device="cuda")
    rng.exponential_block_(out, 2 * lead_rows * V, 2 * total_rows * V, chunk_words=chunk)
    u = rng.uniform_(torch.empty(300, device="cuda"), 1e-20, 1.0)
    rng.store_to_torch()
    assert torch.equal(out.cpu(), ref_all[lead_rows:lead_rows + rows])
    assert torch.equal(u.cpu(), ref_u)
    got_state = torch.get_rng_state()
    # same stream position (the array may be a re-aligned window of the same sequence: compare by drawing)
    a = torch.rand(700)
    torch.set_rng_state(end_state)
    assert torch.equal(a, torch.rand(700))
    assert got_state.numel() == end_state.numel()


# ----------------------------------------------------------------------------- bf16x3: fp32-grade GEMM on the bf16 MFMA
def test_split3_is_exact(eng):
    """vn_split3_f32: three bf16 planes whose fp32 sum is the input bit for bit (incl. tiny / large magnitudes)."""
    x = torch.cat([_rand((4096,), 30), _rand((4096,), 31) * 1e-20, _rand((4096,), 32) * 1e20, torch.zeros(64)])
    p = eng.split3(x.cuda()).cpu().float()
    assert torch.equal(p[0] + p[1] + p[2], x)
    assert torch.equal(p[0], x.to(torch.bfloat16).float())          # plane 0 = RNE bf16 of x (== torch's cast)


X3_CONFIGS = [128, 192, 256, 96, 0]


def _x3_cfg(eng, bm=0, split=-1, abl=-1):
    eng.check(eng.lib.vn_debug_x3_config(eng.handle, bm, split, abl), "vn_debug_x3_config")


@pytest.fixture(params=X3_CONFIGS, ids=["bm128", "bm192", "bm256", "bm96", "auto"])
def x3_pipe(eng, request):
    """the tile heights of gemm_x3.hip — 128 / 192 / 256 rows and the 96-row k-split tile (bf16x3 operands; the f16x2 entries fall back
    to 128 rows) — and the by-shape default through the same bodies (tuning hook of the test's context; reset afterwards)"""
    _x3_cfg(eng, request.param)
    yield request.param
    _x3_cfg(eng)


@pytest.mark.parametrize("M,N,K", [(4600, 3840, 1280), (575, 1280, 2560), (130, 256, 64), (1384, 5120, 1280), (1, 128, 32),
                                   (300, 192, 96), (257, 128, 160)])
def test_gemm_bf16x3_fp32_grade(eng, x3_pipe, M, N, K):
    """Six bf16 MFMA products of exact operand splits == an fp32 GEMM: SAME tolerance as test_gemm_store_bias_residual
    (fp32 accumulation-order class against the float64 product of the fp32 operands).  K = 32 / 64 / 96 / 160 cover the
    one-, two-, three- and odd-tile pipelines of the ping-pong schedules; (575, 1280, 2560) takes the two-pass split-K."""
    from vampnet_amd import _lib
    a, w, b = _rand((M, K), 3), _rand((N, K), 4) / np.sqrt(K), _rand((N,), 5)
    ref64 = a.double() @ w.double().t()
    absdot = a.abs().double() @ w.abs().double().t()
    tol = (2e-6 * absdot + 1e-6).numpy()
    a3, w3 = eng.split3(a.cuda()), eng.split3(w.cuda())
    got = eng.gemm_bf16x3(a3, w3).cpu().double()
    err = np.abs((got - ref64).numpy())
    print(f"bf16x3 {M}x{N}x{K}: max err {err.max():.3e} (fp32 gemm on the host: "
          f"{(a @ w.t()).double().sub(ref64).abs().max().item():.3e})")
    assert np.all(err <= tol)
    got = eng.gemm_bf16x3(a3, w3, bias=b.cuda(), epilogue=_lib.EPI_BIAS).cpu().double()
    assert np.all(np.abs((got - (ref64 + b.double())).numpy()) <= tol)
    c0 = _rand((M, N), 6)
    out = c0.cuda().clone()
    eng.gemm_bf16x3(a3, w3, epilogue=_lib.EPI_RESIDUAL, out=out)
    assert np.all(np.abs((out.cpu().double() - (ref64 + c0.double())).numpy()) <= tol)


def test_gemm_bf16x3_tiles_agree_bitwise_and_are_race_free(eng):
    """The 128 / 192 / 256-row tiles add the same products in the same order: bitwise-equal outputs, through the direct and the
    LDS-staged epilogue alike; 20 back-to-back launches of each under a concurrently streaming kernel reproduce the same bits (LDS-DMA /
    barrier protocol of the ping-pong schedule, LDS image of the staged epilogue).  The 96-row tile splits the k-steps of a k-tile
    between its two wave groups (two partial sums per element, added in the epilogue): reproducible bit for bit too, and within the
    fp32 re-association noise of the others."""
    M, N, K = 4600, 3840, 1280
    a3, w3 = eng.split3(_rand((M, K), 13).cuda()), eng.split3((_rand((N, K), 14) / np.sqrt(K)).cuda())
    outs = {}
    junk = torch.empty(64 << 20, device="cuda")
    side = torch.cuda.Stream()
    for bm in (128, 192, 256, 96):
        _x3_cfg(eng, bm, 1)
        outs[bm] = eng.gemm_bf16x3(a3, w3).clone()
        for it in range(20):
            with torch.cuda.stream(side):
                junk.add_(1.0)                                       # uneven memory load on the other stream
            again = eng.gemm_bf16x3(a3, w3)
            assert torch.equal(again, outs[bm]), (bm, it)
    _x3_cfg(eng)
    torch.cuda.synchronize()
    assert torch.equal(outs[128], outs[256]) and torch.equal(outs[128], outs[192])
    d96 = (outs[96] - outs[128]).abs().max().item()
    print(f"96-row k-split tile vs 128 rows: max |d| = {d96:.3e}")
    assert d96 <= 2e-5
    odd = torch.empty(M * N + 1, device="cuda")[1:].view(M, N)       # base 4 bytes off a 16-byte boundary -> the direct epilogue
    eng.gemm_bf16x3(a3, w3, out=odd)
    assert torch.equal(odd, outs[128])
    _x3_cfg(eng, 96, 1)                                               # ... where the 96-row tile is not offered: falls back to 128 rows
    try:
        eng.gemm_bf16x3(a3, w3, out=odd)
    finally:
        _x3_cfg(eng)
    assert torch.equal(odd, outs[128])


@pytest.mark.parametrize("M,N,K", [(4600, 3840, 1280), (575, 1280, 2560), (130, 256, 64), (1, 128, 32), (300, 192, 96), (257, 128, 160)])
def test_gemm_bf16x3_tiled_operands_equal_planar(eng, x3_pipe, M, N, K):
    """The tiled operand layout ([row / 16][k / 32][plane][16][32]: one LDS-DMA instruction = one contiguous 1 KiB piece; what the
    model path uses for weights and activations) feeds the same products in the same order as the planar planes: bitwise equal
    outputs at every tile height, store / bias / residual epilogues."""
    from vampnet_amd import _lib
    a3, w3 = eng.split3(_rand((M, K), 3).cuda()), eng.split3((_rand((N, K), 4) / np.sqrt(K)).cuda())
    at, wt = eng.tile3(a3), eng.tile3(w3)
    b = _rand((N,), 5).cuda()
    assert torch.equal(eng.gemm_bf16x3(at, wt, tiled_shape=(M, N, K)), eng.gemm_bf16x3(a3, w3))
    assert torch.equal(eng.gemm_bf16x3(at, wt, bias=b, epilogue=_lib.EPI_BIAS, tiled_shape=(M, N, K)),
                       eng.gemm_bf16x3(a3, w3, bias=b, epilogue=_lib.EPI_BIAS))
    c0 = _rand((M, N), 6).cuda()
    o1, o2 = c0.clone(), c0.clone()
    eng.gemm_bf16x3(at, wt, epilogue=_lib.EPI_RESIDUAL, out=o1, tiled_shape=(M, N, K))
    eng.gemm_bf16x3(a3, w3, epilogue=_lib.EPI_RESIDUAL, out=o2)
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("M,N,tokens", [(1280, 3840, 4600), (5120, 1280, 575), (1280, 2560, 2300), (256, 128, 130), (32, 64, 1), (96, 192, 47),
                                        (160, 320, 291), (4096, 1280, 1150)])
def test_gemm_bf16x3_token_major_operands_equal_the_transposed_ones(eng, M, N, tokens):
    """TN operand mode (round 6; the dW GEMMs of training): C = At^T Wt read straight from the token-major tiled planes — the LDS transposing
    read builds the fragments — against the NT kernel on the explicitly transposed matrices: the same products in the same order, so
    BITWISE equal at every tile height, with and without the k-split; ragged token counts (the zero page stands in for the missing
    16-token block, the zero rows of the last block contribute exact zeros); asymmetric inputs (a transposed C or swapped operands
    cannot pass)."""
    at = (_rand((tokens, M), 71) * torch.linspace(0.5, 2.0, M)).cuda()
    wt = (_rand((tokens, N), 72) / np.sqrt(tokens)).cuda()
    at_t, wt_t = eng.tile3(eng.split3(at)), eng.tile3(eng.split3(wt))            # tile3 zero-pads the rows to 16
    Kp = (tokens + 31) // 32 * 32
    pad = lambda x: torch.cat([x, x.new_zeros(Kp - tokens, x.shape[1])], 0).t().contiguous()
    a3, w3 = eng.split3(pad(at)), eng.split3(pad(wt))                            # [3][M][Kp], [3][N][Kp]
    for bm, split in [(128, 1), (192, 1), (256, 1), (128, 2), (0, -1)]:
        _x3_cfg(eng, bm, split)
        try:
            got = eng.gemm_bf16x3_tn(at_t, wt_t, M, N, tokens)
            if bm:
                want = eng.gemm_bf16x3(a3, w3)
                assert torch.equal(got, want), (bm, split, (got - want).abs().max().item())
            else:                                                                # by-shape plan: the NT side may pick the 96-row k-split tile
                ref = at.double().t() @ wt.double()
                assert (got.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-30     # fp32 accumulation over up to 4600 tokens
        finally:
            _x3_cfg(eng)


def test_gemm_bf16x3_identity_and_geglu(eng, x3_pipe):
    from vampnet_amd import _lib
    K = N = 256
    a = torch.eye(K)[:200]
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 1e-3
    got = eng.gemm_bf16x3(eng.split3(a.cuda()), eng.split3(w.cuda())).cpu()
    assert torch.equal(got, w.t()[:200])                                # 1 * w is exact: catches a transposed C write
    M, D = 575, 1280
    x, w1 = _rand((M, D), 7), _rand((4 * D, D), 8, 1.0 / np.sqrt(D))
    ref = O.gated_gelu(torch.nn.functional.linear(x, w1))
    val, gate = w1[:2 * D].reshape(2 * D // 32, 32, D), w1[2 * D:].reshape(2 * D // 32, 32, D)
    w1p = torch.stack([val, gate], dim=1).reshape(4 * D, D)
    got = eng.gemm_bf16x3(eng.split3(x.cuda()), eng.split3(w1p.cuda()), epilogue=_lib.EPI_GEGLU).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)


def test_two_contexts_tune_independently(eng):
    """Tuning state lives in vn_ctx (include/vampnet_hip_debug.h): two contexts in one process run DIFFERENT bf16x3 tile heights
    concurrently on their own streams and give bitwise equal results (the heights add the same products in the same order), and a
    setting made on one context does not leak into the other."""
    from vampnet_amd.engine import Engine
    eng2 = Engine("cuda:0")
    M, N, K = 1150, 3840, 1280
    a, w = _rand((M, K), 50).cuda(), (_rand((N, K), 51) / np.sqrt(K)).cuda()
    a3, w3 = eng.split3(a), eng.split3(w)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    eng.check(eng.lib.vn_debug_x3_config(eng.handle, 128, 1, -1), "vn_debug_x3_config")       # (k-split off: the cost model may split the
    eng2.check(eng2.lib.vn_debug_x3_config(eng2.handle, 256, 1, -1), "vn_debug_x3_config")     # two heights differently at this shape)
    try:
        torch.cuda.synchronize()
        outs1, outs2 = [], []
        for _ in range(4):                           # interleaved launches: both contexts in flight at once
            with torch.cuda.stream(s1):
                outs1.append(eng.gemm_bf16x3(a3, w3))
            with torch.cuda.stream(s2):
                outs2.append(eng2.gemm_bf16x3(a3, w3))
        torch.cuda.synchronize()
        for o1, o2 in zip(outs1, outs2):
            assert torch.equal(o1, outs1[0]) and torch.equal(o2, outs1[0])
        # context 2 still runs 256-row tiles after context 1 is reset: its ablation-free result is unchanged, and a forced k-split
        # on context 1 (different summation order) moves ONLY context 1's result
        eng.check(eng.lib.vn_debug_x3_config(eng.handle, 128, 2, -1), "vn_debug_x3_config")
        split1 = eng.gemm_bf16x3(a3, w3)
        same2 = eng2.gemm_bf16x3(a3, w3)
        torch.cuda.synchronize()
        assert torch.equal(same2, outs1[0])
        assert not torch.equal(split1, outs1[0]) and (split1 - outs1[0]).abs().max().item() < 1e-4
    finally:
        eng.check(eng.lib.vn_debug_x3_config(eng.handle, 0, -1, -1), "vn_debug_x3_config")
        eng2.check(eng2.lib.vn_debug_x3_config(eng2.handle, 0, -1, -1), "vn_debug_x3_config")
