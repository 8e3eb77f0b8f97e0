"""-m gpu: the f16x2 operand format of gemm_x3.hip (two fp16 planes, three matrix-core products, second accumulator) through the
C ABI — the bf16x3 GEMM's own tests at the SAME tolerances, plus the properties of the split itself."""
import numpy as np
import pytest
import torch

from oracle import vampnet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from vampnet_amd.engine import Engine
    return Engine("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32))


def _x3_cfg(eng, bm=0, split=-1, abl=-1):
    eng.check(eng.lib.vn_debug_x3_config(eng.handle, bm, split, abl), "vn_debug_x3_config")


@pytest.fixture(params=[128, 192, 256, 0], ids=["bm128", "bm192", "bm256", "auto"])
def x3_pipe(eng, request):
    _x3_cfg(eng, request.param)
    yield request.param
    _x3_cfg(eng)


def test_split2h_properties(eng):
    """h0 + h1 / 2048 reproduces x to 2^-22 |x| over fp16's normal range, to 2^-24 absolute below it (subnormals are produced, not
    flushed), saturates at +-65504, and the tiled image is the planar one re-laid ([row / 16][k / 32][plane][16][32])."""
    g = torch.Generator().manual_seed(1)
    mag = torch.exp(torch.empty(64, 256).uniform_(np.log(1e-9), np.log(6e4), generator=g))
    x = mag * torch.where(torch.rand(64, 256, generator=g) < 0.5, -1.0, 1.0)
    x[0, :8] = torch.tensor([0.0, -0.0, 65504.0, -65504.0, 1e5, -3e38, 6.1035156250e-05, 5.9604644775390625e-08])
    h = eng.split2h(x.cuda()).cpu()
    assert h.dtype == torch.float16 and h.shape == (2, 64, 256)
    back = h[0].double() + h[1].double() / 2048.0
    xs = x.double().clamp(-65504.0, 65504.0)
    err = (back - xs).abs()
    bound = torch.maximum(xs.abs() * 2.0 ** -22, torch.full_like(xs, 2.0 ** -24))
    assert torch.all(err <= bound), float((err / bound).max())
    assert torch.isfinite(h.float()).all()
    assert back[0, 4] == 65504.0 and back[0, 5] == -65504.0                      # saturation, not inf
    assert h[0][0, 7] != 0                                                      # 2^-24: the smallest fp16 subnormal survives
    t = eng.split2h(x.cuda(), tiled=True).cpu()
    assert torch.equal(t, h.reshape(2, 4, 16, 8, 32).permute(1, 3, 0, 2, 4).contiguous())


@pytest.mark.parametrize("M,N,K", [(4600, 3840, 1280), (575, 1280, 2560), (130, 256, 64), (1, 128, 32), (300, 192, 96), (257, 128, 160),
                                   (1150, 5120, 1280)])
def test_gemm_f16x2_fp32_grade(eng, x3_pipe, M, N, K):
    """Three fp16 products of the two-plane split == an fp32 GEMM: the tolerance of test_gemm_bf16x3_fp32_grade (fp32
    accumulation-order class against the float64 product of the fp32 operands), every tile height, store / bias / residual."""
    from vampnet_amd import _lib
    a, w, b = _rand((M, K), 3), _rand((N, K), 4) / np.sqrt(K), _rand((N,), 5)
    ref64 = a.double() @ w.double().t()
    absdot = a.abs().double() @ w.abs().double().t()
    tol = (2e-6 * absdot + 1e-6).numpy()
    a2, w2 = eng.split2h(a.cuda()), eng.split2h(w.cuda())
    got = eng.gemm_f16x2(a2, w2).cpu().double()
    err = np.abs((got - ref64).numpy())
    rms = float(np.sqrt((err ** 2).mean()) / np.sqrt((ref64.numpy() ** 2).mean()))
    rms32 = float(((a @ w.t()).double() - ref64).pow(2).mean().sqrt() / ref64.pow(2).mean().sqrt())
    print(f"f16x2 {M}x{N}x{K}: max err {err.max():.3e}, rms rel {rms:.3e} (fp32 gemm on the host: max "
          f"{(a @ w.t()).double().sub(ref64).abs().max().item():.3e}, rms rel {rms32:.3e})")
    assert np.all(err <= tol)
    got = eng.gemm_f16x2(a2, w2, bias=b.cuda(), epilogue=_lib.EPI_BIAS).cpu().double()
    assert np.all(np.abs((got - (ref64 + b.double())).numpy()) <= tol)
    c0 = _rand((M, N), 6)
    out = c0.cuda().clone()
    eng.gemm_f16x2(a2, w2, epilogue=_lib.EPI_RESIDUAL, out=out)
    assert np.all(np.abs((out.cpu().double() - (ref64 + c0.double())).numpy()) <= tol)


@pytest.mark.parametrize("scale_a,scale_w", [(0.02, 1.0), (300.0, 1.0), (1.0, 0.03), (1e-3, 30.0)])
def test_gemm_f16x2_operand_magnitudes(eng, scale_a, scale_w):
    """the error stays in the fp32 class when the operands sit elsewhere in fp16's range (attention outputs of a uniform softmax
    ~ 0.02, outlier activations ~ 1e3, small weights): no per-tensor scaling is needed because BOTH planes keep 11 bits"""
    M, N, K = 575, 1280, 1280
    a, w = _rand((M, K), 21) * scale_a, _rand((N, K), 22) / np.sqrt(K) * scale_w
    ref64 = a.double() @ w.double().t()
    got = eng.gemm_f16x2(eng.split2h(a.cuda()), eng.split2h(w.cuda())).cpu().double()
    rms = float((got - ref64).pow(2).mean().sqrt() / ref64.pow(2).mean().sqrt())
    rms32 = float(((a @ w.t()).double() - ref64).pow(2).mean().sqrt() / ref64.pow(2).mean().sqrt())
    print(f"scales {scale_a} x {scale_w}: f16x2 rms rel {rms:.3e}, host fp32 gemm {rms32:.3e}")
    assert rms < 6e-7 and rms < 2.0 * rms32 + 1e-7


def test_gemm_f16x2_tiles_agree_bitwise_and_are_race_free(eng):
    M, N, K = 4600, 3840, 1280
    a2, w2 = eng.split2h(_rand((M, K), 13).cuda()), eng.split2h((_rand((N, K), 14) / np.sqrt(K)).cuda())
    outs = {}
    junk = torch.empty(64 << 20, device="cuda")
    side = torch.cuda.Stream()
    for bm in (128, 192, 256):
        _x3_cfg(eng, bm, 1)
        outs[bm] = eng.gemm_f16x2(a2, w2).clone()
        for it in range(20):
            with torch.cuda.stream(side):
                junk.add_(1.0)
            again = eng.gemm_f16x2(a2, w2)
            assert torch.equal(again, outs[bm]), (bm, it)
    _x3_cfg(eng)
    torch.cuda.synchronize()
    assert torch.equal(outs[128], outs[256]) and torch.equal(outs[128], outs[192])
    odd = torch.empty(M * N + 1, device="cuda")[1:].view(M, N)       # base 4 bytes off a 16-byte boundary -> the direct epilogue
    eng.gemm_f16x2(a2, w2, out=odd)
    assert torch.equal(odd, outs[128])


@pytest.mark.parametrize("M,N,K", [(4600, 3840, 1280), (575, 1280, 2560), (130, 256, 64), (1, 128, 32), (300, 192, 96), (257, 128, 160)])
def test_gemm_f16x2_tiled_operands_equal_planar(eng, x3_pipe, M, N, K):
    from vampnet_amd import _lib
    a, w = _rand((M, K), 3).cuda(), (_rand((N, K), 4) / np.sqrt(K)).cuda()
    a2, w2, at, wt = eng.split2h(a), eng.split2h(w), eng.split2h(a, tiled=True), eng.split2h(w, tiled=True)
    b = _rand((N,), 5).cuda()
    assert torch.equal(eng.gemm_f16x2(at, wt, tiled_shape=(M, N, K)), eng.gemm_f16x2(a2, w2))
    assert torch.equal(eng.gemm_f16x2(at, wt, bias=b, epilogue=_lib.EPI_BIAS, tiled_shape=(M, N, K)),
                       eng.gemm_f16x2(a2, w2, bias=b, epilogue=_lib.EPI_BIAS))
    c0 = _rand((M, N), 6).cuda()
    o1, o2 = c0.clone(), c0.clone()
    eng.gemm_f16x2(at, wt, epilogue=_lib.EPI_RESIDUAL, out=o1, tiled_shape=(M, N, K))
    eng.gemm_f16x2(a2, w2, epilogue=_lib.EPI_RESIDUAL, out=o2)
    assert torch.equal(o1, o2)


def test_gemm_f16x2_identity_and_geglu(eng, x3_pipe):
    from vampnet_amd import _lib
    K = N = 256
    a = torch.eye(K)[:200]
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) * 1e-3
    got = eng.gemm_f16x2(eng.split2h(a.cuda()), eng.split2h(w.cuda())).cpu()
    np.testing.assert_allclose(got.numpy(), w.t()[:200].numpy(), rtol=2.5e-7, atol=0)      # 1 * w: the split's 2^-22, transposition-sensitive
    M, D = 575, 1280
    x, w1 = _rand((M, D), 7), _rand((4 * D, D), 8, 1.0 / np.sqrt(D))
    ref = O.gated_gelu(torch.nn.functional.linear(x, w1))
    val, gate = w1[:2 * D].reshape(2 * D // 32, 32, D), w1[2 * D:].reshape(2 * D // 32, 32, D)
    w1p = torch.stack([val, gate], dim=1).reshape(4 * D, D)
    got = eng.gemm_f16x2(eng.split2h(x.cuda()), eng.split2h(w1p.cuda()), epilogue=_lib.EPI_GEGLU).cpu()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-4, atol=2e-5)
