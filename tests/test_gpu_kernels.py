"""GPU parity tests, kernel by kernel, through the C ABI (libhamer_hip.so via ctypes).

Each HIP kernel is compared with the oracle's statement of the same step on the same seeded
inputs.  16-bit outputs: the oracle rounds where the kernel rounds, so the residual
difference is fp32 accumulation order (a value may flip one 16-bit ulp).  Integer / byte
work (crop) must be bit-exact.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from hamer_yolo_amd import lib as L
from hamer_yolo_amd import ops, synth
from oracle import crop_ref
from oracle import hamer_ref as R

DEV = "cuda"


def _u(name, shape, hw=1.0, seed=0, center=0.0):
    return synth.uniform(name, shape, hw, center, seed=seed)


def _ulp16(dtype):
    return 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11


# ------------------------------------------------------------------------------------ GEMM
def test_gemm_exact_integers_asymmetric():
    """Small-integer operands: every product and partial sum is exact in fp32, so the result must
    be bit-exact whatever the accumulation order; W asymmetric to catch transposed writes."""
    M, N, K = 192, 256, 128
    x = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
    w = ((torch.arange(N * K).reshape(N, K) * 5 + torch.arange(N)[:, None]) % 5 - 2).float()
    ref = x @ w.t()
    for dt in (torch.bfloat16, torch.float16):
        out = ops.gemm(x.to(DEV, dt), w.to(DEV, dt), epilogue=L.HM_EPI_F32)
        assert torch.equal(out.cpu(), ref)


def _tile_variant_exact(variant):
    lib = L.load()
    try:
        L.check(lib.hm_gemm_set_variant(variant))
        for (M, N, K) in ((300, 260, 64), (513, 388, 128), (1000, 1284, 448), (700, 516, 192), (257, 260, 1280)):
            x = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
            w = ((torch.arange(N * K).reshape(N, K) * 5 + torch.arange(N)[:, None]) % 5 - 2).float()
            bias = (torch.arange(N) % 9 - 4).float()
            ref = x @ w.t() + bias
            out = ops.gemm(x.to(DEV, torch.bfloat16), w.to(DEV, torch.bfloat16), bias.to(DEV), L.HM_EPI_F32)
            assert torch.equal(out.cpu(), ref), (variant, M, N, K)
    finally:
        lib.hm_gemm_set_variant(-1)


@pytest.mark.parametrize("variant", [0, 10, 24, 26])
def test_gemm_tile_variants_exact(variant):
    """Every tile / pipeline configuration the product library ships (what pick_variant can choose) on exact-integer data
    (bit-exact whatever the summation order), ragged M and N, several K-tile counts (ring prologue / steady state / tail)."""
    _tile_variant_exact(variant)


@pytest.mark.parametrize("hands", [16, 17, 23, 40, 45, 68, 72])
def test_gemm_tile_rule_exact_at_every_batch_size(hands):
    """Round 3: pick_variant's rate model (HM_OPT_GEMM_TILE_RULE = 0; 1 = round 2's 85 % rule) and the persistent kernel's grid
    (rounded UP to the 8 XCDs when there are fewer tiles than workgroups: 180 tiles -> 184 workgroups, four of them idle; all
    256 CUs when that saves a round) at the ViT-H row counts of 16..72 hands: whatever tile is chosen, exact-integer data must
    come back bit-exact, for the three epilogues the choice depends on, and both rules agree."""
    M = hands * 192
    lib = L.load()
    for (N, K, epi) in ((3840, 128, L.HM_EPI_STORE), (1280, 192, L.HM_EPI_RESID_F32), (5120, 128, L.HM_EPI_STORE)):
        x = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
        w = ((torch.arange(N * K).reshape(N, K) * 5 + torch.arange(N)[:, None]) % 5 - 2).float()
        bias = (torch.arange(N) % 9 - 4).float()
        ref = x @ w.t() + bias
        r = ((torch.arange(M * N).reshape(M, N) % 11) - 5).float() if epi == L.HM_EPI_RESID_F32 else None
        outs = []
        for rule in (0, 1):
            with L.option(L.HM_OPT_GEMM_TILE_RULE, rule):
                out = ops.gemm(x.to(DEV, torch.float16), w.to(DEV, torch.float16), bias.to(DEV), epi,
                               resid=r.to(DEV) if r is not None else None)
            outs.append(out.float().cpu())
        assert torch.equal(outs[0], ref + (r if r is not None else 0)), (hands, N, K, epi)
        assert torch.equal(outs[0], outs[1])


def test_gemm_set_variant_accepts_shipped_tiles_only():
    """The experimental tiles (1-9, 11, 12, 21-23, 25, 28, 29) and the wrong-result ablations live in libhamer_hip_abl.so
    (python -m hamer_yolo_amd.build --ablations); the product refuses them instead of silently running something else."""
    lib = L.load()
    for v in (1, 8, 9, 12, 14, 17, 21, 23, 25, 27, 28, 29, 30, 31, 33, 99):
        assert lib.hm_gemm_set_variant(v) != 0, v
    for v in (0, 10, 24, 26, -1):
        assert lib.hm_gemm_set_variant(v) == 0, v


@pytest.fixture
def experiments_lib(monkeypatch):
    """libhamer_hip_abl.so (experimental tiles; opt-in: HM_TEST_EXPERIMENTS=1 and the library built) in place of the product
    library for one test."""
    import os
    path = L.LIB_PATH.replace(".so", "_abl.so")
    if os.environ.get("HM_TEST_EXPERIMENTS") != "1" or not os.path.exists(path):
        pytest.skip("experimental GEMM tiles: build with `python -m hamer_yolo_amd.build --ablations` and set HM_TEST_EXPERIMENTS=1")
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", path)
    yield L.load()
    L._lib = None                                     # (monkeypatch restores LIB_PATH; the next load() opens the product library again)


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 21, 22, 23])
def test_gemm_experimental_tiles_exact(variant, experiments_lib):
    _tile_variant_exact(variant)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gemm_deep_prefetch_variant_exact(dt):
    """gemm_x3_kernel (variant 24: X three-slot ring, two K-steps ahead, hand-counted vmcnt) on exact-integer data through
    the residual epilogue: bit-exact for 1, 2, 3 and many K-steps, ragged M / N, and repeated launches (race screen)."""
    lib = L.load()
    try:
        L.check(lib.hm_gemm_set_variant(24))
        for (M, N, K) in ((300, 260, 64), (513, 388, 128), (1000, 1284, 192), (700, 516, 448), (257, 260, 1280), (1536, 512, 5120)):
            x = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
            w = ((torch.arange(N * K).reshape(N, K) * 5 + torch.arange(N)[:, None]) % 5 - 2).float()
            bias = (torch.arange(N) % 9 - 4).float()
            resid = ((torch.arange(M * N).reshape(M, N) * 3) % 11 - 5).float()
            ref = x @ w.t() + bias + resid
            xd, wd, bd, rd = x.to(DEV, dt), w.to(DEV, dt), bias.to(DEV), resid.to(DEV)
            for _ in range(4):
                out = ops.gemm(xd, wd, bd, L.HM_EPI_RESID_F32, resid=rd)
                assert torch.equal(out.cpu(), ref), (M, N, K)
            o16 = ops.gemm(xd, wd, bd, L.HM_EPI_STORE)
            assert torch.equal(o16.float().cpu(), (x @ w.t() + bias).to(dt).float()), (M, N, K)
    finally:
        lib.hm_gemm_set_variant(-1)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_persistent_kernel_exact(dt):
    _persistent_kernel_exact(dt, 26)


@pytest.mark.parametrize("pv", [27, 33, 34, 36, 37, 38, 39, 40])
def test_gemm_pipelined_persistent_experiment_exact(pv, experiments_lib):
    _persistent_kernel_exact(torch.float16, pv)


def _persistent_kernel_exact(dt, pv):
    """Variants 27 / 33 = gemm_pp_kernel, round 4, experiments library only: the same persistent walk with a software-pipelined K loop (fragment reads issued from
    asm five groups ahead and waited for by count, the step's barrier at group 12, W two steps ahead) -- same K order, so the same
    bytes as variant 26 / 24 on random data too.
    gemm_px_kernel (variant 26: one workgroup per CU walks its tiles, the LDS-DMA pipeline runs across tile boundaries,
    hand-counted vmcnt over copies AND the epilogue's stores) on exact-integer data: bit-exact against torch for 2 and many
    K-steps, with and without bias, tile counts below / equal to / far above the CU count (1..6 tiles per workgroup, uneven
    shares), repeated launches as a race screen; the GELU epilogue bit-equal to the one-tile kernel's (variant 24)."""
    lib = L.load()
    try:
        for (M, N, K) in ((2048, 256, 128), (2304, 2560, 128), (4096, 4096, 192), (2560, 10240, 128), (12288, 3840, 1280), (5120, 5120, 64 * 7)):
            x = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
            w = ((torch.arange(N * K).reshape(N, K) * 5 + torch.arange(N)[:, None]) % 5 - 2).float()
            bias = (torch.arange(N) % 9 - 4).float()
            xd, wd, bd = x.to(DEV, dt), w.to(DEV, dt), bias.to(DEV)
            ref = (xd.float() @ wd.float().t()).cpu()                     # exact in fp32: small integers
            L.check(lib.hm_gemm_set_variant(pv))
            for rep in range(3):
                o = ops.gemm(xd, wd, bd, L.HM_EPI_STORE)
                assert torch.equal(o.float().cpu(), (ref + bias).to(dt).float()), (M, N, K, rep)
            o = ops.gemm(xd, wd, None, L.HM_EPI_STORE)
            assert torch.equal(o.float().cpu(), ref.to(dt).float()), (M, N, K, "no bias")
            # random operands, GELU: the same rounding order as the one-tile kernel, so bit-equal to it
            xr = _u("px", (M, K), 1.0, seed=M).to(DEV, dt)
            wr = _u("pw", (N, K), 0.05, seed=N).to(DEV, dt)
            g26 = [ops.gemm(xr, wr, bd, L.HM_EPI_GELU) for _ in range(2)]
            s26 = ops.gemm(xr, wr, bd, L.HM_EPI_STORE)
            for form in (1, 2):                                       # both epilogue forms (tile through LDS / lane swaps): the same bytes
                with L.option(L.HM_OPT_PX_LDS_EPILOGUE, form):
                    assert torch.equal(ops.gemm(xr, wr, bd, L.HM_EPI_GELU), g26[0]) and torch.equal(ops.gemm(xr, wr, bd, L.HM_EPI_STORE), s26), (M, N, K, form)
            L.check(lib.hm_gemm_set_variant(24))
            g24 = ops.gemm(xr, wr, bd, L.HM_EPI_GELU)
            s24 = ops.gemm(xr, wr, bd, L.HM_EPI_STORE)
            assert torch.equal(g26[0], g24) and torch.equal(g26[1], g24) and torch.equal(s26, s24), (M, N, K)
        # shapes the persistent kernel does not take (ragged M / N, one K-step) fall back to the one-tile kernels
        L.check(lib.hm_gemm_set_variant(pv))
        for (M, N, K) in ((300, 260, 64), (2048, 2048, 64), (1000, 1284, 192)):
            x = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
            w = ((torch.arange(N * K).reshape(N, K) * 5 + torch.arange(N)[:, None]) % 5 - 2).float()
            o = ops.gemm(x.to(DEV, dt), w.to(DEV, dt), None, L.HM_EPI_STORE)
            assert torch.equal(o.float().cpu(), (x @ w.t()).to(dt).float()), (M, N, K)
    finally:
        lib.hm_gemm_set_variant(-1)


@pytest.mark.parametrize("variant", [25, 28, 29])
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_256x320_tile_exact(dt, variant, experiments_lib):
    """Variant 25 (256x320 tile) and variant 28 (256x160 tile, both operands two K-steps ahead in three-slot rings, hand-counted
    vmcnt with a wave-dependent copy count) on exact-integer data, ragged and whole shapes, 1 / 2 / 3 / many K-steps, every
    epilogue family they can be given: 16-bit store, fp32 out, fp32 residual; repeated launches as a race screen."""
    lib = L.load()
    try:
        L.check(lib.hm_gemm_set_variant(variant))
        for (M, N, K) in ((300, 260, 64), (513, 388, 128), (1000, 1284, 448), (768, 5120, 1280), (2304, 640, 192), (4096, 3840, 192)):
            x = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
            w = ((torch.arange(N * K).reshape(N, K) * 5 + torch.arange(N)[:, None]) % 5 - 2).float()
            bias = (torch.arange(N) % 9 - 4).float()
            resid = ((torch.arange(M * N).reshape(M, N) * 3) % 11 - 5).float()
            xd, wd, bd, rd = x.to(DEV, dt), w.to(DEV, dt), bias.to(DEV), resid.to(DEV)
            ref = (xd.float() @ wd.float().t()).cpu() + bias
            assert torch.equal(ops.gemm(xd, wd, bd, L.HM_EPI_F32).cpu(), ref), (M, N, K)
            for _ in range(3):
                assert torch.equal(ops.gemm(xd, wd, bd, L.HM_EPI_RESID_F32, resid=rd).cpu(), ref + resid), (M, N, K)
            if N % 8 == 0:
                assert torch.equal(ops.gemm(xd, wd, bd, L.HM_EPI_STORE).float().cpu(), ref.to(dt).float()), (M, N, K)
    finally:
        lib.hm_gemm_set_variant(-1)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(384, 1280, 768), (192, 3840, 1280), (200, 132, 64), (64, 6144, 1280), (1, 4, 64)])
def test_gemm_epilogues(M, N, K, dt):
    x = _u("gx", (M, K), 1.0, seed=M).to(dt)
    w = _u("gw", (N, K), 0.05, seed=N).to(dt)
    bias = _u("gb", (N,), 0.5, seed=K)
    resid = _u("gr", (M, N), 1.0, seed=3)
    acc = x.float().double() @ w.float().double().t()
    xd, wd, bd, rd = x.to(DEV), w.to(DEV), bias.to(DEV), resid.to(DEV)
    ulp = _ulp16(dt)
    # f32 outputs
    o = ops.gemm(xd, wd, bd, L.HM_EPI_F32).cpu().double()
    np.testing.assert_allclose(o.numpy(), (acc + bias.double()).numpy(), atol=2e-5 * math.sqrt(K), rtol=1e-5)
    o = ops.gemm(xd, wd, None, L.HM_EPI_F32).cpu().double()
    np.testing.assert_allclose(o.numpy(), acc.numpy(), atol=2e-5 * math.sqrt(K), rtol=1e-5)
    o = ops.gemm(xd, wd, bd, L.HM_EPI_RESID_F32, resid=rd).cpu().double()
    np.testing.assert_allclose(o.numpy(), (acc + bias.double() + resid.double()).numpy(), atol=2e-5 * math.sqrt(K), rtol=1e-5)
    # residual with row modulo (positional embedding) and in-place accumulate
    if M % 4 == 0 and M >= 8:
        mod = M // 4
        o = ops.gemm(xd, wd, bd, L.HM_EPI_RESID_F32, resid=rd[:mod].contiguous(), resid_mod=mod).cpu().double()
        ref = acc + bias.double() + resid[:mod].double().repeat(4, 1)
        np.testing.assert_allclose(o.numpy(), ref.numpy(), atol=2e-5 * math.sqrt(K), rtol=1e-5)
    buf = rd.clone()
    ops.gemm(xd, wd, bd, L.HM_EPI_RESID_F32, resid=buf, out=buf)
    np.testing.assert_allclose(buf.cpu().double().numpy(), (acc + bias.double() + resid.double()).numpy(),
                               atol=2e-5 * math.sqrt(K), rtol=1e-5)
    # 16-bit outputs
    for epi, fn in ((L.HM_EPI_STORE, lambda t: t), (L.HM_EPI_GELU, lambda t: F.gelu(t)), (L.HM_EPI_SILU, lambda t: F.silu(t))):
        ref = fn((acc + bias.double()).float())
        o = ops.gemm(xd, wd, bd, epi).cpu().float()
        np.testing.assert_allclose(o.numpy(), ref.numpy(), atol=1e-4 * math.sqrt(K) + 2e-3, rtol=2 * ulp)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_gemm_inloop_residual_exact(dt):
    """gemm_x3r_kernel (round 3: the fp32 residual rows of proj / fc2 are requested inside the K loop, two accumulator-layout
    pieces per K-step by loads hipcc does not track, added into the accumulators two steps later behind a counted wait that
    names the registers) on exact-integer data: the minimum K (20 steps: 18 unrolled + 2), one more, and fc2's 80; whole
    tiles only (others fall back to the epilogue form); the residual aliasing the output as in the forward (x += ...);
    repeated launches as a race screen; and against the round-2 epilogue form (HM_OPT_RESID_IN_EPILOGUE) on random data,
    where the two differ by the position of one fp32 add."""
    lib = L.load()
    for (M, N, K) in ((512, 1280, 1280), (768, 256, 1344), (256, 512, 5120), (2304, 1280, 1280)):
        x = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
        w = ((torch.arange(N * K).reshape(N, K) * 5 + torch.arange(N)[:, None]) % 5 - 2).float()
        bias = (torch.arange(N) % 9 - 4).float()
        resid = ((torch.arange(M * N).reshape(M, N) * 3 + torch.arange(N)[None, :] * 7) % 1021 - 510).float()
        xd, wd, bd, rd = x.to(DEV, dt), w.to(DEV, dt), bias.to(DEV), resid.to(DEV)
        ref = x @ w.t() + bias + resid
        for _ in range(3):
            assert torch.equal(ops.gemm(xd, wd, bd, L.HM_EPI_RESID_F32, resid=rd).cpu(), ref), (M, N, K)
        inplace = rd.clone()
        ops.gemm(xd, wd, bd, L.HM_EPI_RESID_F32, resid=inplace, out=inplace)
        assert torch.equal(inplace.cpu(), ref), (M, N, K)
        assert torch.equal(ops.gemm(xd, wd, None, L.HM_EPI_RESID_F32, resid=rd).cpu(), ref - bias), (M, N, K)     # no bias
        with L.option(L.HM_OPT_RESID_IN_EPILOGUE, 1):
            assert torch.equal(ops.gemm(xd, wd, bd, L.HM_EPI_RESID_F32, resid=rd).cpu(), ref), (M, N, K)
    M, N, K = 1024, 1280, 5120
    a, wt = _u("ra", (M, K), 1.0, seed=1).to(DEV, dt), _u("rw", (N, K), 0.05, seed=2).to(DEV, dt)
    bb, rr = _u("rb", (N,), 0.5, seed=3).to(DEV), _u("rr", (M, N), 2.0, seed=4).to(DEV)
    new = ops.gemm(a, wt, bb, L.HM_EPI_RESID_F32, resid=rr)
    with L.option(L.HM_OPT_RESID_IN_EPILOGUE, 1):
        old = ops.gemm(a, wt, bb, L.HM_EPI_RESID_F32, resid=rr)
    ref64 = (a.double() @ wt.double().t() + bb.double() + rr.double()).cpu()
    assert float((new - old).abs().max()) < 2e-5                         # one fp32 add moved: a few ulps of |x| <= ~10
    assert float((new.cpu().double() - ref64).abs().max()) <= float((old.cpu().double() - ref64).abs().max()) * 1.5 + 1e-6


@pytest.mark.parametrize("variant", [-1, 0, 10])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_gemm_deferred_layernorm(variant, dt):
    """HM_EPI_RESID_LN -> HM_EPI_LN_STORE / HM_EPI_LN_GELU == residual add, nn.LayerNorm, nn.Linear (vit.py:148-151):
    the producer's x / x*gamma / 64-column statistics against fp64, the consumer against LN in fp64 on the
    producer's x (ragged M; D = 320 and 1280; a row mean well away from zero)."""
    lib = L.load()
    try:
        L.check(lib.hm_gemm_set_variant(variant))
        for (M, D, K0, N) in ((1100, 1280, 256, 3840), (520, 320, 128, 1280), (70, 320, 64, 8)):
            a = _u("da", (M, K0), 1.0, seed=M).to(dt)
            w0 = _u("dw0", (D, K0), 0.08, seed=D).to(dt)
            b0 = _u("db0", (D,), 0.3, seed=1)
            resid = _u("dr", (M, D), 1.0, seed=2, center=0.4)
            gamma, beta = _u("dg", (D,), 0.3, seed=3, center=1.0), _u("dbt", (D,), 0.2, seed=4)
            w1 = _u("dw1", (N, D), 0.05, seed=N).to(dt)
            b1 = _u("db1", (N,), 0.3, seed=5)
            eps = 1e-6
            x_ref = a.double() @ w0.double().t() + b0.double() + resid.double()
            xg = torch.empty(M, D, device=DEV, dtype=dt)
            stats = torch.full((D // 64, M, 2), float("nan"), device=DEV)
            x = ops.gemm(a.to(DEV), w0.to(DEV), b0.to(DEV), L.HM_EPI_RESID_LN, resid=resid.to(DEV), ln_gamma=gamma.to(DEV),
                         ln_xg=xg, ln_stats=stats)
            np.testing.assert_allclose(x.cpu().double().numpy(), x_ref.numpy(), atol=2e-5 * math.sqrt(K0), rtol=1e-5)
            xc = x.cpu().double()
            np.testing.assert_allclose(xg.cpu().double().numpy(), (xc * gamma.double()).numpy(), atol=1e-7, rtol=_ulp16(dt))   # atol: fp16 subnormal step
            parts = xc.reshape(M, D // 64, 64).transpose(0, 1)
            np.testing.assert_allclose(stats[..., 0].cpu().double().numpy(), parts.sum(-1).numpy(), atol=1e-4, rtol=1e-5)
            np.testing.assert_allclose(stats[..., 1].cpu().double().numpy(), (parts * parts).sum(-1).numpy(), atol=1e-4, rtol=1e-5)
            fin = ops.ln_finalize(stats, eps)
            np.testing.assert_allclose(fin[:, 0].cpu().double().numpy(), xc.mean(-1).numpy(), atol=1e-6, rtol=1e-5)
            np.testing.assert_allclose(fin[:, 1].cpu().double().numpy(), (xc.var(-1, unbiased=False) + eps).rsqrt().numpy(), rtol=2e-5)
            # consumer: operands exactly as the product builds them (colsum / bias over the 16-bit weights)
            colsum = (w1.double() @ gamma.double()).float().to(DEV)
            bias_ln = (b1.double() + w1.double() @ beta.double()).float().to(DEV)
            ln = F.layer_norm(xc, (D,), gamma.double(), beta.double(), eps)
            # what the folded form computes exactly, given the 16-bit x*gamma it reads
            mu, var = xc.mean(-1, keepdim=True), xc.var(-1, unbiased=False, keepdim=True)
            folded = ((xg.cpu().double() @ w1.double().t()) - mu * colsum.cpu().double()) / torch.sqrt(var + eps) + bias_ln.cpu().double()
            exact = ln @ w1.double().t() + b1.double()
            for epi, fn in ((L.HM_EPI_LN_STORE, lambda t: t), (L.HM_EPI_LN_GELU, lambda t: F.gelu(t))):
                o = ops.gemm(xg, w1.to(DEV), bias_ln, epi, ln_stats=fin, ln_colsum=colsum).cpu().double()
                np.testing.assert_allclose(o.numpy(), fn(folded).numpy(), atol=2e-3, rtol=2 * _ulp16(dt))
                # and the fold itself is LayerNorm -> Linear up to the 16-bit rounding of x*gamma
                assert (o - fn(exact)).abs().max() < (0.05 if dt == torch.bfloat16 else 0.01)
    finally:
        lib.hm_gemm_set_variant(-1)


@pytest.mark.parametrize("M,N,K,S", [(192, 1280, 5120, 10), (768, 1280, 1280, 4), (768, 1280, 5120, 4), (100, 320, 320, 5),
                                     (1000, 640, 448, 7)])
def test_gemm_split_k_and_layernorm_accum(M, N, K, S):
    """Small-M residual GEMM as split-K partial slabs + the LayerNorm that adds them (x += X.W^T + b; LN(x)):
    slabs against fp64 partial products, the accumulated x and its LayerNorm against the single-pass kernels, and
    bit-reproducible from run to run (fixed summation order, no atomics)."""
    x = _u("sx", (M, K), 1.0, seed=M).to(torch.bfloat16)
    w = _u("sw", (N, K), 0.05, seed=N).to(torch.bfloat16)
    bias, resid = _u("sb", (N,), 0.5, seed=K), _u("sr", (M, N), 1.0, seed=7)
    gamma, beta = _u("sg", (N,), 0.3, seed=3, center=1.0), _u("sbt", (N,), 0.2, seed=4)
    parts = ops.gemm(x.to(DEV), w.to(DEV), None, L.HM_EPI_F32, k_split=S)
    kl = K // S
    for s_ in range(S):
        ref = x[:, s_ * kl:(s_ + 1) * kl].double() @ w[:, s_ * kl:(s_ + 1) * kl].double().t()
        np.testing.assert_allclose(parts[s_].cpu().double().numpy(), ref.numpy(), atol=2e-5 * math.sqrt(kl), rtol=1e-5)
    xa = resid.to(DEV).clone()
    h = ops.layernorm_accum(xa, parts, bias.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-6)
    xb = resid.to(DEV).clone()
    ops.gemm(x.to(DEV), w.to(DEV), bias.to(DEV), L.HM_EPI_RESID_F32, resid=xb, out=xb)
    hb = ops.layernorm(xb, gamma.to(DEV), beta.to(DEV), 1e-6)
    np.testing.assert_allclose(xa.cpu().numpy(), xb.cpu().numpy(), atol=1e-5 * math.sqrt(K), rtol=1e-5)
    assert (h.float() - hb.float()).abs().max().item() <= 2 * _ulp16(torch.bfloat16) * hb.float().abs().max().item()
    xa2 = resid.to(DEV).clone()
    h2 = ops.layernorm_accum(xa2, ops.gemm(x.to(DEV), w.to(DEV), None, L.HM_EPI_F32, k_split=S), bias.to(DEV), gamma.to(DEV),
                             beta.to(DEV), 1e-6)
    assert torch.equal(xa, xa2) and torch.equal(h, h2)


def test_gemm_rejects_bad_arguments():
    x = torch.zeros(16, 96, device=DEV, dtype=torch.bfloat16)
    w = torch.zeros(16, 96, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(L.HipLibraryError):
        ops.gemm(x, w)                       # K % 64 != 0
    with pytest.raises(L.HipLibraryError):
        ops.gemm(torch.zeros(16, 64), torch.zeros(16, 64))   # host tensors: no CPU path


# ------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("D,eps", [(1280, 1e-6), (1024, 1e-5), (320, 1e-6), (256, 1e-5)])
def test_layernorm(D, eps):
    M = 197
    x = _u("lnx", (M, D), 3.0, seed=D, center=0.5)
    g, b = _u("lng", (D,), 0.1, seed=1, center=1.0), _u("lnb", (D,), 0.1, seed=2)
    ref = F.layer_norm(x, (D,), g, b, eps)
    o = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), eps, torch.float32).cpu()
    np.testing.assert_allclose(o.numpy(), ref.numpy(), atol=2e-6, rtol=1e-5)
    for dt in (torch.bfloat16, torch.float16):
        o = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), eps, dt).cpu().float()
        np.testing.assert_allclose(o.numpy(), ref.numpy(), atol=1e-6, rtol=_ulp16(dt))


# ------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,heads", [(2, 16), (3, 4)])
def test_vit_attention(B, heads, dt):
    T, hd = 192, 80
    qkv = _u("qkv", (B * T, 3 * heads * hd), 2.0, seed=heads).to(dt)
    out = ops.vit_attention(qkv.to(DEV), B, T, heads, hd, hd ** -0.5).cpu().float()
    q, k, v = qkv.float().reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-2, -1)) * hd ** -0.5
    p = torch.exp(s - s.amax(-1, keepdim=True))
    ref = ((p.to(dt).float() @ v) / p.sum(-1, keepdim=True)).transpose(1, 2).reshape(B * T, heads * hd)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=2e-3 if dt == torch.bfloat16 else 3e-4, rtol=2 * _ulp16(dt))
    # against the exact softmax (what the reference computes on these operands)
    exact = (s.softmax(-1) @ v).transpose(1, 2).reshape(B * T, heads * hd)
    np.testing.assert_allclose(out.numpy(), exact.numpy(), atol=1.5e-2 if dt == torch.bfloat16 else 2e-3, rtol=0)


def test_vit_attention_peaked_rows():
    """One key dominates each row (max far above the rest): exercises the max subtraction."""
    B, heads, T, hd = 1, 2, 192, 80
    qkv = _u("qkvp", (B * T, 3 * heads * hd), 0.5, seed=9)
    qkv[:, :heads * hd] *= 8.0
    qkv[17, heads * hd:2 * heads * hd] *= 12.0
    qkv = qkv.to(torch.bfloat16)
    out = ops.vit_attention(qkv.to(DEV), B, T, heads, hd, hd ** -0.5).cpu().float()
    q, k, v = qkv.float().reshape(B, T, 3, heads, hd).permute(2, 0, 3, 1, 4)
    exact = (((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(-1) @ v).transpose(1, 2).reshape(B * T, heads * hd)
    assert torch.isfinite(out).all()
    np.testing.assert_allclose(out.numpy(), exact.numpy(), atol=1.5e-2, rtol=0)


# ------------------------------------------------------------------------------------ patch gather
def test_patch_im2col_matches_conv_unfold():
    B = 3
    img = _u("img", (B, 3, 256, 256), 2.0, seed=4)
    for dt in (torch.bfloat16, torch.float16):
        pat = ops.patch_im2col(img.to(DEV), 32, 192, 16, 2, dt).cpu().float()
        ref = F.unfold(img[:, :, :, 32:-32].to(dt).float(), kernel_size=16, stride=16, padding=2)  # (B, 768, 192)
        ref = ref.transpose(1, 2).reshape(B * 192, 768)
        assert torch.equal(pat, ref)


# ------------------------------------------------------------------------------------ decoder pieces
@pytest.mark.parametrize("M,N,K", [(64, 1024, 1024), (5, 112, 1024), (33, 512, 256), (1, 256, 512)])
def test_linear_f32(M, N, K):
    x, w = _u("lx", (M, K), 1.0, seed=M), _u("lw", (N, K), K ** -0.5, seed=N)
    b, r = _u("lb", (N,), 0.1, seed=1), _u("lr", (M, N), 1.0, seed=2)
    xd, wd, bd, rd = x.to(DEV), w.to(DEV), b.to(DEV), r.to(DEV)
    ref = (x.double() @ w.double().t() + b.double())
    np.testing.assert_allclose(ops.linear_f32(xd, wd, bd).cpu().numpy(), ref.float().numpy(), atol=3e-6 * math.sqrt(K), rtol=1e-5)
    np.testing.assert_allclose(ops.linear_f32(xd, wd, bd, rd, act=1).cpu().numpy(),
                               (F.gelu(ref.float()) + r).numpy(), atol=3e-6 * math.sqrt(K), rtol=1e-5)
    np.testing.assert_allclose(ops.linear_f32(xd, wd).cpu().numpy(), (x.double() @ w.double().t()).float().numpy(),
                               atol=3e-6 * math.sqrt(K), rtol=1e-5)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_cross_attention(dt):
    B, T, heads, dh, layers = 5, 192, 8, 64, 3
    inner = heads * dh
    q = _u("caq", (B, inner), 2.0, seed=1)
    kv = _u("cakv", (B * T, layers * 2 * inner), 1.5, seed=2).to(dt)
    li = 1
    out = ops.cross_attention(q.to(DEV), kv.to(DEV), li * 2 * inner, li * 2 * inner + inner, B, T, heads, dh, dh ** -0.5).cpu()
    kvf = kv.float().reshape(B, T, layers, 2, heads, dh)
    k, v = kvf[:, :, li, 0].permute(0, 2, 1, 3), kvf[:, :, li, 1].permute(0, 2, 1, 3)   # (B,h,T,dh)
    a = ((q.reshape(B, heads, 1, dh) @ k.transpose(-1, -2)) * dh ** -0.5).softmax(-1)
    ref = (a @ v).reshape(B, inner)
    np.testing.assert_allclose(out.numpy(), ref.numpy(), atol=2e-5, rtol=1e-4)


# ------------------------------------------------------------------------------------ MANO tail
def test_mano_forward_vs_oracle_and_manopth(golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "mano_manopth.npz"))
    mp = synth.mano_params(seed=int(g["mano_seed"]))
    mpd = {k: v.to(DEV) for k, v in mp.items() if v.dtype == torch.float32}
    B = 37
    pose6d = _u("m6", (B, 96), 1.0, seed=7)
    betas = _u("mb", (B, 10), 2.0, seed=8)
    cam = _u("mc", (B, 3), 0.2, seed=9) + torch.tensor([0.9, 0.0, 0.0])
    o = ops.mano_forward(mpd, pose6d.to(DEV), betas.to(DEV), cam.to(DEV))
    Rr = R.rot6d_to_rotmat(pose6d).view(B, 16, 3, 3)
    verts, joints = R.mano_forward(mp, betas, Rr)
    cam_t = torch.stack([cam[:, 1], cam[:, 2], 2 * 5000.0 / (256.0 * cam[:, 0] + 1e-9)], -1)
    kp2d = R.perspective_projection(joints, cam_t, torch.full((B, 2), 5000.0 / 256.0))
    np.testing.assert_allclose(o["rotmats"].cpu().numpy(), Rr.numpy(), atol=2e-6)
    np.testing.assert_allclose(o["verts"].cpu().numpy(), verts.numpy(), atol=3e-6)
    np.testing.assert_allclose(o["joints"].cpu().numpy(), joints.numpy(), atol=3e-6)
    np.testing.assert_allclose(o["cam_t"].cpu().numpy(), cam_t.numpy(), rtol=1e-6)
    np.testing.assert_allclose(o["kp2d"].cpu().numpy(), kp2d.numpy(), atol=1e-5, rtol=1e-5)
    # golden: the in-tree manopth layer on the same parameters.  Drive the kernel with a 6-D pose
    # equal to the first two columns of the golden rotation matrices.
    Rg = torch.from_numpy(g["rotmats"])                       # (4,16,3,3)
    six = torch.cat([Rg[..., 0], Rg[..., 1]], dim=-1).reshape(4, 96)
    o = ops.mano_forward(mpd, six.to(DEV), torch.from_numpy(g["betas"]).to(DEV), cam[:4].to(DEV))
    np.testing.assert_allclose(o["rotmats"].cpu().numpy(), g["rotmats"], atol=3e-6)
    np.testing.assert_allclose(o["verts"].cpu().numpy(), g["verts"], atol=5e-6)


# ------------------------------------------------------------------------------------ crop (byte-exact)
def test_crop_batch_bit_exact_vs_oracle():
    H, W = 540, 960
    frame = (synth._hash_u32(torch.arange(H * W * 3, dtype=torch.int64), 77) >> 24).to(torch.uint8).reshape(H, W, 3)
    dets = [["right", [100.0, 120.0, 260.0, 300.0]], ["left", [500.0, 200.0, 640.0, 330.0]],
            ["left", [-20.0, -30.0, 90.0, 80.0]], ["right", [880.0, 470.0, 1000.0, 560.0]], ["right", [300.5, 100.25, 420.75, 260.5]]]
    mean = 255.0 * np.array([0.485, 0.456, 0.406]); std = 255.0 * np.array([0.229, 0.224, 0.225])
    ref = crop_ref.prepare_batch_bbox(frame.numpy(), dets, mean, std)
    boxes = []
    for label, (x1, y1, x2, y2) in dets:
        cx, cy, S = crop_ref.bbox_to_center_size(x1, y1, x2, y2)
        boxes.append((cx, cy, S, label != "right"))
    rec = ops.crop_boxes(boxes)
    out = ops.crop_batch(frame.to(DEV), rec.to(DEV), mean, std).cpu().numpy()
    assert out.shape == ref["img"].shape
    assert np.array_equal(out, ref["img"]), f"max diff {np.abs(out - ref['img']).max()}"


def test_crop_identity_sampling():
    """A 256-px box centred on an integer pixel samples source pixels exactly (SURVEY 8a KAT)."""
    H, W = 400, 500
    frame = (synth._hash_u32(torch.arange(H * W * 3, dtype=torch.int64), 5) >> 24).to(torch.uint8).reshape(H, W, 3)
    rec = ops.crop_boxes([(250.0, 200.0, 256.0, False)])
    out = ops.crop_batch(frame.to(DEV), rec.to(DEV), [0, 0, 0], [1, 1, 1]).cpu()
    src = frame[200 - 128:200 + 128, 250 - 128:250 + 128].permute(2, 0, 1).float().flip(0)   # BGR -> RGB
    assert torch.equal(out[0], src)
