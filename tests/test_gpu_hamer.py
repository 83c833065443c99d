"""GPU parity of the whole HaMeR forward (hm_hamer_forward through the C ABI).

Fixtures in tests/golden/ were produced by the reference's own modules (tools/gen_golden.py) on fp32 master
weights, as a real checkpoint holds them (hamer/hamer/models/__init__.py:46); the engine rounds them to its 16-bit
operand type itself.  The tolerance of the headline checks is north_star's 1e-3 abs on MANO theta (rotation
matrices) / beta and on the 778 vertices, against the fp32 CPU path on the same inputs -- met by the default operand
type fp16; bf16 (8 significant bits) stays selectable and its distance is asserted against the looser, documented
bound BF16_TOL.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hamer_yolo_amd import lib as L
from hamer_yolo_amd import synth
from hamer_yolo_amd.engine import HamerEngine
from oracle import hamer_ref as R

TOL = 1e-3        # BASELINE.json north_star: "within 1e-3 abs" (fp16 operands, the default)
BF16_TOL = 4e-3   # bf16 operands vs the fp32 path on fp32 master weights: weight rounding alone is ~1.5e-3


def _tol(dtype):
    return TOL if dtype == torch.float16 else BF16_TOL


def _engine(cfg, seed, dtype=torch.float16, mano_seed=0, fold_ln=None):
    sd = synth.hamer_state_dict(cfg, seed=seed)
    mp = synth.mano_params(seed=mano_seed)
    return HamerEngine(sd, mp, cfg, dtype=dtype, fold_ln=fold_ln), sd, mp


def _report(name, **vals):
    """Achieved distances, kept next to the test log (gpurun_out/ travels back from the GPU box)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps({"test": name, **{k: float(v) for k, v in vals.items()}}) + "\n")
    except OSError:
        pass
    print(name, {k: f"{float(v):.2e}" for k, v in vals.items()})


# fold_ln: LayerNorm deferred into the neighbouring GEMMs, or run as its own kernel (the default)
@pytest.mark.parametrize("fold_ln", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tiny_forward_vs_reference_golden(golden_dir, dtype, fold_ln):
    g = np.load(os.path.join(golden_dir, "hamer_tiny.npz"))
    cfg = synth.tiny_config()
    eng, sd, mp = _engine(cfg, int(g["seed"]), dtype, fold_ln=fold_ln)
    assert eng.fold_ln == fold_ln
    img = synth.normalize_crops(synth.crops_u8(3, seed0=int(g["crop_seed0"])))
    out = eng.forward(img.cuda(), want_tokens=True)
    torch.cuda.synchronize()
    tok = out["tokens"].float().cpu().reshape(3, 192, -1).numpy()
    assert np.abs(tok - g["tokens"]).max() < (6e-2 if dtype == torch.bfloat16 else 8e-3)
    tol = _tol(dtype)
    np.testing.assert_allclose(out["pose6d"].cpu().numpy(), g["pose6d"], atol=tol, rtol=0)
    np.testing.assert_allclose(out["betas"].cpu().numpy(), g["betas"], atol=tol, rtol=0)
    np.testing.assert_allclose(out["pred_cam"].cpu().numpy(), g["cam"], atol=tol, rtol=0)
    np.testing.assert_allclose(out["rotmats"].cpu().numpy(), g["rotmats"], atol=tol, rtol=0)
    # 16-bit-emulating oracle: same rounding points, so the gap is accumulation order only
    with torch.no_grad():
        # (the explicit-LN path only: deferred LN rounds x*gamma instead of LN(x))
        emu = R.hamer_forward(sd, mp, img, cfg, emu="bf16" if dtype == torch.bfloat16 else "fp16") if not fold_ln else None
        ref = R.hamer_forward(sd, mp, img, cfg, emu=False)
    for k_out, k_ref in (("pred_vertices", "pred_vertices"), ("pred_keypoints_3d", "pred_keypoints_3d"),
                         ("pred_keypoints_2d", "pred_keypoints_2d")):
        np.testing.assert_allclose(out[k_out].cpu().numpy(), ref[k_ref].numpy(), atol=tol, rtol=1e-3)
    np.testing.assert_allclose(out["pred_cam_t"].cpu().numpy(), ref["pred_cam_t"].numpy(), rtol=2e-3)
    if emu is not None:
        np.testing.assert_allclose(out["pose6d"].cpu().numpy(), emu["pose6d"].numpy(), atol=3e-4, rtol=0)
        np.testing.assert_allclose(out["pred_vertices"].cpu().numpy(), emu["pred_vertices"].numpy(), atol=1e-4, rtol=0)


@pytest.mark.parametrize("fold_ln", [True, False])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_vith_forward_vs_reference_golden(golden_dir, dtype, fold_ln):
    """Full ViT-H/16 + 6-layer decoder on fp32 master weights; expected values from the reference vit.py / pose_transformer.py."""
    g = np.load(os.path.join(golden_dir, "hamer_vith.npz"))
    assert int(g["bf16_representable"]) == 0
    cfg = synth.HamerConfig()
    sd = synth.hamer_state_dict(cfg, seed=int(g["seed"]), device="cuda")
    mp = synth.mano_params(seed=0)
    eng = HamerEngine(sd, mp, cfg, dtype=dtype, fold_ln=fold_ln)
    nb = g["pose6d"].shape[0]
    img = synth.normalize_crops(synth.crops_u8(nb, seed0=int(g["crop_seed0"])))
    out = eng.forward(img.cuda(), want_tokens=True)
    torch.cuda.synchronize()
    tol = _tol(dtype)
    tok = out["tokens"].float().cpu().reshape(nb, 192, 1280)
    assert np.abs(tok[:, ::16, ::40].numpy() - g["tokens_sub"]).max() < (1e-1 if dtype == torch.bfloat16 else 1.5e-2)
    np.testing.assert_allclose(tok.mean((1, 2)).numpy(), g["tokens_mean"], atol=2e-3)
    verts, joints = R.mano_forward(mp, torch.from_numpy(g["betas"]), torch.from_numpy(g["rotmats"]))   # MANO itself is pinned against manopth
    d = {k: np.abs(out[k].cpu().numpy() - e).max() for k, e in (("pose6d", g["pose6d"]), ("betas", g["betas"]), ("pred_cam", g["cam"]),
                                                               ("rotmats", g["rotmats"]), ("pred_vertices", verts.numpy()),
                                                               ("pred_keypoints_3d", joints.numpy()))}
    _report(f"vith_golden[{str(dtype).split('.')[-1]},fold_ln={fold_ln}]", **d)
    for k, v in d.items():
        assert v < (TOL if k in ("pred_vertices", "pred_keypoints_3d") else tol), (k, v)      # the mesh is within 1e-3 for either type


def test_batch64_vs_fp32_oracle_at_production_tile():
    """BASELINE configs[1] as benchmarked: 64 different crops, fp32 master weights, the default operand type (fp16) and
    the production GEMM tile (M = 12288 -> gemm_x3_kernel), against the fp32 CPU oracle on the same 64 crops:
    theta (rotation matrices), beta and the 778 vertices within north_star's 1e-3."""
    cfg = synth.HamerConfig()
    sd = synth.hamer_state_dict(cfg, seed=0, device="cuda")
    mp = synth.mano_params(seed=0)
    eng = HamerEngine(sd, mp, cfg)
    assert eng.dtype == torch.float16
    img = synth.normalize_crops(synth.crops_u8(64, seed0=1000))
    out = {k: v.cpu() for k, v in eng.forward(img.cuda()).items()}
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    with torch.no_grad():
        ref = R.hamer_forward(sd_cpu, mp, img, cfg)
    rot = torch.cat([ref["global_orient"], ref["hand_pose"]], 1)
    d = {"rotmats": (out["rotmats"] - rot).abs().max(), "betas": (out["betas"] - ref["betas"]).abs().max(),
         "pose6d": (out["pose6d"] - ref["pose6d"]).abs().max(), "pred_cam": (out["pred_cam"] - ref["pred_cam"]).abs().max(),
         "pred_vertices": (out["pred_vertices"] - ref["pred_vertices"]).abs().max(),
         "pred_keypoints_3d": (out["pred_keypoints_3d"] - ref["pred_keypoints_3d"]).abs().max()}
    _report("batch64_fp16_vs_fp32_oracle", **d)
    for k, v in d.items():
        assert float(v) < TOL, (k, float(v))


def test_checkpoint_that_overflows_fp16_stays_within_1e3_by_prescale():
    """VERDICT r3 item 4 (reference: the fp32 hamer.ckpt of hamer/hamer/models/__init__.py:46, tolerance hamer/infer.py:730).
    A ViT-H checkpoint whose 16-bit activations leave fp16 in five different classes -- LayerNorm output (gamma x 2^16 against
    fc1.weight x 2^-16: the same function), the value rows of qkv (x 2^17 against proj.weight x 2^-17: the same function), the
    GELU output (fc1 x 2^16, fc2.weight x 2^-16: another, equally valid network), one decoder layer's to_kv value rows and
    last_norm -- gives NaN with plain fp16 operands and 1.2-1.9e-3 on theta / beta with the old bfloat16 fallback.  HAMER.to()
    now measures the ranges with a bf16 engine, folds powers of two into the weights and keeps fp16 operands: theta, beta and the
    vertices within north_star's 1e-3 of the fp32 CPU oracle on the SAME (unscaled-by-us) checkpoint."""
    from hamer_yolo_amd.hamer.configs import get_config
    from hamer_yolo_amd.hamer.models.hamer import HAMER
    from hamer_yolo_amd.hamer.models.mano_wrapper import MANO
    cfg = synth.HamerConfig()
    sd = synth.hamer_state_dict(cfg, seed=0)
    D, inner = cfg.vit.embed_dim, cfg.dec.inner
    p = "backbone.blocks."
    sd[p + "3.norm2.weight"] = sd[p + "3.norm2.weight"] * 2.0 ** 16; sd[p + "3.norm2.bias"] = sd[p + "3.norm2.bias"] * 2.0 ** 16
    sd[p + "3.mlp.fc1.weight"] = sd[p + "3.mlp.fc1.weight"] * 2.0 ** -16
    w, b = sd[p + "7.attn.qkv.weight"].clone(), sd[p + "7.attn.qkv.bias"].clone()
    w[2 * D:] *= 2.0 ** 17; b[2 * D:] *= 2.0 ** 17
    sd[p + "7.attn.qkv.weight"], sd[p + "7.attn.qkv.bias"] = w, b
    sd[p + "7.attn.proj.weight"] = sd[p + "7.attn.proj.weight"] * 2.0 ** -17
    sd[p + "11.mlp.fc1.weight"] = sd[p + "11.mlp.fc1.weight"] * 2.0 ** 16; sd[p + "11.mlp.fc1.bias"] = sd[p + "11.mlp.fc1.bias"] * 2.0 ** 16
    sd[p + "11.mlp.fc2.weight"] = sd[p + "11.mlp.fc2.weight"] * 2.0 ** -16
    t = "mano_head.transformer.transformer.layers.2.1.fn."
    w = sd[t + "to_kv.weight"].clone(); w[inner:] *= 2.0 ** 17
    sd[t + "to_kv.weight"] = w
    sd[t + "to_out.0.weight"] = sd[t + "to_out.0.weight"] * 2.0 ** -17
    sd["backbone.last_norm.weight"] = sd["backbone.last_norm.weight"] * 2.0 ** 15; sd["backbone.last_norm.bias"] = sd["backbone.last_norm.bias"] * 2.0 ** 15
    for i in range(cfg.dec.depth):
        k = f"mano_head.transformer.transformer.layers.{i}.1.fn.to_kv.weight"
        sd[k] = sd[k] * 2.0 ** -15
    mano = MANO.synthetic(0)
    plain = HamerEngine({k: v.cuda() for k, v in sd.items()}, mano.params, cfg)
    assert not plain.calibration_is_finite()                           # fp16 operands as they are: overflow
    del plain
    m = HAMER(get_config(None), sd, mano, hamer_cfg=cfg)
    with pytest.warns(UserWarning, match="rescaled by powers of two"):
        m.to("cuda")
    assert m.dtype == torch.float16
    pre = m._engine.prescale
    assert pre["blocks"][3]["ln2"] >= 4 and pre["blocks"][7]["v"] >= 4 and pre["blocks"][11]["gelu"] >= 4 and pre["last"] >= 3 and pre["dec"][2][1] >= 4
    img = synth.normalize_crops(synth.crops_u8(8, seed0=2000))
    out, _ = m({"img": img.cuda()})
    torch.cuda.synchronize()
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        ref = R.hamer_forward(sd, mano.params, img, cfg)
    rot = torch.cat([ref["global_orient"], ref["hand_pose"]], 1)
    got = torch.cat([out["pred_mano_params"]["global_orient"], out["pred_mano_params"]["hand_pose"]], 1).cpu()
    d = {"rotmats": float((got - rot).abs().max()), "betas": float((out["pred_mano_params"]["betas"].cpu() - ref["betas"]).abs().max()),
         "pred_vertices": float((out["pred_vertices"].cpu() - ref["pred_vertices"]).abs().max())}
    _report("overflowing_checkpoint_fp16_prescale_vs_fp32_oracle", **d)
    for k, v in d.items():
        assert v < TOL, (k, v)
    # and on a checkpoint that needs nothing the prescale machinery is the identity: exponents all zero -> no prescaled engine
    from hamer_yolo_amd.engine import prescale_from_ranges, prescale_is_identity
    sd0 = synth.hamer_state_dict(cfg, seed=0, device="cuda")
    probe = HamerEngine(sd0, mano.params, cfg, dtype=torch.bfloat16)
    ranges = probe.measure_ranges()
    assert bool(torch.isfinite(ranges).all()) and float(ranges.max()) < 4096 and float(ranges.min()) > 0
    assert prescale_is_identity(prescale_from_ranges(ranges, cfg.vit.depth, cfg.dec.depth))


def test_batch64_is_batch_invariant():
    """BASELINE config 2 size (B=64): per-crop results do not depend on the batch they ride in -- repeated crops give
    identical rows, a batch of 16 (other GEMM tile shapes, same K order) gives the same numbers to fp32 rounding, a batch of 8 (split-K
    path: other summation order, other rounding flips) stays far inside the 1e-3 parity bar -- and the output is finite."""
    cfg = synth.HamerConfig()
    sd = synth.hamer_state_dict(cfg, seed=0, device="cuda")
    eng = HamerEngine(sd, synth.mano_params(seed=0), cfg)
    u8 = synth.crops_u8(16, seed0=0)
    img16 = synth.normalize_crops(u8).cuda()
    img64 = img16.repeat(4, 1, 1, 1)
    o16 = {k: v.clone() for k, v in eng.forward(img16).items()}
    o8 = {k: v.clone() for k, v in eng.forward(img16[:8].contiguous()).items()}
    o64 = eng.forward(img64)
    torch.cuda.synchronize()
    for k in ("pose6d", "betas", "pred_cam", "pred_vertices", "pred_keypoints_3d"):
        a, b = o16[k], o64[k]
        assert torch.isfinite(b).all()
        assert torch.equal(b[:16], b[48:64]), k
        # (round 3: at B = 64 the residual rows are added inside the K loop of proj / fc2, at B = 16 behind it -- one fp32 add
        # per element changes position, 64 times over the backbone, and every flipped fp16 rounding of h / qkv / the MLP hidden
        # behind it is a 5e-4 relative step: measured 3.6e-5 on the outputs)
        np.testing.assert_allclose(b[:16].cpu().numpy(), a.cpu().numpy(), atol=2e-4, rtol=0)
        np.testing.assert_allclose(o8[k].cpu().numpy(), b[:8].cpu().numpy(), atol=5e-4, rtol=0)
    # Round 4: the tight statement, per kernel family.  (i) Batches that take the SAME kernels give the same bytes: 64 vs 128
    # hands (whole 256-row tiles: persistent store / GELU GEMMs, in-loop residual), 16 vs 32 hands (128 x 128 tile, residual in the
    # epilogue).  (ii) What separates 64 from 16 is one thing only, the position of the fp32 residual add (inside the K loop of
    # gemm_x3r_kernel from whole tiles on; behind it otherwise): with HM_OPT_RESID_IN_EPILOGUE = 1 -- the round-2 form, same
    # position at every batch size -- 64 hands equal 16 hands to 2e-5 again.
    o32 = {k: v.clone() for k, v in eng.forward(img16.repeat(2, 1, 1, 1)).items()}
    o128 = {k: v.clone() for k, v in eng.forward(img16.repeat(8, 1, 1, 1)).items()}
    with L.option(L.HM_OPT_RESID_IN_EPILOGUE, 1):
        o64e = {k: v.clone() for k, v in eng.forward(img64).items()}
    torch.cuda.synchronize()
    for k in ("pose6d", "betas", "pred_cam", "pred_vertices", "pred_keypoints_3d"):
        assert torch.equal(o128[k][:16], o64[k][:16]) and torch.equal(o128[k][112:], o64[k][:16]), k
        np.testing.assert_allclose(o32[k][:16].cpu().numpy(), o16[k].cpu().numpy(), atol=2e-5, rtol=0)
        np.testing.assert_allclose(o64e[k][:16].cpu().numpy(), o16[k].cpu().numpy(), atol=2e-5, rtol=0)
    r = o64["rotmats"]
    eye = torch.eye(3, device="cuda").expand_as(r)
    np.testing.assert_allclose((r @ r.transpose(-1, -2)).cpu().numpy(), eye.cpu().numpy(), atol=1e-5)


def test_batches_in_flight_do_not_interfere():
    """HamerEngine.contexts: four different batches issued back to back on two contexts (own stream, workspace, outputs)
    give, each, bit for bit what a lone forward gives."""
    cfg = synth.tiny_config()
    sd = synth.hamer_state_dict(cfg, seed=5, device="cuda")
    eng = HamerEngine(sd, synth.mano_params(seed=5), cfg)
    B = 6
    imgs = [synth.normalize_crops(synth.crops_u8(B, seed0=100 + 10 * i)).cuda() for i in range(4)]
    ref = []
    for im in imgs:
        o = eng.forward(im, want_tokens=True)
        torch.cuda.synchronize()
        ref.append({k: v.clone() for k, v in o.items()})
    ctxs = eng.contexts(B, 2, want_tokens=True)
    got = []
    for rnd in range(2):
        for k in range(2):
            eng.forward_on(ctxs[k], imgs[2 * rnd + k], want_tokens=True)
        for k in range(2):
            ctxs[k].stream.synchronize()
            got.append({n: v.clone() for n, v in ctxs[k].out.items()})
    for r, g_ in zip(ref, got):
        for n in r:
            assert torch.equal(r[n], g_[n]), n


# ------------------------------------------------------------------------------------ token merging
def test_tome_kernels_vs_oracle():
    """hm_tome_attention (any token count, proportional attention) and hm_tome_merge (metric, bipartite matching, size-weighted
    merge) against oracle/tome_ref.py on the same inputs: indices exact, values to fp32 / 16-bit rounding."""
    from hamer_yolo_amd import ops
    from oracle import tome_ref as T
    B, H, d, D = 3, 4, 80, 320
    for tokens, r, with_size in ((192, 16, False), (161, 14, True), (176, 15, True), (23, 8, True), (17, 4, True), (16, 4, True),
                                 (3, 1, True), (1, 0, True)):
        qkv = synth.uniform("tq", (B * tokens, 3 * H * d), 1.2, seed=tokens).half()
        size = (1.0 + (synth._hash_u32(torch.arange(B * tokens, dtype=torch.int64), 7) % 4).float()) if with_size else None
        x = synth.uniform("tx", (B * tokens, D), 1.0, seed=tokens + 1)
        # attention
        got = ops.tome_attention(qkv.cuda(), size.cuda() if with_size else None, B, tokens, H, d, d ** -0.5).float().cpu()
        q, k, v = qkv.float().reshape(B, tokens, 3, H, d).permute(2, 0, 3, 1, 4)
        a = (q @ k.transpose(-1, -2)) * d ** -0.5
        if with_size:
            a = a + size.reshape(B, 1, 1, tokens).log()
        ref = (a.softmax(-1) @ v).transpose(1, 2).reshape(B * tokens, H * d)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=2e-3, rtol=2e-3)
        assert torch.isfinite(got).all()
        if r == 0:
            continue
        # matching + merge
        xo, so, index, metric = ops.tome_merge(qkv.cuda(), x.cuda(), size.cuda() if with_size else None, B, tokens, r, H, d)
        torch.cuda.synchronize()
        m_ref = k.mean(1)                                                    # (B, tokens, d)
        np.testing.assert_allclose(metric.cpu().numpy(), m_ref.numpy(), atol=1e-6)
        unm, src, dst = T.bipartite_match(metric.cpu(), r)                   # on the GPU's own metric: decisions must be identical
        na = (tokens + 1) // 2
        idx = index.cpu().long()
        assert torch.equal(idx[:, 0, :na - r], unm) and torch.equal(idx[:, 1, :r], src) and torch.equal(idx[:, 2, :r], dst)
        sz = size.reshape(B, tokens, 1) if with_size else torch.ones(B, tokens, 1)
        x_ref, s_ref = T.merge_tokens(x.reshape(B, tokens, D), sz, unm, src, dst)
        np.testing.assert_allclose(xo.cpu().reshape(B, tokens - r, D).numpy(), x_ref.numpy(), atol=2e-6, rtol=1e-6)
        np.testing.assert_array_equal(so.cpu().reshape(B, tokens - r, 1).numpy(), s_ref.numpy())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_tome_forward_vs_oracle_and_reference_golden(golden_dir, dtype):
    """The whole forward with token merging (HAMER_INFER(token_merge=True): apply_patch, r = (8, -1)) on the 6-block geometry:
    192 -> 176 -> 164 -> 155 -> 149 -> 146 tokens.  Matching is a discrete decision on 16-bit keys, so the tight comparison is
    with the oracle in the kernel's arithmetic (same keys, same decisions); the reference-module golden (fp32) is the second,
    looser bound."""
    from oracle import tome_ref as T
    g = np.load(os.path.join(golden_dir, "hamer_tome.npz"))
    cfg = synth.tome_tiny_config()
    sd = synth.hamer_state_dict(cfg, seed=int(g["seed"]))
    mp = synth.mano_params(seed=0)
    eng = HamerEngine(sd, mp, cfg, dtype=dtype, token_merge=True)
    assert eng.tome_r == list(g["r_list"]) and eng.ctx_tokens == 146
    img = synth.normalize_crops(synth.crops_u8(3, seed0=int(g["crop_seed0"])))
    out = eng.forward(img.cuda(), want_tokens=True)
    torch.cuda.synchronize()
    tok = out["tokens"].float().cpu()[:3 * 146].reshape(3, 146, -1)
    emu = "fp16" if dtype == torch.float16 else "bf16"
    with torch.no_grad():
        feats = T.vit_forward_tome(sd, img[:, :, :, 32:-32], cfg.vit, (8, -1), emu=emu)
        pose, betas, cam = R.mano_head_forward(sd, R._q(feats, emu), cfg.dec, emu)
    # The merged sequence is compared as a SET: where two proposals have (nearly) equal scores, a last-bit difference in the
    # 16-bit keys swaps their rank, i.e. the position of two unmerged tokens -- harmless unless one of them takes part in a later
    # merge (the decoder's cross-attention does not see token order).  And a near-tie can flip a MERGE: the decision logic
    # itself is pinned exactly in test_tome_kernels_vs_oracle (the oracle's matching on the GPU's own metric), but which side
    # of a tie a forward lands on depends on last-bit rounding of everything before it (measured: the fp32 lane-per-key
    # attention kernel and the MFMA one are equally close to fp64 -- 0.11 % / 0.16 % of outputs off the correctly rounded
    # value -- and flip different crops).  So: fp16 -- most crops are exact permutations of the oracle's tokens, every crop's
    # regressed parameters stay close; bf16 keys are coarser, only the regressed parameters are bounded.
    d_pose = float((out["pose6d"].cpu() - pose).abs().max())
    g_pose = float(np.abs(out["pose6d"].cpu().numpy() - g["pose6d"]).max())
    perm_crops, d_set = 0, 0.0
    for b in range(3):
        dist = torch.cdist(tok[b], feats[b])
        nn = dist.argmin(1)
        if sorted(nn.tolist()) == list(range(146)):
            perm_crops += 1
            d_set = max(d_set, float(dist.min(1).values.max()))
    _report(f"tome_forward[{emu}]", crops_that_are_permutations=perm_crops, tokens_set_distance_vs_emu=d_set, pose6d_vs_emu=d_pose,
            pose6d_vs_reference_fp32=g_pose)
    if dtype == torch.float16:
        # round 3 (matching metric formed in fp32 from the fp32 LayerNorm output): every crop is an exact permutation of the
        # oracle's token set and the regressed pose agrees to fp32 noise (measured 3 / 3, 7.9e-6)
        assert perm_crops == 3 and d_set < 5e-2 and d_pose < 2e-4
    assert d_pose < 1e-2 and g_pose < 1e-2
    assert torch.isfinite(out["pred_vertices"]).all()


def test_tome_vith_geometry_vs_oracle_nondegenerate_schedule():
    """ToMe at the ViT-H geometry against the oracle with a schedule that keeps a usable sequence: r = 8 for the first 8 blocks,
    then none (192 -> 128 tokens; the reference's own (8, -1) collapses ViT-H to one token, next test).  Round 3: the matching
    metric is formed in fp32 from the LayerNorm output (hm_vit_block.kmean_w), so merge decisions track the fp32 reference;
    compared with the oracle in the kernels' arithmetic (token sets, regressed parameters) and with the fp32 oracle."""
    from oracle import tome_ref as T
    cfg = synth.HamerConfig()
    sd = synth.hamer_state_dict(cfg, seed=0)
    mp = synth.mano_params(seed=0)
    sched = [8] * 8 + [0] * 24
    eng = HamerEngine(sd, mp, cfg, token_merge=sched)
    assert eng.ctx_tokens == 128
    img = synth.normalize_crops(synth.crops_u8(2, seed0=40))
    out = eng.forward(img.cuda(), want_tokens=True)
    torch.cuda.synchronize()
    tok = out["tokens"].float().cpu()[:2 * 128].reshape(2, 128, -1)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        feats = T.vit_forward_tome(sd, img[:, :, :, 32:-32], cfg.vit, sched, emu="fp16")
        pose, betas, cam = R.mano_head_forward(sd, R._q(feats, "fp16"), cfg.dec, "fp16")
        feats32 = T.vit_forward_tome(sd, img[:, :, :, 32:-32], cfg.vit, sched, emu=False)
        pose32, betas32, _ = R.mano_head_forward(sd, feats32, cfg.dec, False)
    perm = 0
    for b in range(2):
        nn = torch.cdist(tok[b], feats[b]).argmin(1)
        perm += int(sorted(nn.tolist()) == list(range(128)))
    d_emu = float((out["pose6d"].cpu() - pose).abs().max())
    d_32 = float((out["pose6d"].cpu() - pose32).abs().max())
    d_b32 = float((out["betas"].cpu() - betas32).abs().max())
    _report("tome_vith_r8x8", crops_that_are_permutations=perm, pose6d_vs_emu=d_emu, pose6d_vs_fp32_oracle=d_32, betas_vs_fp32_oracle=d_b32)
    assert torch.isfinite(out["pred_vertices"]).all()
    # measured (round 3): 1.3e-2 vs the oracle in the kernels' arithmetic, 2.2e-2 vs the fp32 oracle, no crop an exact
    # permutation: 8 blocks x 96 proposals ranked per crop leave near-ties that last-bit differences of the fp32 residual stream
    # (accumulation order of the 16-bit GEMMs) resolve differently, and with random-init weights ONE flipped merge moves the
    # regressed pose by ~1e-2.  On the 6-block geometry the same code tracks its oracle to 8e-6 (test above).
    # Bounds = the measured level with a 1.6x margin (round 4; they were a blanket 5e-2): a merge BUG -- wrong partner, wrong size
    # weight, a dropped token -- moves the pose by 1e-1 and more on these weights (debug runs of tools/debug_tome.py).
    assert d_emu < 2.1e-2 and d_32 < 3.5e-2 and d_b32 < 2.1e-2


def test_tome_vith_schedule_runs_to_one_token():
    """ViT-H with the reference's schedule: parse_r(32, (8, -1)) removes 241 of 192 tokens -- capped at half per block, the
    sequence collapses to ONE token by block 19 (176, 161, ... 8, 4, 2, 1): the decoder then attends over a single context
    token.  Whatever one thinks of that setting, it is what HAMER_INFER(token_merge=True) computes; check it runs, is finite,
    deterministic, and that the result differs from the dense backbone."""
    cfg = synth.HamerConfig()
    sd = synth.hamer_state_dict(cfg, seed=0, device="cuda")
    mp = synth.mano_params(seed=0)
    eng = HamerEngine(sd, mp, cfg, token_merge=True)
    assert eng.ctx_tokens == 1 and sum(eng.tome_r) == 241
    img = synth.normalize_crops(synth.crops_u8(4, seed0=0)).cuda()
    a = {k: v.clone() for k, v in eng.forward(img).items()}
    b = eng.forward(img)
    torch.cuda.synchronize()
    assert all(torch.equal(a[k], b[k]) for k in a) and torch.isfinite(a["pred_vertices"]).all()
    dense = HamerEngine(sd, mp, cfg).forward(img)
    assert float((dense["pose6d"] - a["pose6d"]).abs().max()) > 1e-3


def test_tome_late_blocks_split_k_equals_unsplit_route():
    """Once merging has left <= 1536 rows, proj / fc2 of the ToMe path are split over K (slabs added by hm_layernorm_accum, as
    in the few-hands path).  With a schedule that stops merging at 15 tokens (so that no near-tie can flip afterwards: 26
    blocks of 60-row GEMMs follow) the split and the unsplit (HM_OPT_TOME_NO_SPLITK) routes differ by fp32 summation order only."""
    cfg = synth.HamerConfig()
    sd = synth.hamer_state_dict(cfg, seed=0, device="cuda")
    eng = HamerEngine(sd, synth.mano_params(seed=0), cfg, token_merge=[90, 45, 22, 11, 6, 3])
    assert eng.ctx_tokens == 15
    img = synth.normalize_crops(synth.crops_u8(4, seed0=0)).cuda()
    a = {k: v.clone() for k, v in eng.forward(img).items()}
    with L.option(L.HM_OPT_TOME_NO_SPLITK, 1):
        c = eng.forward(img)
        torch.cuda.synchronize()
    d_pose = float((c["pose6d"] - a["pose6d"]).abs().max())
    d_vert = float((c["pred_vertices"] - a["pred_vertices"]).abs().max())
    _report("tome_split_k_vs_unsplit", pose6d=d_pose, vertices=d_vert)
    assert torch.isfinite(a["pred_vertices"]).all() and d_pose < 5e-4 and d_vert < 2e-4       # measured 1.8e-4 / 3e-5


def test_persistent_gemm_is_bit_identical_to_one_tile_kernels_at_batch64():
    """gemm_px_kernel (the default for qkv / fc1 / to_kv at B = 64: 720 / 960 / 1152 tiles on 256 persistent workgroups, LDS-DMA
    pipeline across tile boundaries, hand-counted vmcnt) rounds exactly as the one-tile kernels do, so a whole B = 64 forward
    must come out bit for bit the same with it (default) and without it (variant 24) -- on random data, three times over: any
    tile that ever read a stale or half-landed LDS slot would show up here."""
    from hamer_yolo_amd import lib as L
    cfg = synth.HamerConfig()
    sd = synth.hamer_state_dict(cfg, seed=0, device="cuda")
    eng = HamerEngine(sd, synth.mano_params(seed=0), cfg)
    img = synth.normalize_crops(synth.crops_u8(64, seed0=500)).cuda()
    lib = L.load()
    try:
        L.check(lib.hm_gemm_set_variant(24))
        ref = {k: v.clone() for k, v in eng.forward(img, want_tokens=True).items()}
        L.check(lib.hm_gemm_set_variant(-1))
        for _ in range(3):
            out = eng.forward(img, want_tokens=True)
            torch.cuda.synchronize()
            for k in ref:
                assert torch.equal(out[k], ref[k]), k
    finally:
        lib.hm_gemm_set_variant(-1)
