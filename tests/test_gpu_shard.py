"""BASELINE configs[3] on one rank: the shard job's step (hamer_yolo_amd.shard.ShardJob -- what `bench.py --workload
shard1024` times) over a seeded crop set with a ragged tail: forwards of 64 on two alternating in-flight contexts, the last
forward short, pack_mano -> gather_mano(n_total=...), and the gathered rows against the fp32 CPU oracle
(oracle/hamer_ref.py) within the north-star tolerance of 1e-3.  The reference processes one hand per forward, serially
(hamer/infer.py:1268-1274); the N > 1 collectives are covered on gloo in tests/test_host_logic.py."""
import numpy as np
import pytest
import torch

from hamer_yolo_amd import shard, synth
from hamer_yolo_amd.engine import HamerEngine
from oracle import hamer_ref as R

pytestmark = pytest.mark.gpu
TOL = 1e-3


def test_shard_job_step_matches_the_oracle_with_a_ragged_tail():
    cfg = synth.HamerConfig()
    sd = synth.hamer_state_dict(cfg, seed=0, device="cuda")
    mano = synth.mano_params(seed=0)
    eng = HamerEngine(sd, mano, cfg)
    n = 360                                              # 5 forwards of 64 on alternating contexts + a ragged one of 40
    lo, hi = shard.shard_range(n, 0, 1)
    assert (lo, hi) == (0, n)
    u8 = synth.crops_u8(n, seed0=0)
    crops = synth.normalize_crops(u8)
    job = shard.ShardJob(eng, crops.cuda(), n_total=n, batch=64, in_flight=2)
    assert [b - a for a, b in job.pieces] == [64] * 5 + [40]
    first = job.step().clone()
    again = job.step()                                   # a second job on the same contexts: same rows, bit for bit
    torch.cuda.synchronize()
    assert first.shape == (n, shard.PARAMS_PER_HAND) and torch.isfinite(first).all()
    assert torch.equal(first, again)
    # 64 rows against the oracle: the head of forward 0 (context 0), rows of forward 3 (context 1), the whole region around
    # the seam between the last full forward and the ragged one
    rows = list(range(0, 16)) + list(range(200, 216)) + list(range(312, 344))
    assert len(rows) == 64
    sd_cpu = {k: v.float().cpu() for k, v in sd.items()}
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    got = first.cpu()
    worst = {"rotmats": 0.0, "betas": 0.0, "cam": 0.0}
    with torch.no_grad():
        for i in range(0, 64, 8):
            idx = rows[i:i + 8]
            ref = R.hamer_forward(sd_cpu, mano, crops[idx], cfg)
            rot = torch.cat([ref["global_orient"], ref["hand_pose"]], 1).reshape(8, 144)
            g = got[idx]
            worst["rotmats"] = max(worst["rotmats"], float((g[:, :144] - rot).abs().max()))
            worst["betas"] = max(worst["betas"], float((g[:, 144:154] - ref["betas"]).abs().max()))
            worst["cam"] = max(worst["cam"], float((g[:, 154:157] - ref["pred_cam"]).abs().max()))
    for k, v in worst.items():
        assert v < TOL, (k, v)
    # every hand is its own problem: rows of the job equal a lone forward of the same crops (other batch position)
    lone = eng.forward(crops[296:360].cuda())
    torch.cuda.synchronize()
    np.testing.assert_allclose(shard.pack_mano(lone)[24:].cpu().numpy(), got[320:360].numpy(), atol=2e-4, rtol=0)   # (other tile shapes in the short forward: fp32 order, measured 6e-5)
