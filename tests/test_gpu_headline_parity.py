"""The headline configuration itself under the oracle: BASELINE.json configs[1] -- R101-FPN, one 2048 x 2048 tile, the
default ``f16x2`` arithmetic (P32 activations), threshold 0.3, K = 2 -- first ``predictor(tile)`` against the Detectron2
restatement, then the whole per-tile path (class loop, dedup, contours, 12 measurements) against the dense CPU pipeline
(``oracle/tile_parity.py``, the same functions ``bench.py`` uses for its ``parity`` field).

Tolerances (north_star): same instances in the same order, scores within 1e-4, mask IoU >= 0.999, every measurement
within 1e-4 relative.  Plus the batch behaviour of the f16x2 mode: its operand scales are per tensor per IMAGE (exact
powers of two), so a tile alone and the same tile inside a 16-tile batch must agree bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CLASS_THRESHOLDS = {0: (0.3, 0.7), 1: (0.3, 0.5)}
SMALL = {1}
THR = 0.3


@pytest.fixture(scope="module")
def env(gpu_device):
    from deepemia_amd import synth
    from deepemia_amd.engine import MaskRCNNEngine
    from deepemia_amd.functions.inference import InferencePipeline
    from deepemia_amd.predictor import Predictor
    from oracle import tile_parity as TP

    sd = synth.random_d2_state_dict(101, 2, seed=0)
    eng = MaskRCNNEngine(sd, 101, 2, THR, gpu_device, "f16x2")
    img = synth.em_tile(0, 2048)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = TP.reference_tile(img, sd, 101, THR, CLASS_THRESHOLDS, SMALL)
    return dict(sd=sd, eng=eng, img=img, ref=ref, TP=TP, synth=synth, dev=gpu_device,
                pipe=InferencePipeline([Predictor(eng)], "headline", {}, {}), Predictor=Predictor)


def test_r101_2048_f16x2_predictor_matches_oracle(env):
    raw = env["ref"]["raw"]
    inst = env["Predictor"](env["eng"])(env["img"])["instances"].to("cpu")
    n = raw["scores"].shape[0]
    assert len(inst) == n == 100
    np.testing.assert_array_equal(inst.pred_classes.numpy(), raw["pred_classes"].numpy())
    assert float((inst.scores - raw["scores"]).abs().max()) < 1e-4
    assert bool((inst.scores[:-1] >= inst.scores[1:]).all())
    m, r = inst.pred_masks, raw["pred_masks"]
    assert tuple(m.shape) == (n, 2048, 2048)
    iou = (m & r).sum((1, 2)).float() / (m | r).sum((1, 2)).float().clamp(min=1)
    assert float(iou.min()) >= 0.999, float(iou.min())
    # every pixel on which the two disagree is a TIE of the paste threshold: the CPU path's own sampled probability there
    # is within 1e-4 of 0.5 (scores / mask probabilities agree to ~1e-5; ~30 000 border pixels per tile)
    from oracle import maskrcnn_ref as R
    differing = (m != r).flatten(1).any(1).nonzero().flatten().tolist()
    worst = 0.0
    for i in differing:
        soft = R.paste_masks(raw["mask_probs28"][i:i + 1], raw["pred_boxes"][i:i + 1], 2048, 2048, soft=True)[0]
        worst = max(worst, float((soft[m[i] != r[i]] - 0.5).abs().max()))
    print("instances with differing pixels:", len(differing), "of", n, "; max |p - 0.5| on a differing pixel:", worst)
    assert worst <= 1e-4 and len(differing) <= n // 5


def test_r101_2048_f16x2_whole_tile_path_matches_oracle(env):
    pipe, dev, TP = env["pipe"], env["dev"], env["TP"]
    x = torch.from_numpy(env["img"])[None].to(dev)
    packed, scores, classes, recs = pipe.process_tile_batch("headline", x, SMALL, CLASS_THRESHOLDS)[0]
    assert packed is not None and packed.shape[0] > 10
    dense = pipe.ops.to_dense(packed, 2048)
    res = TP.compare_tile(env["ref"], dense, scores, classes, recs)
    print("headline parity:", {k: v for k, v in res.items()})
    assert res["ok"], res
    assert res["csv_rows"] >= res["instances"] > 10


def test_f16x2_forward_is_batch_invariant(env):
    """Scales are kept per IMAGE (one scale group per tile of the batch, ``demia_conv_p32_desc.groups``), the K order of a
    dot product does not depend on the tile shape, and every other stage works image by image: a tile alone and the same
    tile inside a 16-tile batch -- next to a much brighter tile -- give the SAME bits."""
    eng, synth, dev = env["eng"], env["synth"], env["dev"]
    tiles = np.stack([synth.em_tile(i, 2048) for i in range(16)])
    tiles[3] = (tiles[3].astype(np.int32) * 5 // 2).clip(0, 255).astype(np.uint8)        # one tile much brighter than the others
    x = torch.from_numpy(tiles).to(dev)
    full = eng.forward(x)
    full = type(full)(*[t.clone() if torch.is_tensor(t) else t for t in
                        (full.boxes, full.scores, full.classes, full.valid, full.count, full.packed, full.height, full.width, full.bbox)])
    for i in (0, 3, 15):
        one = eng.forward(x[i:i + 1].contiguous())
        n = int(one.count[0])
        assert n == int(full.count[i]) and n > 10
        assert torch.equal(one.classes[0, :n], full.classes[i, :n])
        assert torch.equal(one.scores[0, :n], full.scores[i, :n])
        assert torch.equal(one.boxes[0, :n], full.boxes[i, :n])
        assert torch.equal(one.packed[0, :n], full.packed[i, :n])


def test_plane_pools_give_the_same_tiles_as_fresh_planes(env):
    """``pooled_planes`` (bench.py's tile-batch loop): the class-pass, cross-class and final mask sets live in pools that stay
    zero outside per-slot boxes, so a gather writes boxes instead of planes.  Three consecutive batches of DIFFERENT tiles
    (different instance counts and boxes per slot, so every slot sees stale boxes of another instance) must give the bits the
    fresh-plane path gives."""
    pipe, dev, synth = env["pipe"], env["dev"], env["synth"]
    batches = [torch.from_numpy(np.stack([synth.em_tile(i, 2048) for i in idx])).to(dev) for idx in ((0, 1, 2), (5, 3), (4, 0, 6))]
    batches[1][1] = 0                                        # a tile with (next to) nothing on it: slots fall empty
    want = []
    for b, x in enumerate(batches):
        res = pipe.process_tile_batch(f"fresh{b}", x, SMALL, CLASS_THRESHOLDS, um_pix=0.5)
        want.append([(None if p is None else p.clone(), s, c, r) for p, s, c, r in res])
    pipe.pooled_planes = True
    try:
        for b, x in enumerate(batches):
            got = pipe.process_tile_batch(f"pool{b}", x, SMALL, CLASS_THRESHOLDS, um_pix=0.5)
            assert len(got) == len(want[b])
            for (pa, sa, ca, ra), (pb, sb, cb, rb) in zip(got, want[b]):
                assert (pa is None) == (pb is None)
                if pa is not None:
                    assert torch.equal(pa, pb)
                assert [float(v) for v in sa] == [float(v) for v in sb] and list(ca) == list(cb)
                for ia, ib in zip(ra or [], rb or []):
                    assert len(ia) == len(ib)
                    for u, v in zip(ia, ib):
                        assert np.array_equal(u["points"], v["points"]) and u["area"] == v["area"] and np.array_equal(u["values"], v["values"])
        assert any(p is not None and p.shape[0] > 10 for p, *_ in want[0])
    finally:
        pipe.pooled_planes = False
