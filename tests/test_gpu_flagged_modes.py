"""the two flagged NON-parity modes named in north_star -- a soft-NMS merge (does not exist in the reference,
SURVEY N1) and adaptive multi-scale inference (dead code in the reference, ``inference.py:1833-2064``).  They are
opt-in and claim no parity with the reference's outputs; what is checked here is that the device versions do exactly what
their dense restatements in ``oracle/`` do on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _blobs(n, size, seed):
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    masks = np.zeros((n, size, size), dtype=bool)
    for i in range(n):
        cx, cy = g.uniform(40, size - 40, 2)
        if i % 3:                                   # two of three sit close to an earlier one: overlapping families
            j = g.integers(0, i)
            py, px = np.nonzero(masks[j])
            cx, cy = px.mean() + g.uniform(-14, 14), py.mean() + g.uniform(-14, 14)
        a, b, th = g.uniform(10, 34), g.uniform(8, 26), g.uniform(0, np.pi)
        u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th)
        v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
        masks[i] = (u / a) ** 2 + (v / b) ** 2 <= 1.0
    return masks


@pytest.fixture(scope="module")
def pipe(gpu_device):
    from deepemia_amd import synth
    from deepemia_amd.engine import MaskRCNNEngine
    from deepemia_amd.functions.inference import InferencePipeline
    from deepemia_amd.predictor import Predictor

    sd = synth.random_d2_state_dict(50, 2, seed=0, mask_bias=0.5, mask_gain=6.0)
    eng = MaskRCNNEngine(sd, 50, 2, 0.3, gpu_device, "f32")
    return InferencePipeline([Predictor(eng)], "flagged", {}, {}), sd, synth


def test_soft_nms_merge_equals_the_dense_restatement(pipe):
    from oracle import postproc_ref as P

    p, _, _ = pipe
    masks = _blobs(40, 320, 7)
    g = np.random.default_rng(8)
    scores = np.round(g.uniform(0.05, 0.99, 40), 3)
    scores[5] = scores[11]                                       # a tie: lower index first
    classes = (np.arange(40) % 2).tolist()
    p.ops.set_frame_width(320)
    packed = p.ops.from_dense(masks)
    for sigma, thr in ((0.5, 0.001), (0.1, 0.2), (2.0, 0.3)):
        km, ks, kc = p.soft_nms_merge(packed, scores.tolist(), classes, sigma, thr)
        order, rs = P.soft_nms_masks(list(masks), scores.tolist(), classes, sigma, thr)
        assert len(ks) == len(order) > 5 and kc == [classes[i] for i in order]
        np.testing.assert_array_equal(p.ops.to_dense(km, 320), masks[order])
        np.testing.assert_allclose(ks, rs, rtol=1e-12, atol=0)
    # a tiny sigma with a threshold above zero is hard NMS at any positive IoU; a huge sigma keeps everything undecayed
    km, ks, kc = p.soft_nms_merge(packed, scores.tolist(), classes, 1e-9, 0.04)
    dm = p.ops.to_dense(km, 320)
    assert 5 < len(ks) < 40
    assert all(not (dm[i] & dm[j]).any() for i in range(len(ks)) for j in range(i) if kc[i] == kc[j])
    km, ks, _ = p.soft_nms_merge(packed, scores.tolist(), classes, 1e12, 0.001)
    assert len(ks) == 40
    np.testing.assert_allclose(sorted(ks), sorted(scores.tolist()), rtol=1e-9)
    assert p.soft_nms_merge(None, [], []) == (None, [], [])


def test_adaptive_multiscale_equals_per_scale_oracle_composition(pipe, gpu_device):
    """Every scale the product visits, redone on the CPU: INTER_LINEAR resize (oracle/pipeline_ref.py), the Detectron2
    restatement, the single-model class pass, the scaled minimum size, INTER_NEAREST back, then the 0.4 cross-scale dedup."""
    from oracle import maskrcnn_ref, pipeline_ref as PR, postproc_ref as P

    p, sd, synth = pipe
    img = synth.em_tile(41, 384)
    small = {1}
    conf, iou_thr, cls = 0.3, 0.6, 0
    km, ks, kc = p.run_adaptive_multiscale_inference(0, "ms", torch.from_numpy(img).to(gpu_device), cls, conf, small, iou_thr)
    h, w = img.shape[:2]
    base_min = max(25, int(h * w * 0.0001))

    def one(scale):
        sh, sw = (h, w) if scale == 1.0 else (int(h * scale), int(w * scale))
        im = img if scale == 1.0 else PR.cv_resize_linear_u8(img, sh, sw)
        out = maskrcnn_ref.predict(im, sd, 50, 0.3)
        m, s, _ = P.single_model_class_pass(out["pred_masks"].numpy(), out["scores"].numpy(), out["pred_classes"].numpy(), (sh, sw),
                                            cls, small, conf, iou_thr, None, True)
        keep = [i for i in range(len(m)) if int(np.asarray(m[i]).sum()) >= int(base_min * scale ** 2)]
        return [P.resize_nearest(np.asarray(m[i]) > 0, h, w) if scale != 1.0 else np.asarray(m[i]) > 0 for i in keep], [float(s[i]) for i in keep]

    per, order = {}, [0.7, 1.0, 1.5]
    for s in order:
        per[s] = one(s)
    base = len(per[1.0][1])
    for unlocked, extra in ((len(per[1.5][1]) > base * 0.1, (2.0, 2.5)), (len(per[0.7][1]) > base * 0.1, (0.5, 0.6))):
        if unlocked:
            for s in extra:
                r = one(s)
                if len(r[1]) < base * 0.05:
                    break
                per[s] = r
                order.append(s)
    rm, rs = P.multiscale_merge(per, order)
    assert len(ks) == len(rs) > 3 and kc == [cls] * len(ks)
    np.testing.assert_allclose(ks, rs, rtol=0, atol=2e-5)
    dm = p.ops.to_dense(km, w)
    iou = [float((dm[i] & rm[i]).sum()) / max(float((dm[i] | rm[i]).sum()), 1.0) for i in range(len(rs))]
    assert min(iou) >= 0.995, iou                      # f32 engine vs CPU fp32 at five image scales: threshold-tie pixels only
    assert len(order) > 3                              # the adaptive phases ran
