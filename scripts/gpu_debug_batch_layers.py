"""Dev script (GPU): a tile alone vs the same tile inside a batch, conv layer by conv layer -- where do the two first differ,
and by how much?  usage: gpu_debug_batch_layers.py [batch] [tile index]"""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from deepemia_amd import synth, engine as E, p32

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
I = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sd = synth.random_d2_state_dict(101, 2, 0)
eng = E.MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f16x2')
tiles = np.stack([synth.em_tile(i, 2048) for i in range(B)])
tiles[3 % B] = (tiles[3 % B].astype(np.int32) * 5 // 2).clip(0, 255).astype(np.uint8)
x = torch.from_numpy(tiles).cuda()

rec = []
orig = eng.conv_p32
state = {"b": B, "i": I}


def wrapped(xx, L, *a, **k):
    out = orig(xx, L, *a, **k)
    b, i = state["b"], state["i"]
    if isinstance(out, p32.P32):
        d = eng.dense(out)
        meta = out.meta.tolist()
    else:
        d, meta = out, None
    n = d.shape[0]
    per = n // b
    rec.append((f"{L.cin}->{L.cout} k{L.kh} s{L.stride} n={n}", d[i * per:(i + 1) * per].float().cpu().clone(), meta, xx.meta.tolist()))
    return out


eng.conv_p32 = wrapped
full = eng.forward(x, keep_intermediates=True)
torch.cuda.synchronize()
rec_full, rec = rec, []
fd = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in full.dbg.items() if k in ("props", "pcount", "logits", "det_boxes", "mask_prob", "xin")}
stem_full = eng.dense(full.dbg["feats"]["stem"])[I].clone()
state.update(b=1, i=0)
one = eng.forward(x[I:I + 1].contiguous(), keep_intermediates=True)
torch.cuda.synchronize()
print('stem input differ', float((fd["xin"][I] - one.dbg["xin"][0]).abs().max()), 'stem+pool output differ',
      float((stem_full - eng.dense(one.dbg["feats"]["stem"])[0]).abs().max()))
first = True
shown = 0
for (na, a, ma, ia), (nb, b, mb, ib) in zip(rec_full, rec):
    d = float((a - b).abs().max())
    if d > 0 or first:
        print(f"{na:34s} max|diff| {d:.3g}  max|x| {float(b.abs().max()):.3g}  differing {int((a != b).sum())}/{a.numel()}  "
              f"out meta batch {ma} single {mb}  in meta batch {ia} single {ib}")
        first = False
        shown += 1
        if shown > 12:
            break
print('layers compared', len(rec), 'layers that differ', sum(float((a[1] - b[1]).abs().max()) > 0 for a, b in zip(rec_full, rec)))
R = fd["props"].shape[1]
print('proposals differ', float((fd["props"][I] - one.dbg["props"][0]).abs().max()), 'count', int(fd["pcount"][I]), int(one.dbg["pcount"][0]))
print('box logits differ', float((fd["logits"][I] - one.dbg["logits"][0]).abs().max()))
print('det boxes differ', float((fd["det_boxes"][I] - one.dbg["det_boxes"][0]).abs().max()))
mp_f, mp_o = fd["mask_prob"].view(B, -1), one.dbg["mask_prob"].view(1, -1)
print('mask prob differ', float((mp_f[I] - mp_o[0]).abs().max()))
n = int(one.count[0])
diff = eng.unpack(one.packed[0, :n].contiguous(), 2048, 2048) != eng.unpack(full.packed[I, :n].contiguous(), 2048, 2048)
print('mask pixels differing per instance', diff.sum((1, 2)).tolist())
print('scores differ', float((one.scores[0, :n] - full.scores[I, :n]).abs().max()))
