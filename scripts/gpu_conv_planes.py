"""Dev script: the split conv kernel fed with activations ALREADY split into three bf16 planes (no split arithmetic in
the K loop) against the f32-input kernel, on the layers that carry the time.  Bounds what a planes-in-HBM activation
layout could gain."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepemia_amd import synth, engine as E, _lib
from deepemia_amd._lib import F32X3, F32

sd = synth.random_d2_state_dict(101, 2, 0)
eng = E.MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f32x3')
lib = eng.lib


def run(x, L, dtype, xin):
    n, h, w, cin = x.shape
    ho = (h + 2 * L.pad - L.kh) // L.stride + 1; wo = (w + 2 * L.pad - L.kw) // L.stride + 1
    out = torch.empty((n, ho, wo, L.cout), dtype=torch.float32, device=x.device)
    d = _lib.ConvDesc(_lib.ptr(xin), _lib.ptr(L.w3), _lib.ptr(L.scale), _lib.ptr(L.bias), 0, _lib.ptr(out),
                      n, h, w, cin, ho, wo, L.cout, L.cout_pad, L.kh, L.kw, L.stride, L.pad, dtype, F32, 1, 0, L.cout, 0)
    st = int(torch.cuda.current_stream().cuda_stream)
    rc = lib.demia_conv2d_nhwc(ctypes.byref(d), st)
    assert rc == 0, rc
    return out


def bench(name, x, L):
    planes = E.split3_bf16(x).contiguous()          # [3, n, h, w, c]
    flops = 2.0 * x.shape[0] * ((x.shape[1] + 2 * L.pad - L.kh) // L.stride + 1) * ((x.shape[2] + 2 * L.pad - L.kw) // L.stride + 1) * L.cout * L.kh * L.kw * L.cin
    res = {}
    for tag, dt, xin in (("f32 in", F32X3, x), ("planes in", 4, planes)):
        for _ in range(3):
            y = run(x, L, dt, xin)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            y = run(x, L, dt, xin)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res[tag] = (ms, y)
    d = (res["f32 in"][1] - res["planes in"][1]).abs().max().item()
    print(f"{name:34s} f32 in {res['f32 in'][0]:7.3f} ms {flops / res['f32 in'][0] / 1e9:6.1f} TF/s | planes in "
          f"{res['planes in'][0]:7.3f} ms {flops / res['planes in'][0] / 1e9:6.1f} TF/s | max diff {d:.2e}")


g = torch.Generator(device='cuda').manual_seed(0)
def rnd(*s): return torch.randn(*s, device='cuda', generator=g)

B = 16
bench("p2 3x3 256->256 M=640000", rnd(B, 200, 200, 256), eng.fpn_output[2])
bench("mask 3x3 256->256 M=313600", rnd(B * 100, 14, 14, 256), eng.mask_fcn[0])
blk = eng.blocks[2][3]
bench("res4 3x3 256->256 M=40000", rnd(B, 50, 50, 256), blk["conv2"])
bench("res4 1x1 256->1024 M=40000", rnd(B, 50, 50, 256), blk["conv3"])
bench("res4 1x1 1024->256 M=40000", rnd(B, 50, 50, 1024), blk["conv1"])
bench("fc1 12544->1024 M=16000", rnd(B * 1000, 1, 1, 12544), eng.fc1)
blk = eng.blocks[0][1]
bench("res2 1x1 64->256 M=640000", rnd(B, 200, 200, 64), blk["conv3"])
bench("res2 3x3 64->64 M=640000", rnd(B, 200, 200, 64), blk["conv2"])
