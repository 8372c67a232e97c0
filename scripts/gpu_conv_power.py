"""Dev script: shader clock and board power while the backbone convs run back to back (sysfs hwmon / pp_dpm_sclk,
read from a sampling thread), for random data and for zero data.  Answers whether the conv's ceiling is the clock
the chip holds under this load rather than the kernel's structure."""
import sys, os, glob, time, threading, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepemia_amd import synth, engine as E


def find_nodes():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(dev + "/hwmon/hwmon*")
        if not hw:
            continue
        out.append((dev, hw[0]))
    return out


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def sample(dev, hw):
    p = read(hw + "/power1_average") or read(hw + "/power1_input")
    f = read(hw + "/freq1_input")
    dpm = read(dev + "/pp_dpm_sclk")
    cur = None
    if dpm:
        for line in dpm.splitlines():
            if line.endswith("*"):
                cur = line
    return (float(p) / 1e6 if p else None, float(f) / 1e6 if f else None, cur)


def run(zero, seconds=4.0):
    sd = synth.random_d2_state_dict(101, 2, 0)
    if zero:
        sd = {k: (v * 0 if v.dtype.is_floating_point and "running_var" not in k else v) for k, v in sd.items()}
    eng = E.MaskRCNNEngine(sd, 101, 2, 0.3, 'cuda:0', 'f32x3')
    x = torch.from_numpy(np.stack([synth.em_tile(i, 2048) for i in range(8)])).cuda()
    xin, newh, neww, ph, pw = eng.preprocess(x)
    if zero:
        xin = xin * 0
    for _ in range(2):
        eng.backbone(xin, ph, pw)
    torch.cuda.synchronize()
    nodes = find_nodes()
    samples = {i: [] for i in range(len(nodes))}
    stop = False

    def sampler():
        while not stop:
            for i, (dev, hw) in enumerate(nodes):
                samples[i].append(sample(dev, hw))
            time.sleep(0.05)
    th = threading.Thread(target=sampler); th.start()
    eng.conv_events = []
    t0 = time.time(); it = 0
    while time.time() - t0 < seconds:
        eng.backbone(xin, ph, pw); it += 1
        torch.cuda.synchronize()
    stop = True; th.join()
    ev = eng.conv_events
    tot_t = sum(e[0].elapsed_time(e[1]) for e in ev); tot_f = sum(e[2] for e in ev)
    print(("ZERO data" if zero else "random data"), f"{it} backbone passes, conv {tot_f / tot_t / 1e9:.1f} TF/s")
    for i, (dev, hw) in enumerate(nodes):
        s = samples[i][len(samples[i]) // 4:]          # skip the ramp
        pw_ = [a[0] for a in s if a[0] is not None]; fr = [a[1] for a in s if a[1] is not None]
        print(f"  {dev}: power mean {np.mean(pw_) if pw_ else float('nan'):.0f} W max {max(pw_) if pw_ else float('nan'):.0f} W;"
              f" sclk mean {np.mean(fr) if fr else float('nan'):.0f} MHz min {min(fr) if fr else float('nan'):.0f}"
              f" max {max(fr) if fr else float('nan'):.0f}; dpm {s[-1][2] if s else None}; cap {read(hw + '/power1_cap')}")


if __name__ == "__main__":
    print("idle:", [sample(d, h) for d, h in find_nodes()])
    run(False)
    run(True)
