"""Shader clock / power under (a) the bf16 conv3x3 tap kernel looped, (b) idle: rocm-smi sampled from a side thread."""
import os, sys, subprocess, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault("UD_RANDOM_INIT", "1")
import torch
from unidistill_amd.ops import conv2d as c2
dev = torch.device("cuda:0")

def smi():
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in r.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l) and "GPU[0]" in l]
    return " | ".join(k.split(":", 1)[1].strip() if ":" in k else k for k in keep)

def run(name, fn, secs=8):
    stop = [False]; out = []
    def sampler():
        time.sleep(2)
        while not stop[0]:
            out.append(smi()); time.sleep(1.5)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(200): fn()
        torch.cuda.synchronize(); n += 200
    dt = time.time() - t0
    stop[0] = True; th.join()
    print(f"== {name}: {dt / n * 1e6:.1f} us per call"); [print("   ", o) for o in out[:4]]

print("idle:", smi())
for name, N, Cin, H, W, Cout in [("trunk 128->128 @180", 4, 128, 180, 180, 128), ("fusion 512->256 @180", 4, 512, 180, 180, 256)]:
    x = torch.randn(N, Cin, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = c2.tap_major(torch.randn(Cout, Cin, 3, 3, device=dev) * 0.02)
    run(name, lambda: c2._launch(x, wt, Cout))
    xz = torch.zeros_like(x); wz = torch.zeros_like(wt)
    run(name + " (all-zero operands)", lambda: c2._launch(xz, wz, Cout))
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
run("hipBLASLt 8192^3 bf16 GEMM", lambda: a @ a)
a32 = torch.randn(8192, 8192, device=dev)
run("library 8192^3 fp32 GEMM", lambda: a32 @ a32, secs=6)
from unidistill_amd.ops import conv2d_f32 as f32
x = torch.randn(4, 128, 180, 180, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(128, 128, 3, 3, device=dev) * 0.02
run("fp32 Winograd 128->128 @180", lambda: f32._launch3(x, w, None), secs=6)
x1 = torch.randn(24, 1024, 16, 44, device=dev).contiguous(memory_format=torch.channels_last)
w1 = (torch.randn(256, 1024, device=dev) * 0.02)
run("fp32 1x1 1024->256 @16x44x24", lambda: f32._launch1(x1, w1, 256, None), secs=6)
