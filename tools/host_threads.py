"""Where does the host time of a training step go?  Per-THREAD CPU time (main thread = forward + optimizer, autograd engine
thread = backward, the rest = HIP runtime helpers) per step, with GPU waits blocking instead of spinning, next to the wall
time; then a cProfile of the main thread.  AC=bf16 for the mixed-precision step, WL=<workload>."""
import cProfile, ctypes, os, pstats, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
os.environ.setdefault("UD_RANDOM_INIT", "1")
import torch
from unidistill_amd import train as T

TICK = os.sysconf("SC_CLK_TCK")


def thread_cpu():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{tid}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / TICK)
        except Exception:
            pass
    return out


dev = torch.device("cuda:0")
path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
hip = ctypes.CDLL(path)
_rc = hip.hipSetDeviceFlags(ctypes.c_uint(4))
if "--json" not in sys.argv:
    print("blocking sync flag rc:", _rc)
torch.manual_seed(0)
AC = torch.bfloat16 if os.environ.get("AC") == "bf16" else None
tr = T.Trainer(T.DistillStep(os.environ.get("WL", "camera_exp_distill_lidar")), device=dev, autocast_dtype=AC, channels_last=True)
batch = T.synthetic_batch(dev, int(os.environ.get("B", 4)))
for _ in range(5):
    tr.step(batch)
torch.cuda.synchronize()
N = int(os.environ.get("STEPS", 20 if "--json" in sys.argv else 10))
c0, w0, p0 = thread_cpu(), time.perf_counter(), time.process_time()
for _ in range(N):
    tr.step(batch)
p1 = time.process_time()
torch.cuda.synchronize()
w1, c1 = time.perf_counter(), thread_cpu()
if "--json" in sys.argv:        # bench.py's host_enqueue leg: one JSON line, no profile
    import json
    main_tid = threading.get_native_id()
    ms = {"main": 0.0, "autograd": 0.0, "runtime_and_other": 0.0}
    for tid, (name, t) in c1.items():
        d = (t - c0.get(tid, (name, 0.0))[1]) / N * 1e3
        ms["main" if tid == main_tid else "autograd" if name.startswith("pt_autograd") else "runtime_and_other"] += d
    print(json.dumps({"host_enqueue_ms": ms["main"] + ms["autograd"], "main_thread_ms": ms["main"],
                      "autograd_thread_ms": ms["autograd"], "runtime_and_other_threads_ms": ms["runtime_and_other"],
                      "wall_ms": 1e3 * (w1 - w0) / N, "steps": N, "sync_mode": "blocking (hipDeviceScheduleBlockingSync set before the first launch)"}))
    sys.exit(0)
print(f"wall {1e3 * (w1 - w0) / N:.2f} ms/step, process CPU {1e3 * (p1 - p0) / N:.2f} ms/step; main tid {threading.get_native_id()}")
for tid, (name, t) in sorted(c1.items(), key=lambda kv: -(kv[1][1] - c0.get(kv[0], ("", 0))[1])):
    d = t - c0.get(tid, ("", 0.0))[1]
    if d > 0.0005 * N:
        print(f"  tid {tid} {name:20s} {1e3 * d / N:8.2f} ms/step")
# forward-only and backward-only host time (main thread wall, GPU drained first so that nothing blocks)
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tr.step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:int(os.environ.get("TOP", 30))]
print("main thread, self time per step:")
for (fn, line, name), (cc, nc, tt, ct, _callers) in rows:
    short = fn.split("unidistill_amd/")[-1] if "unidistill_amd/" in fn else os.path.basename(fn)
    print(f"{tt / N * 1e3:8.3f} ms self {ct / N * 1e3:8.3f} ms cum {nc / N:8.1f} calls  {short}:{line} {name}")
