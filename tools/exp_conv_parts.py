import ctypes, sys, os, torch
dev = torch.device("cuda:0")
def run(path, N, Cin, H, W, Cout):
    lib = ctypes.CDLL(path)
    f = lib.ud_conv3x3_nhwc_bf16
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p]*3 + [ctypes.c_int]*5 + [ctypes.c_void_p]*4 + [ctypes.c_int, ctypes.c_void_p]
    x = torch.randn(N, H, W, Cin, device=dev).bfloat16(); w = (torch.randn(Cout, 9, Cin, device=dev)*0.02).bfloat16()
    y = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    def go(): f(x.data_ptr(), w.data_ptr(), y.data_ptr(), N, H, W, Cin, Cout, None, None, None, None, 0, st)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for shape in [(4, 128, 180, 180, 128), (4, 256, 90, 90, 256), (4, 64, 180, 180, 2688)]:
    print(shape, {v: round(run(f"tools/_exp/libexp_{v}.so", *shape), 1) for v in ("BASE", "NO_COMMIT_B", "NO_FETCH_B", "NO_MFMA")})
