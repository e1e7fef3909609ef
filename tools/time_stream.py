"""Streaming reference points on this GPU (ud_bench_stream: read, copy, block-contiguous nontemporal read of a 484 MB tensor, cache scrubbed): what a pure streaming kernel reaches, next to which the gather kernels are judged."""
import os as _os; _os.environ.setdefault("UD_RANDOM_INIT", "1")   # synthetic weights (tools never train for real)
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cvpr2023-unidistill_amd")]
import torch
from unidistill_amd import _lib
lib = _lib.load(); d = torch.device("cuda:0")
n = 484 * 1024 * 1024 // 4
src = torch.randn(n, device=d); dst = torch.empty(n, device=d)
scrub = torch.empty(512 << 20, dtype=torch.uint8, device=d)
for mode, name, nbytes in ((0, "bench.stream_read", n * 4), (1, "bench.stream_copy", n * 8), (2048, "bench.stream_read_blk", n * 4), (8192, "bench.stream_read_blk", n * 4), (32768, "bench.stream_read_blk", n * 4)):
    for _ in range(2): lib.ud_bench_stream(_lib.ptr(src), _lib.ptr(dst), n, mode, _lib.stream_of(src))
    torch.cuda.synchronize(); _lib.prof_enable(True)
    for _ in range(8):
        scrub.zero_(); lib.ud_bench_stream(_lib.ptr(src), _lib.ptr(dst), n, mode, _lib.stream_of(src))
    torch.cuda.synchronize(); _lib.prof_enable(False)
    ms, c = _lib.prof_read(name)
    print(f"{name} mode={mode}: {ms/c*1e3:.1f} us for {nbytes/1e6:.0f} MB -> {nbytes/(ms/c)/1e9:.2f} TB/s (cache scrubbed)")
