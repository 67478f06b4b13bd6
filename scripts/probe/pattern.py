"""HBM rate of the streaming kernel's access pattern alone (scripts/probe/pattern_probe.hip)."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libpattern.so")
L = ctypes.CDLL(so)
L.pattern_run.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
seg, n_seg = 50, 10_000_000
H = seg * n_seg
cols = [torch.ones(H, dtype=torch.int32, device="cuda") for _ in range(2)] + [torch.ones(H, dtype=torch.float64, device="cuda")] + [torch.ones(H, dtype=torch.int32, device="cuda") for _ in range(2)]
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
nbytes = 24 * H
s = torch.cuda.current_stream().cuda_stream
import itertools
outp = torch.zeros(n_seg * 8 + (1 << 24), dtype=torch.int32, device='cuda')
for (W, U), lds, sm in itertools.product(((4, 1), (5, 1), (5, 2), (5, 4)), (0,), (0, 18)):
    for grid in (2048,):
        args = [c.data_ptr() for c in cols] + [n_seg, seg, sink.data_ptr(), grid, U, W, s, lds, outp.data_ptr(), sm]
        for _ in range(2):
            L.pattern_run(*args)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            L.pattern_run(*args)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print(f"W={W} U={U} lds={lds} store_mode={sm} grid={grid}: {ms:.3f} ms  {nbytes / ms / 1e6:.0f} GB/s", flush=True)
