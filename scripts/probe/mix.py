"""Memory-system ceiling of the stream kernel's C3 line mix (scripts/probe/mix_probe.hip): ms per 10 M-query pass for each
dependency shape and occupancy.  Usage (GPU box): python scripts/probe/mix.py"""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libmix.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, "mix_probe.hip")):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "--offload-arch=gfx950", os.path.join(here, "mix_probe.hip"), "-o", so], check=True)
L = ctypes.CDLL(so)
L.mix_run.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
n_q, n_rows = 10_000_000, 2_400_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
bits = torch.randint(0, 1 << 20, (n_q * 50,), dtype=torch.int32, device="cuda", generator=g)
side = torch.randint(0, 1 << 20, (n_q * 50 * 4,), dtype=torch.int32, device="cuda", generator=g)
rows = torch.randint(0, 1 << 20, (n_rows * 32,), dtype=torch.int32, device="cuda", generator=g)
out = torch.zeros(n_q * 8, dtype=torch.int32, device="cuda")
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
cus = torch.cuda.get_device_properties(0).multi_processor_count
lines = 15.6e6 + 11.7e6 + 10e6          # bit-score, side-record and row lines of the product kernel on C3 (profiles/)
for mode, waves in ((0, 8), (0, 12), (0, 16), (1, 8), (1, 12), (1, 16), (2, 12)):
    args = [bits.data_ptr(), side.data_ptr(), rows.data_ptr(), out.data_ptr(), n_q, n_rows, mode, sink.data_ptr(), waves, cus, s]
    for _ in range(2):
        assert L.mix_run(*args) == 0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        L.mix_run(*args)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"mode {mode} ({('independent', 'chain', 'chain, two tasks in flight')[mode]}) {waves} waves/CU: {ms:.3f} ms  "
          f"(~{(lines * 128 + n_q * 32) / ms / 1e6:.0f} GB/s of lines)", flush=True)
