"""Does the NUMBER of concurrent streams cost bandwidth?  K arrays of 10 GB / K read in lockstep (stream_probe.hip)."""
import ctypes, os, subprocess
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libprobe.so")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "--offload-arch=gfx950", os.path.join(here, "stream_probe.hip"), "-o", so], check=True)
L = ctypes.CDLL(so)
L.probe_read_k.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
total = 10 * (1 << 30)
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for K in (1, 2, 5):
    each = total // K // 4096 * 4096
    bufs = [torch.empty(each // 4, dtype=torch.int32, device="cuda").fill_(1) for _ in range(K)]
    ptrs = torch.tensor([b.data_ptr() for b in bufs], dtype=torch.int64, device="cuda")
    best = 0
    for grid in (2048, 4096, 8192):
        for _ in range(2):
            L.probe_read_k(ptrs.data_ptr(), K, each, sink.data_ptr(), grid, s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            L.probe_read_k(ptrs.data_ptr(), K, each, sink.data_ptr(), grid, s)
        b.record(); torch.cuda.synchronize()
        best = max(best, each * K * 5 / (a.elapsed_time(b) * 1e-3) / 1e9)
    print(f"K={K}: {best:.0f} GB/s")
    del bufs
