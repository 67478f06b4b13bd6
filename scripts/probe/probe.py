"""Read-only HBM streaming ceiling of this box (GB/s): scripts/probe/stream_probe.hip over a 12 GiB buffer."""
import ctypes, os, subprocess, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libprobe.so")
if not os.path.exists(so):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "--offload-arch=gfx950", os.path.join(here, "stream_probe.hip"), "-o", so], check=True)
L = ctypes.CDLL(so)
L.probe_read.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
nbytes = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 12 << 30
buf = torch.empty(nbytes // 4, dtype=torch.int32, device="cuda").fill_(1)
sink = torch.zeros(4, dtype=torch.int32, device="cuda")
best = 0
for grid in (1024, 2048, 4096, 8192):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        L.probe_read(buf.data_ptr(), nbytes, sink.data_ptr(), grid, s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        L.probe_read(buf.data_ptr(), nbytes, sink.data_ptr(), grid, s)
    b.record(); torch.cuda.synchronize()
    gbps = nbytes * 5 / (a.elapsed_time(b) * 1e-3) / 1e9
    best = max(best, gbps)
    print(f"grid {grid}: {gbps:.0f} GB/s")
print(f"stream_read_ceiling_GBps {best:.0f}")
