"""Does every HIP stream reach the whole chip?  A 1 GB fill and a big GEMM-like elementwise kernel timed on each of 16 freshly created streams
(GPU_MAX_HW_QUEUES hardware queues), against the default stream."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "8")
import torch
x = torch.empty(1 << 28, device="cuda")          # 1 GB
y = torch.randn(1 << 26, device="cuda")
def t_on(s):
    with torch.cuda.stream(s):
        x.zero_(); torch.sin(y)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(); x.zero_(); e1.record()
        for _ in range(4): torch.sin(y)
        e2.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), e1.elapsed_time(e2) / 4
print("GPU_MAX_HW_QUEUES", os.environ["GPU_MAX_HW_QUEUES"])
print("default stream: fill %.3f ms, sin %.3f ms" % t_on(torch.cuda.current_stream()))
ss = []
for i in range(16):
    s = torch.cuda.Stream()
    ss.append(s)
    print(f"stream {i:2d} {s.cuda_stream:#x}: fill %.3f ms, sin %.3f ms" % t_on(s), flush=True)
