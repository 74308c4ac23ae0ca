"""Cross-stream event hand-off cost between the default stream and each of 16 fresh streams: 200 round trips
(main kernel -> event -> side waits -> side kernel -> event -> main waits), per round trip; plus how two 0.3 ms kernels overlap."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "8")
import torch
main = torch.cuda.current_stream()
a = torch.zeros(1024, device="cuda"); b = torch.zeros(1024, device="cuda")
big1 = torch.randn(1 << 25, device="cuda"); big2 = torch.randn(1 << 25, device="cuda")
def pingpong(s, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        a.add_(1.0)
        e = torch.cuda.Event(); e.record(main); s.wait_event(e)
        with torch.cuda.stream(s):
            b.add_(1.0)
            e2 = torch.cuda.Event(); e2.record(s)
        main.wait_event(e2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
def fork_join(s, n=50):
    """main: K1 (0.1 ms); fork; side: K2 (0.1 ms) || main: K3 (0.1 ms); join -> ideal 0.2 ms per iteration, serial 0.3"""
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        torch.sin(big1)
        e = torch.cuda.Event(); e.record(main); s.wait_event(e)
        with torch.cuda.stream(s):
            torch.sin(big2)
            e2 = torch.cuda.Event(); e2.record(s)
        torch.sin(big1)
        main.wait_event(e2)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("GPU_MAX_HW_QUEUES", os.environ["GPU_MAX_HW_QUEUES"])
ss = []
for i in range(16):
    s = torch.cuda.Stream(); ss.append(s)
    with torch.cuda.stream(s): b.add_(1.0)
    torch.cuda.synchronize()
    pingpong(s, 20)
    print(f"stream {i:2d}: ping-pong {pingpong(s):7.1f} us per round trip; fork-join {fork_join(s):.3f} ms", flush=True)
