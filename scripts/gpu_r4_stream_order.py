"""Are kernels this library launches on the stream PyTorch reports as current (handle 0 = the default stream) ordered with PyTorch's own
kernels on that stream?  (a) library kernel first (a 200 ms clock probe), then a torch op + current_stream().synchronize(): ordered <=> the
wait takes ~200 ms.  (b) torch first (torch.cuda._sleep), then the library's probe: its first time stamp must lie behind the sleep."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pta_replicator_amd import _lib, device as dv
print("current stream handle:", torch.cuda.current_stream().cuda_stream, "default:", torch.cuda.default_stream().cuda_stream)
buf = torch.zeros((64, 2), dtype=torch.int64, device="cuda")
x = torch.zeros(1, device="cuda")
torch.cuda.synchronize()
for name, sp in (("NULL (what dv.stream_ptr() passes for handle 0)", None), ("an explicit torch.cuda.Stream", "side")):
    side = torch.cuda.Stream()
    ptr = None if sp is None else side.cuda_stream
    ctx = torch.cuda.stream(side) if sp else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        t0 = time.perf_counter()
        _lib.call("pta_clock_probe", buf.data_ptr(), 64, 200000, 10000, ptr)
        x += 1
        torch.cuda.current_stream().synchronize()
        dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"(a) library kernel on {name}, then a torch op on the current stream: current_stream().synchronize() returned after {dt * 1e3:.1f} ms "
          f"({'ORDERED' if dt > 0.15 else 'NOT ORDERED'})")
# (b) torch first
torch.cuda.synchronize()
t0 = time.perf_counter()
torch.cuda._sleep(int(2.4e9 * 0.2))          # ~200 ms of spinning on the current stream
_lib.call("pta_clock_probe", buf.data_ptr(), 64, 1000, 100, None)
torch.cuda.synchronize()
print(f"(b) torch._sleep(200 ms) then a 1 ms library probe on NULL: total {1e3 * (time.perf_counter() - t0):.1f} ms")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
ev0.record(); torch.cuda._sleep(int(2.4e9 * 0.2)); _lib.call("pta_clock_probe", buf.data_ptr(), 64, 50000, 1000, None); ev1.record(); torch.cuda.synchronize()
print(f"    events around sleep(200 ms) + probe(50 ms): {ev0.elapsed_time(ev1):.1f} ms (ordered <=> ~250; overlapped <=> ~200)")
