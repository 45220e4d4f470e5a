"""Host cost of looking up the current stream (the C ABI takes a raw stream handle): torch.cuda.current_stream().cuda_stream
against torch._C._cuda_getCurrentRawStream(device)."""
import time, torch
torch.cuda.init()
d = torch.cuda.current_device()
N = 20000
t = time.perf_counter()
for _ in range(N): a = torch.cuda.current_stream().cuda_stream
t1 = time.perf_counter() - t
t = time.perf_counter()
for _ in range(N): b = torch._C._cuda_getCurrentRawStream(d)
t2 = time.perf_counter() - t
t = time.perf_counter()
for _ in range(N): b = torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
t3 = time.perf_counter() - t
print("current_stream().cuda_stream %.2f us | _cuda_getCurrentRawStream(d) %.2f us | with current_device() %.2f us | equal %s" % (1e6 * t1 / N, 1e6 * t2 / N, 1e6 * t3 / N, a == b))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    print("on a side stream: equal", torch.cuda.current_stream().cuda_stream == torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()), s.cuda_stream != a)
