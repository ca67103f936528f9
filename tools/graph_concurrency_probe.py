"""Does a replayed multi-stream hipGraph run its forked branches concurrently on this ROCm?  (GPU only.)  Two independent
chains of small-grid kernels (each leaves most CUs idle) on two streams: eager two-stream time vs one-stream time vs the same
two-stream work captured in ONE graph."""
import time

import torch

dev = torch.device("cuda:0")
a = torch.randn(64, 1 << 14, device=dev)
b = torch.randn(64, 1 << 14, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 200


def chain(x):
    for _ in range(N):
        x = torch.cumsum(x, 1) * 0.5        # a long, narrow kernel: 64 rows -> few workgroups


def two_streams():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        chain(b)
    chain(a)
    cur.wait_stream(s2)


def one_stream():
    chain(b)
    chain(a)


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.cuda.stream(s1):
    print(f"eager one stream   {timeit(one_stream):8.2f} ms")
    print(f"eager two streams  {timeit(two_streams):8.2f} ms")
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s1):
        two_streams()
    print(f"graph two streams  {timeit(g.replay):8.2f} ms")
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=s1):
        one_stream()
    print(f"graph one stream   {timeit(g1.replay):8.2f} ms")
