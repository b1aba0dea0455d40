import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print({k: v for k, v in os.environ.items() if any(s in k for s in ("HIP", "AMD", "GPU", "HSA", "ROC"))}, flush=True)
dev = torch.device("cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def sleep1():
    with torch.cuda.stream(s1): torch.cuda._sleep(200_000_000)
def sleep2():
    with torch.cuda.stream(s1): torch.cuda._sleep(200_000_000)
    with torch.cuda.stream(s2): torch.cuda._sleep(200_000_000)
print(f"_sleep: one stream {t(sleep1):.1f} ms, two streams {t(sleep2):.1f} ms", flush=True)
a = torch.randn(512, 4096, device=dev).half(); b = torch.randn(4096, 512, device=dev).half()
c1 = torch.empty(512, 512, device=dev, dtype=torch.half); c2 = torch.empty_like(c1)
def mm1():
    with torch.cuda.stream(s1):
        for _ in range(200): torch.mm(a, b, out=c1)
def mm2():
    for _ in range(200):
        with torch.cuda.stream(s1): torch.mm(a, b, out=c1)
        with torch.cuda.stream(s2): torch.mm(a, b, out=c2)
print(f"torch.mm 512x512x4096 x200: one stream {t(mm1):.2f} ms, two streams {t(mm2):.2f} ms", flush=True)
x = torch.randn(64, 1 << 20, device=dev)
def red1():
    with torch.cuda.stream(s1):
        for _ in range(50): x[:8].sum(dim=1)
def red2():
    for _ in range(50):
        with torch.cuda.stream(s1): x[:8].sum(dim=1)
        with torch.cuda.stream(s2): x[8:16].sum(dim=1)
print(f"row sums (8 rows x 1M) x50: one stream {t(red1):.2f} ms, two streams {t(red2):.2f} ms", flush=True)
