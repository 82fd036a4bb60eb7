"""DEV: what the host link of this box gives: page-locked host -> device copies (one stream, two streams), and back."""
import time, torch
dev = torch.device("cuda:0")
n = 1 << 28  # 256 MiB
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
d = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
s = [torch.cuda.Stream() for _ in range(2)]
def run(label, fn, nbytes, reps=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {nbytes * reps / dt / 1e9:.1f} GB/s", flush=True)
def one():
    with torch.cuda.stream(s[0]): d[0].copy_(h[0], non_blocking=True)
def two():
    for k in range(2):
        with torch.cuda.stream(s[k]): d[k].copy_(h[k], non_blocking=True)
def back():
    with torch.cuda.stream(s[0]): h[0].copy_(d[0], non_blocking=True)
def both():
    with torch.cuda.stream(s[0]): d[0].copy_(h[0], non_blocking=True)
    with torch.cuda.stream(s[1]): h[1].copy_(d[1], non_blocking=True)
run("H2D one stream, 256 MiB copies", one, n)
run("H2D two streams", two, 2 * n)
run("D2H one stream", back, n)
run("H2D + D2H at once (sum)", both, 2 * n)
for mb in (1, 4, 16, 64):
    m = mb << 20
    def small():
        with torch.cuda.stream(s[0]):
            for o in range(0, n, m): d[0][o:o + m].copy_(h[0][o:o + m], non_blocking=True)
    run(f"H2D one stream in {mb} MiB pieces", small, n, reps=3)
