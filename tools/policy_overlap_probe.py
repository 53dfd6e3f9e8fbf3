"""do two policy calls on two streams overlap on the chip?  time of one call, of two back to back on one stream, of two on two streams (eager launches, events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from hhmarl_2d_amd.pilots import PolicyBank
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 19600
dev = torch.device("cuda", 0)
banks = [PolicyBank.random_init(dev, seed=k, max_rows=rows) for k in range(2)]
g = torch.Generator(device="cuda"); g.manual_seed(0)
obs = [torch.rand((rows, 30), device=dev, generator=g) for _ in range(2)]
sel = [torch.tensor([5, 6, 9, 10], dtype=torch.uint8, device=dev)[torch.randint(0, 4, (rows,), device=dev, generator=g)] for _ in range(2)]
act = [torch.zeros((rows, 4), dtype=torch.int8, device=dev) for _ in range(2)]
for k in range(2):
    banks[k].act(obs[k], sel[k], act[k])     # bins once; later calls re-use the lists (sel=None)
torch.cuda.synchronize()
print("kernel:", banks[0].kernel_name(rows))
s = [torch.cuda.Stream(), torch.cuda.Stream()]
def timed(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def one():
    banks[0].act(obs[0], None, act[0])
def two_serial():
    banks[0].act(obs[0], None, act[0]); banks[1].act(obs[1], None, act[1])
def two_streams():
    cur = torch.cuda.current_stream()
    for k in range(2):
        s[k].wait_stream(cur)
        with torch.cuda.stream(s[k]):
            banks[k].act(obs[k], None, act[k])
    for k in range(2):
        cur.wait_stream(s[k])
print(f"rows {rows}: one call {timed(one):.1f} us, two on one stream {timed(two_serial):.1f} us, two on two streams {timed(two_streams):.1f} us")
