"""do two policy calls on two streams overlap on the chip?  time of one call, of two back to back on one stream, of two on two streams (eager launches, events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from hhmarl_2d_amd.pilots import PolicyBank
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 19600
nets = int(sys.argv[2]) if len(sys.argv) > 2 else 4   # how many of the four networks the rows use (1: every row flies Fight1 — a 1.2 MB weight working set per bank)
dev = torch.device("cuda", 0)
banks = [PolicyBank.random_init(dev, seed=k, max_rows=rows) for k in range(2)]
g = torch.Generator(device="cuda"); g.manual_seed(0)
obs = [torch.rand((rows, 30), device=dev, generator=g) for _ in range(2)]
sel = [torch.tensor([5, 6, 9, 10], dtype=torch.uint8, device=dev)[torch.randint(0, nets, (rows,), device=dev, generator=g)] for _ in range(2)]
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
def graphed(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):     # ten repetitions per replay: the replay's own launch cost is spread
            fn()
    return lambda: gr.replay()
g1, g2, g3 = graphed(one), graphed(two_serial), graphed(two_streams)
print(f"rows {rows}, {nets} network(s), replayed from a HIP graph (10 repetitions each): one call {timed(g1, 50) / 10:.1f} us, two on one stream {timed(g2, 50) / 10:.1f} us, "
      f"two on two streams {timed(g3, 50) / 10:.1f} us")
print(f"rows {rows}, {nets} network(s): one call {timed(one):.1f} us, two on one stream {timed(two_serial):.1f} us, two on two streams {timed(two_streams):.1f} us")
