"""How many dependent launch chains run side by side?  K streams, each a chain of M short kernels (torch.cuda._sleep) replayed from one HIP graph per stream (the
bench's pipelined form) or issued eagerly; concurrent chains take the time of one, chains that share a hardware queue the sum.
    python tools/queue_probe.py            (HH_BENCH_STREAMS=default0|side|pool, GPU_MAX_HW_QUEUES=n)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
CYC = 100_000   # ~45 us at 2.2 GHz
M = 40
torch.cuda._sleep(CYC); torch.cuda.synchronize()
for K in (1, 2, 3, 4, 5, 6, 8):
    streams = bench.make_streams(torch, K)   # K > 3: the first chain on the default stream (HH_BENCH_STREAMS=side: side streams only)
    graphs = []
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            torch.cuda._sleep(CYC)
    torch.cuda.synchronize()
    for s in streams:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=(s if s.cuda_stream != 0 else torch.cuda.Stream())):
            for _ in range(M):
                torch.cuda._sleep(CYC)
        graphs.append(g)
    def run(graph):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(3):
            for s, g in zip(streams, graphs):
                with torch.cuda.stream(s):
                    if graph: g.replay()
                    else:
                        for _ in range(M): torch.cuda._sleep(CYC)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3 * 1e3
    run(True); run(False)
    print(f"K={K}: graphs {run(True):7.3f} ms   eager {run(False):7.3f} ms   (one chain of {M} kernels alone: see K=1)", flush=True)
