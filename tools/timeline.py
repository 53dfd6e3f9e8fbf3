"""Workgroup timeline of the commander step with the pilot networks in the loop (variant rows, K sub-worlds on K streams in one HIP graph).
Needs a -DHH_TIMELINE build (bash tools/build_variant.sh timeline -DHH_TIMELINE; HH_WORLD_LIB=hhmarl_2d_amd/lib/abl_timeline.so): every workgroup of
hh_k_hier_oct_v (tag 1) and hh_k_policy_w16 (tag 2 = a tile with rows, 3 = early exit) records CU, start and end in s_memrealtime ticks (10 ns).
Prints, for one replayed commander step: per launch (clustered by sub-world and tag) first start / last end / workgroups / CUs touched, then the chip's
occupancy over time (CUs holding a policy tile, CUs holding phase waves, idle) and the per-sub-step period.
    python tools/timeline.py [arenas] [K] [warm-up commander steps]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
os.environ.setdefault("HH_POLICY_W", "3")
from hhmarl_2d_amd import _lib
from hhmarl_2d_amd.env_hier import macro_step
from hhmarl_2d_amd.pilots import VariantNetPilot
from hhmarl_2d_amd.world import World, make_config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = _lib.lib()
lib.hh_debug_timeline.restype = C.c_longlong
lib.hh_debug_timeline.argtypes = [C.c_void_p, C.c_longlong]
n = N // K
worlds = [World(make_config(n_arenas=n, env_kind=1, seed=0, auto_reset=True, arena_offset=k * n)) for k in range(K)]
for w in worlds:
    w.reset()
pilots = [VariantNetPilot(w, seed=0) for w in worlds]
cmd = [(torch.rand((n, 3), device="cuda") * 3).to(torch.int8).contiguous() for _ in range(K)]
outs = [w.alloc_outputs() for w in worlds]
pbufs = [w.alloc_pilot_variants() for w in worlds]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
streams = bench.make_streams(torch, K, allow_default=os.environ.get("HH_TL_JOINED") is None)


def step():
    cur = torch.cuda.current_stream()
    for k in range(K):
        streams[k].wait_stream(cur)
        with torch.cuda.stream(streams[k]):
            macro_step(worlds[k], cmd[k], pilots[k], out=outs[k], pilot_buf=pbufs[k])
    for k in range(K):
        cur.wait_stream(streams[k])


for _ in range(2):
    step()
torch.cuda.synchronize()
PIPE = os.environ.get("HH_TL_JOINED") is None   # one graph per sub-world, replayed on its own stream: the sub-worlds do not meet at commander-step boundaries
STEPS = 3 if PIPE else 1
if PIPE:
    graphs = []
    for k in range(K):
        gk = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gk, stream=(streams[k] if streams[k].cuda_stream != 0 else torch.cuda.Stream())):
            macro_step(worlds[k], cmd[k], pilots[k], out=outs[k], pilot_buf=pbufs[k])
        graphs.append(gk)
    def replay(m):
        cur = torch.cuda.current_stream()
        for k in range(K):
            streams[k].wait_stream(cur)
        for _ in range(m):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    graphs[k].replay()
        for k in range(K):
            cur.wait_stream(streams[k])
else:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    def replay(m):
        for _ in range(m):
            g.replay()
WARM = int(sys.argv[3]) if len(sys.argv) > 3 else 600   # commander steps before the recorded one: the alive fractions (row counts) settle after a few hundred
for i in range(WARM // 8):
    for k in range(K):
        cmd[k].copy_((torch.rand((n, 3), device="cuda") * 3).to(torch.int8))
    replay(8)
torch.cuda.synchronize()
lib.hh_debug_timeline(None, 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
replay(STEPS)
e1.record()
torch.cuda.synchronize()
print(f"N={N} K={K}: the {STEPS} recorded commander step(s) took {e0.elapsed_time(e1) / STEPS:.3f} ms each (instrumented build; {'one graph per sub-world' if PIPE else 'one graph, joined at every step'})")
cap = 1 << 20
buf = np.zeros((cap, 4), dtype=np.uint64)
cnt = lib.hh_debug_timeline(buf.ctypes.data, cap)
buf = buf[:cnt]
tag = (buf[:, 0] & 0xff).astype(int)
wave = ((buf[:, 0] >> 8) & 0xff).astype(int)
aux = (buf[:, 0] >> 32).astype(np.int64)
hw = (buf[:, 1] & 0xffffffff).astype(np.int64)
xcc = (buf[:, 1] >> 32).astype(np.int64)
cu = xcc * 4096 + ((hw >> 8) & 0xf) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 7)   # (xcc, se, sh, cu): unique per CU
t0 = buf[:, 2].astype(np.int64)
t1 = t0 + (buf[:, 3] & 0xfffff).astype(np.int64)
marks = np.stack([((buf[:, 3] >> (20 + 11 * k)) & 0x7ff).astype(np.int64) * 0.02 for k in range(4)], axis=1)   # us since the workgroup's start
base = t0.min()
t0 = (t0 - base) / 100.0   # us
t1 = (t1 - base) / 100.0
print(f"{cnt} workgroup records, {len(np.unique(cu))} distinct CUs seen, span {t1.max():.1f} us")
# sub-worlds: a phase launch carries its world's arena offset, a policy call its bank's counter address; a sub-world's phase launches and policy calls
# alternate without overlapping, which pairs the two
cls = np.where(tag == 1, 1, 2)
w_ids, p_ids = sorted(set(aux[cls == 1])), sorted(set(aux[cls == 2]))
def overlap(a, b):
    A = np.where((aux == a) & (cls == 1))[0]; B = np.where((aux == b) & (cls == 2) & (tag == 2))[0]
    tot = 0.0
    for lo, hi in zip(t0[A][::16], t1[A][::16]):
        tot += np.clip(np.minimum(hi, t1[B]) - np.maximum(lo, t0[B]), 0, None).sum()
    return tot
pair, free = {}, list(p_ids)
for a in w_ids:
    b = min(free, key=lambda x: overlap(a, x))
    pair[a] = b
    free.remove(b)
sub = np.zeros(cnt, dtype=int)
for k, a in enumerate(w_ids):
    sub[(cls == 1) & (aux == a)] = k
    sub[(cls == 2) & (aux == pair[a])] = k
launches = []
for k in range(len(w_ids)):
    m = np.where(sub == k)[0]
    order = m[np.argsort(t0[m], kind="stable")]
    cur = [order[0]]
    for i in order[1:]:
        if cls[i] != cls[cur[0]]:
            launches.append((k, cls[cur[0]], np.array(cur)))
            cur = [i]
        else:
            cur.append(i)
    launches.append((k, cls[cur[0]], np.array(cur)))
launches.sort(key=lambda x: t0[x[2]].min())
print("sub-world  kind   start     end     dur   wgs  with-rows  CUs   median-wg-us  max-wg-us   gap-after-previous-of-this-sub-world")
last_end = {}
for k, c, idx in launches[:30 * K]:
    work = idx[tag[idx] != 3]
    st = t0[idx].min()
    print(f"   {k:3d}      {'W' if c == 1 else 'P'}   {st:8.1f} {t1[idx].max():8.1f} {t1[idx].max() - st:6.1f} {len(idx):5d} {len(work):6d} {len(np.unique(cu[work])) if len(work) else 0:5d}"
          f"      {np.median(t1[work] - t0[work]) if len(work) else 0:6.1f}     {(t1[work] - t0[work]).max() if len(work) else 0:6.1f}     {st - last_end.get(k, st):6.1f}")
    last_end[k] = t1[idx].max()
# occupancy over time, 1 us bins
T = int(np.ceil(t1.max())) + 1
ncu = len(np.unique(cu))
cu_ids = {c_: i for i, c_ in enumerate(np.unique(cu))}
occP = np.zeros((ncu, T), dtype=bool)
occW = np.zeros((ncu, T), dtype=np.int16)
for i in range(cnt):
    if tag[i] == 3:
        continue
    a_, b_ = int(t0[i]), int(np.ceil(t1[i]))
    if tag[i] == 2:
        if wave[i] == 0:
            occP[cu_ids[cu[i]], a_:b_] = True
    else:
        occW[cu_ids[cu[i]], a_:b_] += 1
p_cus = occP.sum(0)
w_cus = (occW > 0).sum(0)
w_waves = occW.sum(0)
both = (occP & (occW > 0)).sum(0)
idle = ncu - (occP | (occW > 0)).sum(0)
print(f"mean over the step: CUs with a policy tile {p_cus.mean():.1f}, CUs with phase waves {w_cus.mean():.1f} ({w_waves.mean():.1f} waves), both {both.mean():.1f}, idle {idle.mean():.1f} of {ncu}")
print("time-us  P-CUs  W-CUs  W-waves  idle   (every 10 us, first 400 us)")
for t in range(0, min(T, 400), 10):
    print(f"{t:6d} {p_cus[t]:6d} {w_cus[t]:6d} {w_waves[t]:7d} {idle[t]:6d}")
wm = np.where((tag == 1) & (marks[:, 2] > 0))[0]   # act-tick waves (the begin launch sets no mark 1 / 2)
if len(wm):
    d = t1[wm] - t0[wm]
    md = np.median(marks[wm], axis=0)
    print(f"phase waves (act-tick launches, n={len(wm)}), median us since the wave's start: state loaded + positions {md[0]:.2f} | both sides acted {md[1]:.2f} | tick done {md[2]:.2f} | "
          f"rows built, stored, listed {md[3]:.2f} | end (state stored) {np.median(d):.2f}")
pl = [x for x in launches if x[1] == 2 and x[0] == 0]
if len(pl) > 4:
    starts = np.array([t0[x[2]].min() for x in pl])
    print("sub-world 0: policy call starts every", np.round(np.diff(starts), 1))
for p in pilots:
    p.close()
