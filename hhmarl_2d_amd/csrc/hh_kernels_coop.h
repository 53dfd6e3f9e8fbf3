/*
 * hh_kernels_coop.h — HighLevelEnv.step with the pilot networks INSIDE: one persistent cooperative launch per commander step.
 *
 * The reference evaluates a frozen network for every live unit inside every sub-step, agents before opponents, because the
 * opponents observe the agents' same-sub-step weapon flags (envs/env_hier.py:114-140, env_base.py:349-398).  Launch by launch
 * that is 66 dependent kernels per commander step — hh_hl_begin, 16 x { policy, hh_hl_agents_act, policy, hh_hl_tick }, hh_hl_end —
 * and the world phases among them are single passes through ~50 KB of cold code that re-read the whole state each time
 * (14 / 36 us per launch for 8 us of arithmetic per sub-step).  Here the same device functions run inside ONE
 * kernel whose workgroups stay resident for the whole step:
 *
 *     world(BEGIN) | policy(agents) | world(AGENTS_ACT) | policy(opponents) | world(TICK) | policy(agents) | ... | world(END)
 *
 * separated by grid barriers.  A workgroup is 256 threads = the policy kernel's tile shape (hh_policy_kernel_h16.h: 32 rows of one
 * network, two workgroups per CU); during a world phase each of its waves runs oct_phase_body (hh_kernels_oct.h: eight arenas per
 * wave, register exchange) for the arena groups wave * gridDim + blockIdx, ..., with its own slice of the same LDS.  The world
 * phases bin the pilot rows they emit into the policy bank's row lists (two counter sets alternate: the set a policy phase has
 * consumed is cleared while the next world phase fills the other); the policy phases walk the tiles grid-stride.  The code is
 * fetched once and stays in the instruction caches, nothing is dispatched or drained between phases.
 * Same device functions, same operation order: bit-identical to the launch-by-launch path (tests/test_gpu_hier_nets.py).
 *
 * MEASURED AND NOT THE DEFAULT (round 3): 7.9 ms per commander step of 8192 arenas against 2.57 ms launch by launch.  Merged into one
 * kernel body the policy tiles lose their register allocation (hh_k_policy_h alone: no spill; here 83 scratch accesses and 639
 * v_readlane / v_writelane between the MFMAs, a policy phase takes 122 us instead of 48), the world phases run the 256-register form
 * with spills (24 us per phase), and half the workgroups wait at every barrier for the ones that drew two policy tiles.  Phases as
 * separate noinline functions were worse still (13 ms: the callees' own spills).  hh_hl_step_nets stays as the tested one-launch
 * alternative (bench.py --workload hier --pilot net --coop); the product path is the 66-launch HIP graph.
 *
 * The grid barrier is a sense-reversing counter in global memory (agent-scope release / acquire).  It needs every workgroup
 * resident: the launch is cooperative (hipLaunchCooperativeKernel refuses a grid that does not fit), and a waiting workgroup
 * gives up after a bounded number of polls and raises an error flag that makes all others leave — a mistake here must not hang
 * the GPU.
 */
#ifndef HH_KERNELS_COOP_H
#define HH_KERNELS_COOP_H

#include "hh_kernels_oct.h"

#define HH_COOP_SET_INTS (16 * HH_BIN_STRIDE) /* ints of one counter set: rows per network [8] + spare, each on its own 128-byte line */
#define HH_COOP_SPIN_LIMIT (1u << 21)         /* polls (each >= 64 x 8 cycles of s_sleep): ~0.5 s before a workgroup gives up */

struct CoopCtl {
    unsigned *bar_cnt, *bar_gen; /* grid barrier */
    int *err;                    /* != 0: a barrier timed out, the step is invalid */
    int *run_cnt;                /* [16] arenas still inside their macro step after sub-step k */
    int *counts;                 /* [2][HH_COOP_SET_INTS] the two sets of row counters */
};

__device__ __forceinline__ bool coop_grid_barrier(const CoopCtl &ctl, unsigned nblocks, int *sflag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned g = __hip_atomic_load(ctl.bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned prev = __hip_atomic_fetch_add(ctl.bar_cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT); /* release: this workgroup's writes */
        if (prev == nblocks - 1) {
            __hip_atomic_store(ctl.bar_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(ctl.bar_gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(ctl.bar_gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > HH_COOP_SPIN_LIMIT || __hip_atomic_load(ctl.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    __hip_atomic_store(ctl.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        *sflag = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); /* every wave: what the other workgroups wrote before the barrier is visible */
    return *sflag != 0;
}

/* everything the phases read, as ONE kernel argument: the phases are separate (noinline) functions with a register allocation of
 * their own — inlined into one body, the world code's live ranges made the policy GEMM loops spill (83 scratch accesses and 639
 * v_readlane / v_writelane between the MFMAs, 122 us per policy phase) — and they fetch what they need from the kernel-argument
 * segment (constant address space: scalar loads) instead of holding a hundred pointers in registers across the whole step */
struct CoopParams {
    DevPtrs P;
    DevCfg c;
    HhpBank bank;
    HhpBankH bankh;
    CoopCtl ctl;
    const int8_t *cmd;
    float *pilot_obs;
    int8_t *actions;
    float *obs_out, *reward_out;
    uint8_t *valid_out, *done_out;
    int *counters;
    int n_nets;
};
typedef const __attribute__((address_space(4))) CoopParams *CoopParamsPtr;

/* (bodies for the device pass only: the host pass cannot copy structs out of the constant address space and never runs them) */
__device__ __forceinline__ void coop_world_phase(CoopParamsPtr kp, unsigned char *ldsb, int phase, int set, int sub) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
    OctShared &sh = *reinterpret_cast<OctShared *>(ldsb + (size_t)wave * ((sizeof(OctShared) + 15) / 16 * 16));
    DevPtrs Pw = kp->P;
    const DevCfg c = kp->c;
    const CoopCtl ctl = kp->ctl;
    Pw.pol_counts = ctl.counts + set * HH_COOP_SET_INTS;
    int *counters = kp->counters;
    unsigned long long *tick_total = counters ? reinterpret_cast<unsigned long long *>(counters + 2) : nullptr;
    const int n_groups = (c.N + 7) / 8, nb = (int)gridDim.x;
    for (int grp = wave * nb + (int)blockIdx.x; grp < n_groups; grp += 4 * nb)
        oct_phase_body<2>(Pw, c, phase, grp, lane, sh, kp->cmd, kp->actions, kp->pilot_obs, nullptr, kp->obs_out, kp->reward_out, kp->valid_out, kp->done_out,
                          phase == HH_HL_TICK ? ctl.run_cnt + sub : nullptr, tick_total);
#endif
}

__device__ __forceinline__ void coop_policy_phase(CoopParamsPtr kp, unsigned char *ldsb, int set) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int *cs = kp->ctl.counts + set * HH_COOP_SET_INTS;
    const int n_nets = kp->n_nets, max_rows = kp->P.pol_max_rows;
    int cn[HH_POLICY_MAX_NETS];
#pragma unroll
    for (int n = 0; n < HH_POLICY_MAX_NETS; n++) cn[n] = n < n_nets ? min(cs[n * HH_BIN_STRIDE], max_rows) : 0;
    const HhpBank bank = kp->bank;
    const HhpBankH bankh = kp->bankh;
    hhp_forward_tiles<1>(bank, bankh, cn, kp->pilot_obs, 30, kp->P.pol_lists, max_rows, kp->actions, nullptr, ldsb, (int)blockIdx.x, (int)gridDim.x);
#endif
}

__global__ __launch_bounds__(256, 2) void hh_k_hier_nets(CoopParams params) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __align__(16) unsigned char ldsb[]; /* the policy tile; during a world phase wave w owns an OctShared inside it */
    __shared__ int sflag;
    static_assert(4 * ((sizeof(OctShared) + 15) / 16 * 16) <= HHPH_LDS_BYTES(1), "the world waves' exchange areas alias the policy tile");
    CoopParamsPtr kp = (CoopParamsPtr)__builtin_amdgcn_kernarg_segment_ptr(); /* the one argument sits at offset 0 */
    const CoopCtl ctl = kp->ctl;
    const unsigned nb = gridDim.x;
    int set = 0;
    auto consumed = [&]() { /* the set the policy phase just read: cleared while the next world phase fills the other one */
        if (blockIdx.x == 0 && threadIdx.x < HH_POLICY_MAX_NETS) ctl.counts[set * HH_COOP_SET_INTS + (int)threadIdx.x * HH_BIN_STRIDE] = 0;
        set ^= 1;
    };
#ifdef HH_COOP_PROFILE /* tuning builds: wall-clock ticks (100 MHz) block 0 spends in world phases / policy phases / barriers -> ctl.run_cnt[16..] */
    long long pt_ = wall_clock64(), pacc_[3] = {0, 0, 0};
#define HH_CT(k) do { long long t_ = wall_clock64(); pacc_[k] += t_ - pt_; pt_ = t_; } while (0)
#else
#define HH_CT(k)
#endif
    if (blockIdx.x == 0 && threadIdx.x < 16) ctl.run_cnt[threadIdx.x] = 0;
    coop_world_phase(kp, ldsb, HH_HL_BEGIN, set, 0);
    HH_CT(0);
    if (!coop_grid_barrier(ctl, nb, &sflag)) return;
    HH_CT(2);
    for (int sub = 0; sub < 16; sub++) {
        coop_policy_phase(kp, ldsb, set); /* the agents' rows */
        HH_CT(1);
        if (!coop_grid_barrier(ctl, nb, &sflag)) return;
        HH_CT(2);
        consumed();
        coop_world_phase(kp, ldsb, HH_HL_AGENTS_ACT, set, sub);
        HH_CT(0);
        if (!coop_grid_barrier(ctl, nb, &sflag)) return;
        HH_CT(2);
        coop_policy_phase(kp, ldsb, set); /* the opponents' rows */
        HH_CT(1);
        if (!coop_grid_barrier(ctl, nb, &sflag)) return;
        HH_CT(2);
        consumed();
        coop_world_phase(kp, ldsb, HH_HL_TICK, set, sub);
        HH_CT(0);
        if (!coop_grid_barrier(ctl, nb, &sflag)) return;
        HH_CT(2);
        if (__hip_atomic_load(ctl.run_cnt + sub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) break; /* no arena is inside its macro step any more */
    }
    coop_world_phase(kp, ldsb, HH_HL_END, set, 0); /* also drops the rows the last tick binned */
    HH_CT(0);
#ifdef HH_COOP_PROFILE
    if (blockIdx.x == 0 && threadIdx.x == 0) { ctl.run_cnt[16] = (int)pacc_[0]; ctl.run_cnt[17] = (int)pacc_[1]; ctl.run_cnt[18] = (int)pacc_[2]; }
#endif
#endif /* __HIP_DEVICE_COMPILE__ */
}

#endif /* HH_KERNELS_COOP_H */
