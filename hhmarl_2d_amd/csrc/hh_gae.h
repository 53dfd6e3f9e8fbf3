/*
 * hh_gae.h — generalised advantage estimation over the [T, N, n_agents] rollout tensors the world kernel
 * writes (the reference leaves this to RLlib: train_hetero.py:216 gamma=0.99,
 * lambda_=0.95, batch_mode="complete_episodes").
 *
 *   delta_t = r_t + gamma * V_{t+1} * (1 - done_t) - V_t
 *   A_t     = delta_t + gamma * lambda * (1 - done_t) * A_{t+1},      R_t = A_t + V_t
 *
 * One lane per (arena, agent) walks its column backwards in time: the scan is sequential in t but every
 * load/store is unit-stride across lanes ([t][n][agent] layout), so each step of the scan is one fully
 * coalesced row — an HBM-streaming kernel (3 reads + 2 writes of 4 B per element).  `valid` (the
 * reference's "agent has a reward key this step") masks dead agents: their advantage and return are 0 and
 * they do not propagate.  `done` cuts the recursion at episode boundaries (auto-reset arenas).
 */
#ifndef HH_GAE_H
#define HH_GAE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void hh_k_gae(int T, int N, int nA, const float *__restrict__ reward,
                                                const float *__restrict__ value /* [T+1, N, nA] */,
                                                const uint8_t *__restrict__ valid, const uint8_t *__restrict__ done /* [T, N] */,
                                                float gamma, float lam, float *__restrict__ adv, float *__restrict__ ret) {
    const size_t cols = (size_t)N * nA;
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= cols) return;
    const size_t n = col / nA;
    float a_next = 0.0f;
    float v_next = value[(size_t)T * cols + col];
    for (int t = T - 1; t >= 0; t--) {
        const size_t i = (size_t)t * cols + col;
        const float nd = done[(size_t)t * N + n] ? 0.0f : 1.0f;
        const float v = value[i];
        float a = 0.0f;
        if (valid[i]) {
            const float delta = reward[i] + gamma * v_next * nd - v;
            a = delta + gamma * lam * nd * a_next;
        }
        adv[i] = a;
        ret[i] = valid[i] ? a + v : 0.0f;
        a_next = valid[i] ? a : 0.0f;
        v_next = v;
    }
}

#endif /* HH_GAE_H */
