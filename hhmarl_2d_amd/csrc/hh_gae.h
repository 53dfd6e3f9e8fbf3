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

/* The recursion exactly as RLlib 2.4 applies it to the reference's step stream (train_hetero.py:212,216 / train_hier.py:182,186,
 * ray/rllib/evaluation/postprocessing.py compute_advantages + discount_cumsum, and the sampler's
 * `rewards[env_id].get(agent_id, 0.0)`):
 *   - the reference returns an observation for EVERY agent id on every step, dead ones as zeros (env_hetero.py:65-103), but a
 *     reward only for the ids alive at step start (env_hetero.py:217-223), and `terminateds` carries "__all__" only: RLlib keeps
 *     collecting a row per agent until the episode ends, with reward 0.0 where the key is missing — nothing is masked;
 *   - terminateds["__all__"] is set on every episode end (also at the horizon: env_base.py:108 returns the same dict as
 *     terminateds and truncateds), so the value after an episode's last row is last_r = 0.0;
 *   - vpred_t = np.concatenate([VF_PREDS (float32), np.array([last_r])]) is a float64 array, so delta_t = r_t + gamma * V_{t+1} - V_t
 *     is evaluated in float64 from the float32 inputs; A = scipy.signal.lfilter([1], [1, -gamma*lambda], delta[::-1])[::-1], i.e.
 *     A_t = delta_t + (gamma*lambda) A_{t+1} in float64; advantages = float32(A_t), value target = float32(A_t(float64) + V_t).
 * `done` cuts the columns into episodes; the rows after an arena's last done inside the window bootstrap from value[T] (RLlib's
 * truncated-rollout rule; with batch_mode="complete_episodes" they are carried into the next batch: rollout.episode_segments). */
__global__ __launch_bounds__(256) void hh_k_gae_rllib(int T, int N, int nA, const float *__restrict__ reward,
                                                      const float *__restrict__ value /* [T+1, N, nA] */, const uint8_t *__restrict__ done /* [T, N] */,
                                                      double gamma, double lam, float *__restrict__ adv, float *__restrict__ ret) {
    const size_t cols = (size_t)N * nA;
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= cols) return;
    const size_t n = col / nA;
    const double gl = gamma * lam;
    double a_next = 0.0;
    float v_next = value[(size_t)T * cols + col];
    for (int t = T - 1; t >= 0; t--) {
        const size_t i = (size_t)t * cols + col;
        const bool last = done[(size_t)t * N + n] != 0;
        const float v = value[i];
        const double delta = __dsub_rn(__dadd_rn((double)reward[i], __dmul_rn(gamma, last ? 0.0 : (double)v_next)), (double)v);
        const double a = __dadd_rn(delta, __dmul_rn(gl, last ? 0.0 : a_next));
        adv[i] = (float)a;
        ret[i] = (float)__dadd_rn(a, (double)v);
        a_next = a;
        v_next = v;
    }
}

#endif /* HH_GAE_H */
