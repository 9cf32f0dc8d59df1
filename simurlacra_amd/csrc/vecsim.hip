// vecsim.hip -- libvecsim: HIP kernels (gfx950 / CDNA4) + the C-ABI of include/vecsim.h.
//
// One environment per wavefront lane, all per-env data fp32 struct-of-arrays [dim][ld] so that every load/store of a
// wave is one contiguous 256-B segment.  The step of SimPyEnv (reward -> clip -> dead zone -> integrate -> done ->
// final reward -> observe, P/environments/pysim/base.py:217-241) is ONE kernel; there is no CPU fallback anywhere:
// every entry point either runs on the GPU or returns an error.
//
// Kernels:
//   k_step          vs_step           one step per launch, actions from the caller (policy in the loop)
//   k_rollout       vs_step_random    k steps per launch, on-device uniform policy, state in registers, optional records
//   k_rollout_ws    vs_step_random    the same on two cooperating waves per 64 envs (physics | reward + records) through
//                                     LDS: what runs up to one 256-env workgroup per compute unit (65 536 envs)
//   k_*_mixed       vs_mixed_*        several families in one launch (one workgroup = one family)
//   k_step_jac      vs_step_jac       step + Jacobians by forward-mode dual numbers
//   k_reset / k_set_params / k_sample_params / k_observe   control path
// Variants carrying the wrapper pipeline (action noise / delay, observation normalisation / noise) are separate
// instantiations (template parameter PIPE): the default kernels do not pay for it.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/vecsim.h"
#include "vecsim_envs.h"

namespace vs {

constexpr int BLOCK = 256;

struct DrSpecs {
    int n;
    vs_dp_spec s[MAXP];
};

// The cheap wrappers scripts stack around these envs, fused into the step instead of being Python objects around it:
// GaussianActNoiseWrapper (P/environment_wrappers/action_noise.py:38-79), ActDelayWrapper (action_delay.py:37-112),
// ObsNormWrapper (observation_normalization.py:41-126) and GaussianObsNoiseWrapper (observation_noise.py:38-73).
// Action side, after ActNormWrapper's de-normalisation:  a -> [+ noise] -> delay ring -> [+ noise] -> env.step
// Observation side:  obs' = obs * scale + shift + std * z   (any stack of norm / noise stages composes to this)
struct Pipe {
    int act_on, obs_on;
    int delay;              // ActDelayWrapper: the action applied at step t is the one commanded at t - delay (0 before)
    int act_noise, obs_noise;
    int noise_normed;       // the noise wrapper sits outside ActNormWrapper: its draw is in [-1, 1] units
    int noise_after_delay;  // the noise wrapper sits inside ActDelayWrapper
    float a_mean[MAXA], a_std[MAXA];
    float o_scale[MAXO], o_shift[MAXO], o_std[MAXO];
    float* ring;            // [delay][A][ld]
    uint64_t seed;          // noise streams: Philox(seed; env, RNG_*_NOISE, episode index << 32 | step)
};

// device pointers of one handle, passed to kernels by value
struct Dev {
    Pipe pipe;
    float *state, *hidden, *obs, *rew, *ret, *consts, *params, *consts_uni;
    uint8_t *done, *failed, *err, *yielded;
    int* step;
    uint32_t* ep_idx;   // per-env episode counter: the Philox counter of the NEXT reset of that env
    // per-env statistics of completed episodes since vs_clear_episodes: plain per-lane accumulators, no atomics
    uint32_t* es_count;
    float* es_retsum;
    int* es_lensum;
    int log_episodes;   // opt-in: also append (return, length, env) to the global ring with ballot compaction
    DrSpecs drv;        // the live randomizer BY VALUE: kernel arguments live in the constant address space, so a spec is
                        // fetched with scalar loads (lgkmcnt).  Behind a pointer it was a vector load per spec, and on
                        // gfx950 a vector load waits for every older record store (one in-order vmcnt): 1-2 us each
    uint32_t idx0;      // global index of lane 0: every Philox stream is keyed by (idx0 + lane), so results do not depend
                        // on how a set of envs is split into handles, batches or GPUs
    const float* pbuf;  // DomainRandWrapperBuffer: [P][pbuf_n] parameter sets (nullptr: none)
    int pbuf_n, pbuf_mode;  // number of sets; 0 cyclic, 1 random
    int dr_n;           // its number of specs, by value: the reset path must not wait on a load to learn there is none
    float* ep_ret;
    int *ep_len, *ep_env;
    unsigned* ep_count;
    unsigned ep_cap;
    float* traj_rec;     // packed per-step records, see store_record
    uint8_t* traj_done;
    int traj_t0;  // record row offset of the next vs_step_random(record = 1)
    float *jac_s, *jac_r, *jac_o;  // step Jacobians (vs_step_jac), allocated on first use
    int n, ld;
};

// ------------------------------------------------------------------------------------------------- wrapper pipeline
__device__ __forceinline__ void box_muller(uint32_t b0, uint32_t b1, float& z0, float& z1) { Rng::box_muller(b0, b1, z0, z1); }

// a: the action in the env's own units (after ActNormWrapper); step: curr_step of the lane before this step
template <class E>
__device__ __forceinline__ void pipe_act(const Dev& d, int i, uint32_t epi, int step, const float* c, float* a) {
    static_assert(E::A <= MAXA, "action width");
    const Pipe& p = d.pipe;
    float nz[E::A];
#pragma unroll
    for (int j = 0; j < E::A; ++j) nz[j] = 0.f;
    if (p.act_noise) {
        uint4 b = Rng::philox(p.seed, d.idx0 + (uint32_t)i, RNG_ACT_NOISE, ((uint64_t)epi << 32) | (uint32_t)step);
        float z[2];
        box_muller(b.x, b.y, z[0], z[1]);
        float lb[E::A], ub[E::A];
        E::act_bounds(c, lb, ub);
#pragma unroll
        for (int j = 0; j < E::A; ++j)
            nz[j] = (p.a_mean[j] + p.a_std[j] * z[j]) * (p.noise_normed ? 0.5f * (ub[j] - lb[j]) : 1.0f);
    }
    if (!p.noise_after_delay) {
#pragma unroll
        for (int j = 0; j < E::A; ++j) a[j] += nz[j];
    }
    if (p.delay > 0) {
        // the queue of the reference starts as `delay` zero actions at reset; a ring slot is only read once the
        // episode has written it, so nothing has to be cleared at reset
        int slot = step % p.delay;
#pragma unroll
        for (int j = 0; j < E::A; ++j) {
            float* r = p.ring + ((size_t)slot * E::A + j) * d.ld + i;
            float prev = step >= p.delay ? *r : 0.f;
            *r = a[j];
            a[j] = prev;
        }
    }
    if (p.noise_after_delay) {
#pragma unroll
        for (int j = 0; j < E::A; ++j) a[j] += nz[j];
    }
}

// step: curr_step of the lane the observation belongs to (0 for the observation reset() returns)
template <class E>
__device__ __forceinline__ void pipe_obs(const Dev& d, int i, uint32_t epi, int step, const float* ob, float* out) {
    static_assert(E::O <= MAXO, "observation width");
    const Pipe& p = d.pipe;
    float z[MAXO];
#pragma unroll
    for (int j = 0; j < MAXO; ++j) z[j] = 0.f;
    if (p.obs_noise) {
        Rng g(p.seed, d.idx0 + (uint32_t)i, RNG_OBS_NOISE, ((uint64_t)epi << 32) | (uint32_t)step);
#pragma unroll
        for (int j = 0; j < E::O; j += 2) {
            uint32_t b0 = g.next(), b1 = g.next();
            box_muller(b0, b1, z[j], z[j + 1]);
        }
    }
#pragma unroll
    for (int j = 0; j < E::O; ++j) out[j] = fmaf(ob[j], p.o_scale[j], p.o_shift[j]) + p.o_std[j] * z[j];
}

// ---------------------------------------------------------------------------------------------------- reward / step
// DesStateTask.step_rew / RadiallySymmDesStateTask.step_rew + the three reward functions
// (P/tasks/desired_state.py:107-110,146-155; P/tasks/reward_functions.py:212-221,237-244,276-282)
template <class E, class R>
__device__ __forceinline__ R step_reward(const Task& T, const float* c, const R* s, const R* a_raw) {
    R cost = 0.f;
#pragma unroll
    for (int j = 0; j < E::S; ++j) {
        R e = T.des[j] - s[j];
        if (E::RADIAL >= 0) {
            if (j == E::RADIAL) e = fmod_2pi(e);
            e = fold_pi(e);  // all dims (Q4)
        }
        cost += e * (T.qd[j] * e);
    }
    R ca = 0.f;
#pragma unroll
    for (int j = 0; j < E::A; ++j) ca += a_raw[j] * (T.rd[j] * a_raw[j]);  // err_a = -act
    cost += ca;
    if (E::REW == REW_QUADR) return -cost;
    if (E::REW == REW_EXP) return exp_neg_fast(-cost);
    return exp_neg_fast(-c[E::CMAX >= 0 ? E::CMAX : 0] * cost);
}

// not state_space.contains(s') (Q9, Q10) for a symmetric box: some |s_j| > hi_j  <=>  max_j (|s_j| - hi_j) > 0.
// One compare at the end instead of one per dimension: every v_cmp feeds a scalar mask and a chain of s_or, and each
// VALU -> SALU hand-over stalls a lone wave.  The differences are exact (|s| - hi is 0 or at least an ulp of values of order
// 1..100, never subnormal); NaN dimensions drop out of the max (maxNum), as `NaN > hi` is false.
template <int S>
__device__ __forceinline__ bool outside_symmetric_box(const float* sv, const float* hi) {
    float m = fabsf(sv[0]) - hi[0];
#pragma unroll
    for (int j = 1; j < S; ++j) m = fmaxf(m, fabsf(sv[j]) - hi[j]);
    return m > 0.f;
}

template <class R>
struct StepOutT {
    R rew;
    bool done, failed, err;
};
using StepOut = StepOutT<float>;

// SimPyEnv.step for one lane (P/environments/pysim/base.py:217-241); s, h, step, yielded are updated in place.
// ob: observe() of the pre-step state if the caller holds it in registers (saves the trig it shares with the dynamics)
template <class E, class R>
__device__ __forceinline__ StepOutT<R> step_one(const Task& T, const float* c, R* s, R* h, const R* a_raw, int& step,
                                                bool& yielded, const R* ob, const Dev* dp = nullptr, int lane = 0,
                                                uint32_t epi = 0u) {
    StepOutT<R> o;
    // ActNormWrapper._process_act (action_normalization.py:66-72), branch-free: a wave-uniform select keeps the step one
    // basic block for the scheduler
    R an[E::A];
    {
        const bool nrm = (T.flags & VS_FLAG_ACT_NORM) != 0;
        float lb[E::A], ub[E::A];
        E::act_bounds(c, lb, ub);
#pragma unroll
        for (int j = 0; j < E::A; ++j) {
            R m = lb[j] + (a_raw[j] + 1.0f) * (ub[j] - lb[j]) * 0.5f;
            an[j] = vsel(nrm, m, a_raw[j]);
        }
        if constexpr (std::is_same<R, float>::value) {
            if (dp && dp->pipe.act_on) pipe_act<E>(*dp, lane, epi, step, c, an);  // wave-uniform branch
        }
        a_raw = an;
    }
#ifdef VS_ABLATE_REWARD  // diagnostic builds only (profiling by ablation); never defined in the shipped library
    o.rew = a_raw[0];
#else
    o.rew = step_reward<E, R>(T, c, s, a_raw);  // pre-step state, unclipped action (Q3)
#endif
    float alo[E::A], ahi[E::A];
    R a[E::A];
    E::act_bounds(c, alo, ahi);
    o.err = false;
#pragma unroll
    for (int j = 0; j < E::A; ++j) o.err |= visnan(a_raw[j]);
    E::limit_act(c, alo, ahi, a_raw, a);  // Env.limit_act -> act_space.project_to
#ifdef VS_ABLATE_DYNAMICS
    s[0] += a[0] * 1e-6f;
#else
    E::dynamics(T, c, s, h, a, ob);
#endif
    step += 1;
    float slo[E::S], shi[E::S];
    E::state_bounds(c, slo, shi);
    o.failed = false;
    float svv[E::S];
#pragma unroll
    for (int j = 0; j < E::S; ++j) {
        svv[j] = val(s[j]);
        o.err |= isnan(svv[j]);
        // not state_space.contains(s') (Q9, Q10): s < lo or s > hi, NaN compares false as in NumPy
        if (!E::SYMMETRIC_BOX) o.failed |= (svv[j] < slo[j]) | (svv[j] > shi[j]);
    }
    if (E::SYMMETRIC_BOX) o.failed = outside_symmetric_box<E::S>(svv, shi);
    o.done = o.failed | (step >= T.max_steps);
    if (E::FINAL != FINAL_NONE) {
        // FinalRewTask.compute_final_rew, paid once per episode (P/tasks/final_reward.py:130-135)
        if (o.done && !yielded) {
            if (o.failed) {
                if (E::FINAL == FINAL_CONST_MALUS) {
                    o.rew += -1000.0f;  // always_negative, factor 1e3 (:165-174)  [R += float]
                } else {
                    // state- and time-dependent (:215-226): -remaining_steps * |step_rew(s', act = 0)|, remaining_steps as
                    // computed before the step (pysim/base.py:219); 0 for max_steps = inf
                    R zero[E::A];
#pragma unroll
                    for (int j = 0; j < E::A; ++j) zero[j] = 0.f;
                    float remaining = T.max_steps == INT_MAX ? 0.f : (float)(T.max_steps - step);
                    o.rew += -1.0f * remaining * vabs(step_reward<E, R>(T, c, s, zero));
                }
            }
            yielded = true;
        }
    }
    return o;
}

template <class E, bool UNI>
__device__ __forceinline__ void load_consts(const Dev& d, int i, float* c, int first, int last) {
#pragma unroll
    for (int k = 0; k < E::K; ++k)
        if (k >= first && k < last) c[k] = UNI ? d.consts_uni[k] : d.consts[(size_t)k * d.ld + i];
}

// DomainRandomizer.randomize for one lane (domain_parameter.py:104-132): draw -> clamp; writes the raw params
template <class E>
__device__ __forceinline__ void draw_params(const DrSpecs* dr, Rng& g, float* p) {
    int n = dr->n;
    for (int q = 0; q < n; ++q) {
        const vs_dp_spec sp = dr->s[q];
        float v;
        if (sp.kind == VS_DP_NORMAL) v = sp.mean + sp.spread * g.normal();
        else if (sp.kind == VS_DP_UNIFORM) v = g.uniform(sp.mean - sp.spread, sp.mean + sp.spread);
        else v = g.u01() < sp.aux ? sp.spread : sp.mean;  // Bernoulli(prob_1): val_1 with probability prob_1, else val_0
        v = fminf(fmaxf(v, sp.clip_lo), sp.clip_up);
        if (sp.roundint) v = rintf(v);  // torch.round: half to even
#pragma unroll
        for (int k = 0; k < E::P; ++k)
            if (k == sp.param_index) p[k] = v;
    }
}

template <class E>
__device__ __forceinline__ void redraw_lane_params(const Task& T, const Dev& d, const DrSpecs* dr, int i, uint64_t seed,
                                                   uint64_t epi, float* c) {
    float p[E::P];
#pragma unroll
    for (int k = 0; k < E::P; ++k) p[k] = d.params[(size_t)k * d.ld + i];
    Rng gp(seed, d.idx0 + (uint32_t)i, RNG_PARAM, epi);
    draw_params<E>(dr, gp, p);
    E::calc_consts(T, p, c);
#pragma unroll
    for (int k = 0; k < E::P; ++k) d.params[(size_t)k * d.ld + i] = p[k];
#pragma unroll
    for (int k = 0; k < E::K; ++k) d.consts[(size_t)k * d.ld + i] = c[k];
}

// SimPyEnv.reset for one lane with a sampled init state (P/environments/pysim/base.py:166-203), incl. the
// DomainRandWrapperLive redraw (environment_wrappers/domain_randomization.py:141-148) when a randomizer is set.
// Every draw is a pure function of (seed, env index, episode index epi): independent of launch geometry, of how the
// steps are chunked into launches and of hipGraph replay.
template <class E>
__device__ __forceinline__ void reset_lane_sampled(const Task& T, const Dev& d, bool with_dr, int i, uint64_t seed,
                                                   uint64_t epi, float* c, float* s, float* h) {
    if (with_dr && d.dr_n > 0) redraw_lane_params<E>(T, d, &d.drv, i, seed, epi, c);
    if (with_dr && d.pbuf_n > 0) {
        // DomainRandWrapperBuffer.reset (domain_randomization.py:236-251): next set of the ring, or a random one
        uint32_t k;
        if (d.pbuf_mode == 0) k = (uint32_t)(((uint64_t)d.idx0 + (uint64_t)i + epi) % (uint64_t)d.pbuf_n);
        else k = Rng::philox(seed, d.idx0 + (uint32_t)i, RNG_PARAM, epi).x % (uint32_t)d.pbuf_n;
        float p[E::P];
#pragma unroll
        for (int q = 0; q < E::P; ++q) p[q] = d.pbuf[(size_t)q * d.pbuf_n + k];
        E::calc_consts(T, p, c);
#pragma unroll
        for (int q = 0; q < E::P; ++q) d.params[(size_t)q * d.ld + i] = p[q];
#pragma unroll
        for (int q = 0; q < E::K; ++q) d.consts[(size_t)q * d.ld + i] = c[q];
    }
    Rng g(seed, d.idx0 + (uint32_t)i, RNG_INIT, epi);
    float init[E::I];
    E::sample_init(T, c, g, init);
    E::state_from_init(init, s);
    E::init_hidden(T, c, nullptr, s, h, false);
}

// completed-episode append with a wavefront ballot: one atomic per wave, lanes ranked by popcount of the lower mask
__device__ __forceinline__ void append_episode(const Dev& d, bool fin, int i, float ret, int len) {
    unsigned long long m = __builtin_amdgcn_ballot_w64(fin);
    if (m == 0ull) return;
    unsigned lane = __lane_id();
    int leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if ((int)lane == leader) base = atomicAdd(d.ep_count, (unsigned)__popcll(m));
    base = __shfl(base, leader);
    if (fin) {
        unsigned slot = (base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))) % d.ep_cap;
        d.ep_ret[slot] = ret;
        d.ep_len[slot] = len;
        d.ep_env[slot] = i;
    }
}

// per-lane bookkeeping of finished episodes, kept in registers by the kernels
struct EpStat {
    uint32_t epi;    // Dev::ep_idx
    uint32_t count;  // Dev::es_count
    float retsum;    // Dev::es_retsum
    int lensum;      // Dev::es_lensum
};

// auto-reset of the lanes of a wave that finished an episode (wave-uniform early out: most waves have none).
// No memory traffic unless live domain randomisation rewrites the lane's params/constants or the episode log is on.
template <class E, bool UNI>
__device__ __forceinline__ void auto_reset(const Task& T, const Dev& d, bool fin, int i, uint64_t seed, float* c,
                                           float* s, float* h, int& step, float& ret, bool& yielded, EpStat& es) {
    if (__builtin_amdgcn_ballot_w64(fin) == 0ull) return;
    if (d.log_episodes) append_episode(d, fin, i, ret, step);
    if (fin) {
        es.count += 1u;
        es.retsum += ret;
        es.lensum += step;
        load_consts<E, UNI>(d, i, c, E::KS, E::K);  // reset-only constants
        reset_lane_sampled<E>(T, d, !UNI, i, seed, (uint64_t)es.epi, c, s, h);
        es.epi += 1u;
        step = 0;
        ret = 0.f;
        yielded = false;
    }
}

// ------------------------------------------------------------------------------------------------------- step kernel
// vs_step: one fused SimPyEnv.step per lane.  AR = auto-reset of finished lanes inside the same launch.
// No host-side counter enters the kernel: replaying a captured hipGraph of vs_step launches is safe.
// PIPE: the wrapper pipeline (struct Pipe) is compiled in; the default kernels do not carry it.
template <class E, bool UNI, bool AR, bool PIPE = false>
__device__ __forceinline__ void step_body(const Task& T, const Dev& d, const float* __restrict__ act, long env_stride,
                                          long dim_stride, uint64_t seed, int block) {
    int i = block * BLOCK + threadIdx.x;
    if (i >= d.ld) return;
    const size_t ld = d.ld;
    float c[E::K], s[E::S], h[E::H > 0 ? E::H : 1], a[E::A];
    load_consts<E, UNI>(d, i, c, 0, E::KS);
#pragma unroll
    for (int j = 0; j < E::S; ++j) s[j] = d.state[j * ld + i];
#pragma unroll
    for (int j = 0; j < E::H; ++j) h[j] = d.hidden[j * ld + i];
    bool valid = i < d.n;
#pragma unroll
    for (int j = 0; j < E::A; ++j) a[j] = valid ? act[(size_t)i * env_stride + (size_t)j * dim_stride] : 0.f;
    int step = d.step[i];
    bool yielded = E::FINAL != FINAL_NONE ? d.yielded[i] != 0 : false;
    const bool noisy = PIPE && (d.pipe.act_noise | d.pipe.obs_noise);  // wave-uniform
    uint32_t epi = noisy ? d.ep_idx[i] : 0u;

    StepOut o = step_one<E, float>(T, c, s, h, a, step, yielded, (const float*)nullptr, PIPE ? &d : (const Dev*)nullptr,
                                   i, epi);

    float ret = d.ret[i] + o.rew;
    d.rew[i] = o.rew;
    d.done[i] = o.done;
    d.failed[i] = o.failed;
    if (o.err && valid) d.err[i] = 1;  // sticky, write-only

    if (AR) {
        bool fin = o.done && valid;
        if (__builtin_amdgcn_ballot_w64(fin) != 0ull) {  // single-step kernel: the per-env counters are touched by finishing lanes only
            EpStat es{0u, 0u, 0.f, 0};
            if (fin) es = EpStat{d.ep_idx[i], d.es_count[i], d.es_retsum[i], d.es_lensum[i]};
            auto_reset<E, UNI>(T, d, fin, i, seed, c, s, h, step, ret, yielded, es);
            if (fin) {
                d.ep_idx[i] = es.epi;
                d.es_count[i] = es.count;
                d.es_retsum[i] = es.retsum;
                d.es_lensum[i] = es.lensum;
            }
        }
    }

    float ob[E::O];
    E::observe(s, ob);
    if (PIPE && d.pipe.obs_on) {
        if (AR && noisy) epi = d.ep_idx[i];  // a lane that was just reset shows the first observation of its new episode
        pipe_obs<E>(d, i, epi, step, ob, ob);
    }
#pragma unroll
    for (int j = 0; j < E::S; ++j) d.state[j * ld + i] = s[j];
#pragma unroll
    for (int j = 0; j < E::H; ++j) d.hidden[j * ld + i] = h[j];
#pragma unroll
    for (int j = 0; j < E::O; ++j) d.obs[j * ld + i] = ob[j];
    d.step[i] = step;
    d.ret[i] = ret;
    if (E::FINAL != FINAL_NONE) d.yielded[i] = yielded;
}

template <class E, bool UNI, bool AR, bool PIPE>
__global__ __launch_bounds__(BLOCK) void k_step(Task T, Dev d, const float* __restrict__ act, long env_stride,
                                                long dim_stride, uint64_t seed) {
    step_body<E, UNI, AR, PIPE>(T, d, act, env_stride, dim_stride, seed, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------- Jacobian kernel
// vs_step_jac: vs_step plus d(s', r, obs') / d(s, a) by forward-mode differentiation of the very same step code
// (Dual<S+A> instead of float, vecsim_dual.h).  What the fork obtains with torch autograd around its re-implemented
// QCartPole dynamics (P/sampling/rollout.py:836-837, quanser_cartpole.py:233-431) -- here for every family.
// Input x = (s_0 .. s_{S-1}, a_0 .. a_{A-1}); the hidden state (qcp th_ddot, qbb plate angles) is held constant.
// Layouts: jac_s [S][S+A][ld], jac_r [S+A][ld], jac_o [O][S+A][ld].  The step values come from the float path and are
// bit-identical to vs_step.
template <class E, bool UNI>
__global__ __launch_bounds__(BLOCK) void k_step_jac(Task T, Dev d, const float* __restrict__ act, long env_stride,
                                                    long dim_stride) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= d.ld) return;
    constexpr int NI = E::S + E::A;
    using D = Dual<NI>;
    const size_t ld = d.ld;
    float c[E::K];
    load_consts<E, UNI>(d, i, c, 0, E::KS);
    D s[E::S], h[E::H > 0 ? E::H : 1], a[E::A], ob[E::O];
    bool valid = i < d.n;
#pragma unroll
    for (int j = 0; j < E::S; ++j) {
        s[j] = D(d.state[j * ld + i]);
        s[j].d[j] = 1.f;
    }
#pragma unroll
    for (int j = 0; j < E::H; ++j) h[j] = D(d.hidden[j * ld + i]);
#pragma unroll
    for (int j = 0; j < E::A; ++j) {
        a[j] = D(valid ? act[(size_t)i * env_stride + (size_t)j * dim_stride] : 0.f);
        a[j].d[E::S + j] = 1.f;
    }
    int step = d.step[i];
    bool yielded = E::FINAL != FINAL_NONE ? d.yielded[i] != 0 : false;
    // values: the float path, so that they are bit-identical to vs_step (operator-by-operator dual arithmetic cannot
    // reproduce the FMA contraction of the float expressions); tangents: the dual path on the same inputs
    float sf[E::S], hf[E::H > 0 ? E::H : 1], af[E::A], obf[E::O];
#pragma unroll
    for (int j = 0; j < E::S; ++j) sf[j] = s[j].v;
#pragma unroll
    for (int j = 0; j < E::H; ++j) hf[j] = h[j].v;
#pragma unroll
    for (int j = 0; j < E::A; ++j) af[j] = a[j].v;
    int step_d = step;
    bool yielded_d = yielded;
    StepOut of = step_one<E, float>(T, c, sf, hf, af, step, yielded, (const float*)nullptr);
    E::observe(sf, obf);
    StepOutT<D> o = step_one<E, D>(T, c, s, h, a, step_d, yielded_d, (const D*)nullptr);
    E::observe(s, ob);
    d.ret[i] = d.ret[i] + of.rew;
    d.rew[i] = of.rew;
    d.done[i] = of.done;
    d.failed[i] = of.failed;
    if (of.err && valid) d.err[i] = 1;
#pragma unroll
    for (int j = 0; j < E::S; ++j) {
        d.state[j * ld + i] = sf[j];
#pragma unroll
        for (int k = 0; k < NI; ++k) d.jac_s[((size_t)j * NI + k) * ld + i] = s[j].d[k];
    }
#pragma unroll
    for (int j = 0; j < E::H; ++j) d.hidden[j * ld + i] = hf[j];
#pragma unroll
    for (int j = 0; j < E::O; ++j) {
        d.obs[j * ld + i] = obf[j];
#pragma unroll
        for (int k = 0; k < NI; ++k) d.jac_o[((size_t)j * NI + k) * ld + i] = ob[j].d[k];
    }
#pragma unroll
    for (int k = 0; k < NI; ++k) d.jac_r[(size_t)k * ld + i] = o.rew.d[k];
    d.step[i] = step;
    if (E::FINAL != FINAL_NONE) d.yielded[i] = yielded;
}

// ---------------------------------------------------------------------------------------------------- step records
// One env step is recorded as F = O + A + 1 floats  [obs (before the step) | action of the policy | reward].
// A step's records are stored as planes of 4, 2 or 1 floats per env -- F = 4 * NQ + 2 * H2 + H1 -- each plane [ld][w]:
// a lane writes its w floats with ONE dwordx4 / dwordx2 / dword store and a wave writes 64 * 4 * w contiguous bytes.
// QQube: 8 floats = 2 stores instead of 8 (and one address computation instead of eight); no padding for any family.
// Row t of the buffer starts at float offset t * F * ld; plane q at  4 * ld * q  (then the 2-wide, then the 1-wide plane).
template <int F>
struct Planes {
    static constexpr int NQ = F / 4, H2 = (F % 4) >= 2 ? 1 : 0, H1 = F % 2;
    // v[0 .. F) of env i into a row of planes with `ld` envs per plane (global memory or LDS)
    __device__ __forceinline__ static void store(float* __restrict__ row, size_t ld, int i, const float* v) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            reinterpret_cast<float4*>(row + (size_t)q * 4 * ld)[i] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        if (H2) reinterpret_cast<float2*>(row + (size_t)NQ * 4 * ld)[i] = make_float2(v[4 * NQ], v[4 * NQ + 1]);
        if (H1) row[((size_t)NQ * 4 + H2 * 2) * ld + i] = v[F - 1];
    }
    __device__ __forceinline__ static void load(const float* row, size_t ld, int i, float* v) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float4 x = reinterpret_cast<const float4*>(row + (size_t)q * 4 * ld)[i];
            v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
        }
        if (H2) {
            float2 x = reinterpret_cast<const float2*>(row + (size_t)NQ * 4 * ld)[i];
            v[4 * NQ] = x.x, v[4 * NQ + 1] = x.y;
        }
        if (H1) v[F - 1] = row[((size_t)NQ * 4 + H2 * 2) * ld + i];
    }
};
template <class E>
struct Rec {
    static constexpr int F = E::O + E::A + 1;
};
template <class E>
__device__ __forceinline__ void store_record(float* __restrict__ row, size_t ld, int i, const float* ob, const float* a,
                                             float rew) {
    constexpr int F = Rec<E>::F;
    float v[F];
#pragma unroll
    for (int j = 0; j < E::O; ++j) v[j] = ob[j];
#pragma unroll
    for (int j = 0; j < E::A; ++j) v[E::O + j] = a[j];
    v[F - 1] = rew;
    Planes<F>::store(row, ld, i, v);
}

// ---------------------------------------------------------------------------------------------------- rollout kernel
// vs_step_random: rollout() with DummyPolicy (rollout.py:185-239, dummy.py:77-84) -- k env steps per launch, state,
// hidden state and constants stay in registers; only the per-step records stream to HBM when REC.
// Without auto-reset a finished lane freezes (rollout stops at done).
// Actions: Philox4x32-10 keyed by `seed`, counter (env, RNG_ACT, absolute step / SPB); one block feeds SPB = 4 / A
// consecutive steps (the block boundary is wave-uniform because it depends on the launch-global step index only).
template <class E, bool UNI, bool AR, bool REC, bool PIPE = false>
__device__ __forceinline__ void rollout_body(const Task& T, const Dev& d, int k_steps, uint64_t seed, uint64_t reset_seed,
                                             uint64_t epoch0, int block) {
    const size_t rec0 = (size_t)d.traj_t0;  // first record row of this launch (vs_set_traj_offset)
    int i = block * BLOCK + threadIdx.x;
    if (i >= d.ld) return;
    const size_t ld = d.ld;
    bool valid = i < d.n;
    float c[E::K], s[E::S], h[E::H > 0 ? E::H : 1], a[E::A], ob[E::O];
    load_consts<E, UNI>(d, i, c, 0, E::KS);
#pragma unroll
    for (int j = 0; j < E::S; ++j) s[j] = d.state[j * ld + i];
#pragma unroll
    for (int j = 0; j < E::H; ++j) h[j] = d.hidden[j * ld + i];
    int step = d.step[i];
    float ret = d.ret[i];
    bool yielded = E::FINAL != FINAL_NONE ? d.yielded[i] != 0 : false;
    bool frozen = !AR && d.done[i] != 0;
    float rew = d.rew[i];
    bool done = d.done[i] != 0, failed = d.failed[i] != 0;
    EpStat es{d.ep_idx[i], d.es_count[i], d.es_retsum[i], d.es_lensum[i]};
    float alo[E::A], ahi[E::A];
    E::act_bounds(c, alo, ahi);
    constexpr unsigned SPB = 4 / E::A;
    uint4 blk = make_uint4(0, 0, 0, 0);
    if (REC) E::observe(s, ob);
    // gfx950 has ONE vmcnt for loads and stores, in issue order.  Drain the prologue loads here (0x0F70 = vmcnt(0) only) so
    // that the waitcnt pass knows nothing is pending at the loop header: otherwise the conservative `vmcnt(N)` it places
    // at the first in-loop use of a prologue load makes every later iteration wait for its own record stores to land.
    __builtin_amdgcn_s_waitcnt(0x0F70);

    for (int t = 0; t < k_steps; ++t) {
        uint64_t ta = epoch0 + (uint64_t)t;
        unsigned sub = (unsigned)(ta % SPB);
#ifdef VS_ABLATE_RNG
        blk = make_uint4(blk.x + 0x9E3779B9u * (unsigned)i, blk.y + 77u, blk.z + 5u, blk.w + 1u);
#else
        if (t == 0 || sub == 0) blk = Rng::philox(seed, d.idx0 + (uint32_t)i, RNG_ACT, ta / SPB);
#endif
#pragma unroll
        for (int j = 0; j < E::A; ++j) {
            unsigned e = sub * E::A + j;  // wave-uniform element index
            uint32_t bits = e == 0 ? blk.x : e == 1 ? blk.y : e == 2 ? blk.z : blk.w;
            bool nrm = (T.flags & VS_FLAG_ACT_NORM) != 0;  // the policy then acts in the wrapper's space [-1, 1]
            a[j] = E::sample_action(c, nrm ? -1.0f : alo[j], nrm ? 1.0f : ahi[j], Rng::to_u01(bits), j);  // act_space.sample_uniform()
        }
        float ow[E::O];  // the observation as recorded: wrapped when the pipeline is on (the raw one stays in `ob`)
        if (REC) {
            if (PIPE && d.pipe.obs_on) {  // wave-uniform
                pipe_obs<E>(d, i, es.epi, step, ob, ow);
            } else {
#pragma unroll
                for (int j = 0; j < E::O; ++j) ow[j] = ob[j];
            }
        }
        if (!frozen) {
            StepOut o = step_one<E, float>(T, c, s, h, a, step, yielded, REC ? (const float*)ob : (const float*)nullptr,
                                           PIPE ? &d : (const Dev*)nullptr, i, es.epi);
            rew = o.rew;
            done = o.done;
            failed = o.failed;
            ret += o.rew;
            if (o.err && valid) d.err[i] = 1;
        } else {
            rew = 0.f;
        }
        if (REC) {
            store_record<E>(d.traj_rec + (rec0 + (size_t)t) * Rec<E>::F * ld, ld, i, ow, a, rew);
            d.traj_done[(rec0 + (size_t)t) * ld + i] = done;
        }
        bool fin = done && valid && !frozen;
        if (AR) {
            auto_reset<E, UNI>(T, d, fin, i, reset_seed, c, s, h, step, ret, yielded, es);
            if (!UNI) E::act_bounds(c, alo, ahi);  // the action space may depend on redrawn params (omo, bob)
        } else {
            if (fin) {  // rollout() ends here for this lane: book the episode once, then freeze
                es.count += 1u;
                es.retsum += ret;
                es.lensum += step;
            }
            if (d.log_episodes) append_episode(d, fin, i, ret, step);
            frozen |= done;
            // early termination: a wave whose 64 rollouts have all ended has nothing left to do (the ballot is
            // wave-uniform, so the whole wave leaves the loop together); with records on it keeps writing its frozen rows
            if (!REC && __builtin_amdgcn_ballot_w64(!frozen) == 0ull) break;
        }
#ifdef VS_ABLATE_OBSERVE
        if (REC) { for (int j = 0; j < E::O; ++j) ob[j] = s[j % E::S]; }
#else
        if (REC) E::observe(s, ob);
#endif
    }
    if (!REC) E::observe(s, ob);
    if (PIPE && d.pipe.obs_on) pipe_obs<E>(d, i, es.epi, step, ob, ob);
#pragma unroll
    for (int j = 0; j < E::S; ++j) d.state[j * ld + i] = s[j];
#pragma unroll
    for (int j = 0; j < E::H; ++j) d.hidden[j * ld + i] = h[j];
#pragma unroll
    for (int j = 0; j < E::O; ++j) d.obs[j * ld + i] = ob[j];
    d.step[i] = step;
    d.ret[i] = ret;
    d.rew[i] = rew;
    d.done[i] = done;
    d.failed[i] = failed;
    if (E::FINAL != FINAL_NONE) d.yielded[i] = yielded;
    d.ep_idx[i] = es.epi;
    d.es_count[i] = es.count;
    d.es_retsum[i] = es.retsum;
    d.es_lensum[i] = es.lensum;
}

template <class E, bool UNI, bool AR, bool REC, bool PIPE>
__global__ __launch_bounds__(BLOCK) void k_rollout(Task T, Dev d, int k_steps, uint64_t seed, uint64_t reset_seed,
                                                   uint64_t epoch0) {
    rollout_body<E, UNI, AR, REC, PIPE>(T, d, k_steps, seed, reset_seed, epoch0, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------ wave-specialised rollout kernel
// At the size of the headline metric (65 536 envs) k_rollout has exactly one wave per SIMD, and a lone wave issues a VALU
// instruction only every ~7 cycles while the SIMD takes one every 4 from two or more waves (the same
// kernel does 2x the envs in 1.2x the time).  This variant gives every SIMD TWO waves without needing more envs: the step
// of 64 envs is split between a
//   P wave ("physics"): ActNorm -> clip -> dead zone -> integrate -> bounds / done -> auto-reset, observe() of the new state
//                       (owns state, hidden state, step counter)
//   C wave ("critic"):  the policy's action (Philox), reward of (s_t, a_t), final reward, returns / episode statistics,
//                       the record stores
// They exchange through LDS in batches of WS_R env steps, double buffered in both directions: in phase b the P wave
// integrates batch b with the actions C drew in phase b - 1 and leaves (s_t, obs_t, flags_t) per step; C meanwhile works
// off the messages of batch b - 1 and draws the actions of batch b + 1.  ONE workgroup barrier per phase (LDS-only wait: the
// C wave never drains its record stores).  A workgroup is 512 threads = 4 P waves + 4 C waves for 256 envs: the hardware
// places wave w and wave w + 4 of a workgroup on the same SIMD (scratch/ubench/wave_place.hip: 256 of 256 workgroups), so
// every SIMD holds one P and one C wave with about half of the instruction stream each.
// The arithmetic is statement for statement that of step_one / rollout_body (bit-identical results, tested against k_step).
// Not covered here (vs_step_random falls back to k_rollout): live domain randomisation for the families whose C wave reads
// per-env constants (action bounds, c_max: they change at a reset inside the launch), the wrapper pipeline, the
// state-and-time dependent final reward (needs s_{t+1} on the C side).
constexpr int WS_BLOCK = 512, WS_ENVS = 256;
enum : unsigned { WSF_DONE = 1u, WSF_FAILED = 2u, WSF_FROZEN = 4u, WSF_FIN = 8u };

__device__ __forceinline__ void ws_barrier() {
#ifdef VS_WS_NOSYNC  // diagnostic builds only (timing without the exchange; results are wrong)
    return;
#endif
    // LDS traffic only: wait for this wave's LDS ops (lgkmcnt(0)), not for its global stores (a __syncthreads() would also
    // drain vmcnt and stall the C wave on its record stores in every phase)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <class E, bool UNI, bool AR, bool REC, int WS_R>
__global__ __launch_bounds__(WS_BLOCK) void k_rollout_ws(Task T, Dev d, int k_steps, uint64_t seed, uint64_t reset_seed,
                                                         uint64_t epoch0) {
    static_assert(E::FINAL != FINAL_STATE_TIME, "needs the post-step state on the reward side");
    constexpr int M0 = E::S + (REC ? E::O : 0) + 1;  // message of one step: s_t | obs_t (records only) | flags
    constexpr int M = M0 % 4 == 3 ? M0 + 1 : M0;     // 4k + 3 floats would be three LDS ops for the tail; pad to a quad
    __shared__ __attribute__((aligned(16))) float l_msg[2][WS_R][M * WS_ENVS];
    __shared__ __attribute__((aligned(16))) float l_act[2][WS_R][E::A * WS_ENVS];
    const int wave = threadIdx.x >> 6;
    const bool role_c = wave >= WS_BLOCK / 128;
    const int le = threadIdx.x & (WS_ENVS - 1);  // env slot inside the workgroup
    const int i = blockIdx.x * WS_ENVS + le;
    const size_t ld = d.ld;
    const bool valid = i < d.n;
    const int nb = (k_steps + WS_R - 1) / WS_R;
    float c[E::K];
    load_consts<E, UNI>(d, i, c, 0, E::KS);
    float alo[E::A], ahi[E::A];
    E::act_bounds(c, alo, ahi);

    if (!role_c) {
        // ------------------------------------------------------------------------------------------- P wave
        float s[E::S], h[E::H > 0 ? E::H : 1], ob[E::O];
#pragma unroll
        for (int j = 0; j < E::S; ++j) s[j] = d.state[j * ld + i];
#pragma unroll
        for (int j = 0; j < E::H; ++j) h[j] = d.hidden[j * ld + i];
        int step = d.step[i];
        uint32_t epi = d.ep_idx[i];
        bool frozen = !AR && d.done[i] != 0;
        bool done = d.done[i] != 0, failed = d.failed[i] != 0;
        if (REC) E::observe(s, ob);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        ws_barrier();  // the actions of batch 0 are in l_act[0]
#ifdef VS_WS_NOP  // diagnostic: the P wave only keeps the barriers
        for (int b = 0; b < nb; ++b) ws_barrier();
        if (false)
#endif
        for (int b = 0; b < nb; ++b) {
            const int nr = min(WS_R, k_steps - b * WS_R);
            // all actions of the batch up front: one LDS round trip per batch instead of one per step on the critical path
            float a_all[WS_R][E::A];
#pragma unroll
            for (int r = 0; r < WS_R; ++r) Planes<E::A>::load(l_act[b & 1][r], WS_ENVS, le, a_all[r]);
#pragma unroll
            for (int r = 0; r < WS_R; ++r) {
                if (r >= nr) break;
                const float* a = a_all[r];
                float v[M];
#pragma unroll
                for (int j = 0; j < E::S; ++j) v[j] = s[j];
                if (REC) {
#pragma unroll
                    for (int j = 0; j < E::O; ++j) v[E::S + j] = ob[j];
                }
                bool fin = false;  // (used for the reset on this side)
                if (!frozen) {
                    // the P half of step_one: ActNorm -> limit_act -> _step_dynamics -> curr_step += 1 -> is_done
                    float an[E::A], ac[E::A];
                    {
                        const bool nrm = (T.flags & VS_FLAG_ACT_NORM) != 0;
                        float lb[E::A], ub[E::A];
                        E::act_bounds(c, lb, ub);
#pragma unroll
                        for (int j = 0; j < E::A; ++j) {
                            float m = lb[j] + (a[j] + 1.0f) * (ub[j] - lb[j]) * 0.5f;
                            an[j] = vsel(nrm, m, a[j]);
                        }
                    }
                    bool err = false;
#pragma unroll
                    for (int j = 0; j < E::A; ++j) err |= visnan(an[j]);
                    E::limit_act(c, alo, ahi, an, ac);
                    E::dynamics(T, c, s, h, ac, REC ? (const float*)ob : (const float*)nullptr);
                    step += 1;
                    float slo[E::S], shi[E::S];
                    E::state_bounds(c, slo, shi);
                    failed = false;
#pragma unroll
                    for (int j = 0; j < E::S; ++j) {
                        err |= isnan(s[j]);
                        if (!E::SYMMETRIC_BOX) failed |= (s[j] < slo[j]) | (s[j] > shi[j]);
                    }
                    if (E::SYMMETRIC_BOX) failed = outside_symmetric_box<E::S>(s, shi);
                    done = failed | (step >= T.max_steps);
                    if (err && valid) d.err[i] = 1;
                    fin = done && valid;
                }
                unsigned fl = (done ? WSF_DONE : 0u) | (E::FINAL != FINAL_NONE && failed ? WSF_FAILED : 0u) |
                              (!AR && frozen ? WSF_FROZEN : 0u);  // fin = done & !frozen & valid is recomputed by C
                v[M0 - 1] = __uint_as_float(fl);
                if (M > M0) v[M - 1] = 0.f;
                Planes<M>::store(l_msg[b & 1][r], WS_ENVS, le, v);
                if (AR) {
                    if (__builtin_amdgcn_ballot_w64(fin) != 0ull) {
                        if (fin) {
                            load_consts<E, UNI>(d, i, c, E::KS, E::K);
                            // live domain randomisation redraws the lane's parameters here: allowed for the families
                            // whose C wave does not read constants (use_ws)
                            reset_lane_sampled<E>(T, d, !UNI, i, reset_seed, (uint64_t)epi, c, s, h);
                            epi += 1u;
                            step = 0;
                        }
                        if (!UNI) E::act_bounds(c, alo, ahi);
                    }
                } else {
                    frozen |= done;
                }
                if (REC) E::observe(s, ob);
            }
            ws_barrier();
        }
        if (!REC) E::observe(s, ob);
#pragma unroll
        for (int j = 0; j < E::S; ++j) d.state[j * ld + i] = s[j];
#pragma unroll
        for (int j = 0; j < E::H; ++j) d.hidden[j * ld + i] = h[j];
#pragma unroll
        for (int j = 0; j < E::O; ++j) d.obs[j * ld + i] = ob[j];
        d.step[i] = step;
        d.ep_idx[i] = epi;
        d.done[i] = done;
        d.failed[i] = failed;
    } else {
        // ------------------------------------------------------------------------------------------- C wave
        const size_t rec0 = (size_t)d.traj_t0;
        float ret = d.ret[i];
        float rew = d.rew[i];
        bool yielded = E::FINAL != FINAL_NONE ? d.yielded[i] != 0 : false;
        EpStat es{0u, d.es_count[i], d.es_retsum[i], d.es_lensum[i]};
        int len = d.step[i];
        constexpr unsigned SPB = 4 / E::A;
        uint4 blk = make_uint4(0, 0, 0, 0);
        // the actions of batch bb: act_space.sample_uniform() per step, the very stream of k_rollout
        auto draw = [&](int bb) {
            const int nr = min(WS_R, k_steps - bb * WS_R);
#pragma unroll
            for (int r = 0; r < WS_R; ++r) {
                if (r >= nr) break;
                const int t = bb * WS_R + r;
                uint64_t ta = epoch0 + (uint64_t)t;
                unsigned sub = (unsigned)(ta % SPB);
                if (t == 0 || sub == 0) blk = Rng::philox(seed, d.idx0 + (uint32_t)i, RNG_ACT, ta / SPB);
                float a[E::A];
#pragma unroll
                for (int j = 0; j < E::A; ++j) {
                    unsigned e = sub * E::A + j;
                    uint32_t bits = e == 0 ? blk.x : e == 1 ? blk.y : e == 2 ? blk.z : blk.w;
                    bool nrm = (T.flags & VS_FLAG_ACT_NORM) != 0;
                    a[j] = E::sample_action(c, nrm ? -1.0f : alo[j], nrm ? 1.0f : ahi[j], Rng::to_u01(bits), j);
                }
                Planes<E::A>::store(l_act[bb & 1][r], WS_ENVS, le, a);
            }
        };
        // reward, returns and records of the steps of batch bb
        auto work_off = [&](int bb) {
            const int nr = min(WS_R, k_steps - bb * WS_R);
#pragma unroll
            for (int r = 0; r < WS_R; ++r) {
                if (r >= nr) break;
                const int t = bb * WS_R + r;
                float v[M], a[E::A];
                Planes<M>::load(l_msg[bb & 1][r], WS_ENVS, le, v);
                Planes<E::A>::load(l_act[bb & 1][r], WS_ENVS, le, a);
                const unsigned fl = __float_as_uint(v[M0 - 1]);
                const bool was_frozen = !AR ? (fl & WSF_FROZEN) != 0u : false;
                const bool done = (fl & WSF_DONE) != 0u, failed = (fl & WSF_FAILED) != 0u;
                const bool fin = done && !was_frozen && valid;
                if (!was_frozen) {
                    len += 1;  // curr_step of the running episode, counted on this side too
                    float an[E::A];
                    {
                        const bool nrm = (T.flags & VS_FLAG_ACT_NORM) != 0;
                        float lb[E::A], ub[E::A];
                        E::act_bounds(c, lb, ub);
#pragma unroll
                        for (int j = 0; j < E::A; ++j) {
                            float m = lb[j] + (a[j] + 1.0f) * (ub[j] - lb[j]) * 0.5f;
                            an[j] = vsel(nrm, m, a[j]);
                        }
                    }
                    rew = step_reward<E, float>(T, c, v, an);  // pre-step state, unclipped action (Q3)
                    if (E::FINAL == FINAL_CONST_MALUS) {  // once per episode (final_reward.py:130-135, 165-174)
                        if (done && !yielded) {
                            if (failed) rew += -1000.0f;
                            yielded = true;
                        }
                    }
                    ret += rew;
                } else {
                    rew = 0.f;
                }
                if (REC) {
                    store_record<E>(d.traj_rec + (rec0 + (size_t)t) * Rec<E>::F * ld, ld, i, v + E::S, a, rew);
                    d.traj_done[(rec0 + (size_t)t) * ld + i] = done;
                }
                if (__builtin_amdgcn_ballot_w64(fin) != 0ull) {
                    if (d.log_episodes) append_episode(d, fin, i, ret, len);
                    if (fin) {
                        es.count += 1u;
                        es.retsum += ret;
                        es.lensum += len;
                        if (AR) {
                            ret = 0.f;
                            yielded = false;
                            len = 0;
                        }
                    }
                }
            }
        };
        __builtin_amdgcn_s_waitcnt(0x0F70);
        draw(0);
        ws_barrier();
#ifdef VS_WS_NOC  // diagnostic: the C wave only keeps the barriers
        for (int b = 0; b < nb; ++b) ws_barrier();
        if (false)
#endif
        for (int b = 0; b < nb; ++b) {
            if (b >= 1) work_off(b - 1);  // reads l_act[(b - 1) & 1] before draw(b + 1) overwrites the same buffer
            if (b + 1 < nb) draw(b + 1);
            ws_barrier();
        }
        work_off(nb - 1);
        d.ret[i] = ret;
        d.rew[i] = rew;
        if (E::FINAL != FINAL_NONE) d.yielded[i] = yielded;
        d.es_count[i] = es.count;
        d.es_retsum[i] = es.retsum;
        d.es_lensum[i] = es.lensum;
    }
}

// ---------------------------------------------------------------------------------------------------- mixed batches
// BASELINE config 5: several env families in ONE launch.  Lanes are sorted by type (one segment = one ordinary handle),
// a workgroup belongs to exactly one segment, so the type switch is uniform per workgroup and every wavefront takes
// a single branch.  The bodies are the very functions the single-type kernels run: results are bit-identical.
constexpr int MAX_SEG = 5;
struct Seg {
    int type;
    int block_end;  // exclusive prefix of workgroups
    Task T;
    Dev d;
    const float* act;
    long env_stride, dim_stride;
    uint64_t reset_seed, epoch0;
};
struct Segs {
    int n;
    Seg s[MAX_SEG];
};

#define MIXED_DISPATCH(type, ...)                                 \
    switch (type) {                                               \
        case VS_ENV_OMO: { using E = Omo; __VA_ARGS__; } break;     \
        case VS_ENV_BOB: { using E = Bob; __VA_ARGS__; } break;     \
        case VS_ENV_QQ_SU: { using E = QQ; __VA_ARGS__; } break;    \
        case VS_ENV_QCP_SU: { using E = Qcp; __VA_ARGS__; } break;  \
        case VS_ENV_QBB: { using E = Qbb; __VA_ARGS__; } break;     \
        case VS_ENV_QQ_ST: { using E = QQSt; __VA_ARGS__; } break;  \
        case VS_ENV_QCP_ST: { using E = QcpSt; __VA_ARGS__; } break;\
        case VS_ENV_PEND: { using E = Pend; __VA_ARGS__; } break;   \
        default: { using E = BobD; __VA_ARGS__; } break;            \
    }

template <bool AR, bool REC>
__global__ __launch_bounds__(BLOCK) void k_rollout_mixed(const Segs* __restrict__ segs, int k_steps, uint64_t seed) {
    int b = blockIdx.x, q = 0, first = 0;
    int n = segs->n;
    while (q < n - 1 && b >= segs->s[q].block_end) first = segs->s[q++].block_end;
    const Seg& sg = segs->s[q];
    MIXED_DISPATCH(sg.type, (rollout_body<E, false, AR, REC>(sg.T, sg.d, k_steps, seed, sg.reset_seed, sg.epoch0, b - first)));
}

template <bool AR>
__global__ __launch_bounds__(BLOCK) void k_step_mixed(const Segs* __restrict__ segs) {
    int b = blockIdx.x, q = 0, first = 0;
    int n = segs->n;
    while (q < n - 1 && b >= segs->s[q].block_end) first = segs->s[q++].block_end;
    const Seg& sg = segs->s[q];
    MIXED_DISPATCH(sg.type, (step_body<E, false, AR>(sg.T, sg.d, sg.act, sg.env_stride, sg.dim_stride, sg.reset_seed, b - first)));
}

// -------------------------------------------------------------------------------------------- params / reset kernels
// domain_param setter (P/environments/pysim/base.py:112-124): _calc_constants + spaces + task.reset for masked lanes.
// src == nullptr: recompute from the stored params; bcast: src is one [P] vector for every lane.
template <class E>
__global__ __launch_bounds__(BLOCK) void k_set_params(Task T, Dev d, const float* __restrict__ src, long pitch,
                                                      int bcast, const uint8_t* __restrict__ mask) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= d.ld) return;
    if (mask && (i >= d.n || mask[i] == 0)) return;
    int is = i < d.n ? i : d.n - 1;  // padding lanes mirror the last env (keeps them finite)
    float p[E::P], c[E::K];
#pragma unroll
    for (int k = 0; k < E::P; ++k)
        p[k] = src ? (bcast ? src[k] : src[(size_t)k * pitch + is]) : d.params[(size_t)k * d.ld + is];
    E::calc_consts(T, p, c);
#pragma unroll
    for (int k = 0; k < E::P; ++k) d.params[(size_t)k * d.ld + i] = p[k];
#pragma unroll
    for (int k = 0; k < E::K; ++k) d.consts[(size_t)k * d.ld + i] = c[k];
    if (bcast && i == 0) {
#pragma unroll
        for (int k = 0; k < E::K; ++k) d.consts_uni[k] = c[k];
    }
}

// DomainRandomizer.randomize(N) + get_params on device; `specs` is a device copy of the spec list
template <class E>
__global__ __launch_bounds__(BLOCK) void k_sample_params(Task T, Dev d, const DrSpecs* __restrict__ specs,
                                                         uint64_t seed, const uint8_t* __restrict__ mask) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= d.n) return;
    if (mask && mask[i] == 0) return;
    float c[E::K];
    redraw_lane_params<E>(T, d, specs, i, seed, 0ull, c);
}

// SimPyEnv.reset (P/environments/pysim/base.py:166-203) for masked lanes; init == nullptr samples the init space.
// The draws of an explicit reset use episode index 0: reset(seed) is a pure function of (seed, env index), like
// pyrado.set_seed(seed) followed by env.reset() in the reference.
template <class E>
__global__ __launch_bounds__(BLOCK) void k_reset(Task T, Dev d, const float* __restrict__ init, long pitch,
                                                 int full_state, const uint8_t* __restrict__ mask, uint64_t seed) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= d.ld) return;
    bool valid = i < d.n;
    if (mask && (!valid || mask[i] == 0)) return;
    const size_t ld = d.ld;
    float c[E::K], s[E::S], h[E::H > 0 ? E::H : 1], ob[E::O];
    load_consts<E, false>(d, i, c, 0, E::K);
    if (init == nullptr || !valid) {
        reset_lane_sampled<E>(T, d, valid, i, seed, 0ull, c, s, h);
    } else {
        // DomainRandWrapperLive / Buffer .reset with an explicit init_state still redraws the params
        if (d.dr_n > 0 || d.pbuf_n > 0) {
            float s_tmp[E::S], h_tmp[E::H > 0 ? E::H : 1];
            reset_lane_sampled<E>(T, d, true, i, seed, 0ull, c, s_tmp, h_tmp);  // params + constants (state discarded)
        }
        if (full_state) {
#pragma unroll
            for (int j = 0; j < E::S; ++j) s[j] = init[(size_t)j * pitch + i];  // copied verbatim (base.py:184-188)
        } else {
            float in[E::I];
#pragma unroll
            for (int j = 0; j < E::I; ++j) in[j] = init[(size_t)j * pitch + i];
            E::state_from_init(in, s);
        }
        float p[E::P];
#pragma unroll
        for (int k = 0; k < E::P; ++k) p[k] = d.params[(size_t)k * ld + i];
        E::init_hidden(T, c, p, s, h, full_state != 0);
    }
    E::observe(s, ob);
    if (d.pipe.obs_on) pipe_obs<E>(d, i, 1u, 0, ob, ob);  // EnvWrapperObs.reset processes the first observation too
#pragma unroll
    for (int j = 0; j < E::S; ++j) d.state[j * ld + i] = s[j];
#pragma unroll
    for (int j = 0; j < E::H; ++j) d.hidden[j * ld + i] = h[j];
#pragma unroll
    for (int j = 0; j < E::O; ++j) d.obs[j * ld + i] = ob[j];
    d.step[i] = 0;
    d.ret[i] = 0.f;
    d.rew[i] = 0.f;
    d.done[i] = 0;
    d.failed[i] = 0;
    d.err[i] = 0;
    d.yielded[i] = 0;
    d.ep_idx[i] = 1u;
}

// re-derive VS_OBS from VS_STATE after a host-side `state` assignment
template <class E>
__global__ __launch_bounds__(BLOCK) void k_observe(Dev d) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= d.ld) return;
    float s[E::S], ob[E::O];
#pragma unroll
    for (int j = 0; j < E::S; ++j) s[j] = d.state[(size_t)j * d.ld + i];
    E::observe(s, ob);
    if (d.pipe.obs_on) pipe_obs<E>(d, i, d.ep_idx[i], d.step[i], ob, ob);
#pragma unroll
    for (int j = 0; j < E::O; ++j) d.obs[(size_t)j * d.ld + i] = ob[j];
}

__global__ void k_count_err(const uint8_t* err, int n, unsigned long long* out) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    bool e = i < n && err[i] != 0;
    unsigned long long m = __builtin_amdgcn_ballot_w64(e);
    if (__lane_id() == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}

__global__ void k_fill4(float4* __restrict__ dst, size_t n4, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const float4 x = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
    for (; i < n4; i += stride) dst[i] = x;
}

__global__ void k_copy4(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) dst[i] = src[i];
}

}  // namespace vs

// ====================================================================================================================
// host side
// ====================================================================================================================
using namespace vs;

struct EnvInfo {
    const char* name;
    int S, A, O, P, H, I, K;
    const char* pnames[MAXP];
    float nominal[MAXP];
    float des[MAXS], qd[MAXS], rd[MAXA];
};

// names/nominal values: get_nominal_domain_param of each env; task defaults: _create_task of each env
static const EnvInfo ENV_INFO[VS_ENV_COUNT] = {
    {"omo", Omo::S, Omo::A, Omo::O, Omo::P, Omo::H, Omo::I, Omo::K,
     {"mass", "stiffness", "damping"},
     {1.0f, 30.0f, 0.5f},  // one_mass_oscillator.py:82-86
     {0, 0}, {1e1f, 1e-2f}, {1e-6f}},  // :70-73
    {"bob", Bob::S, Bob::A, Bob::O, Bob::P, Bob::H, Bob::I, Bob::K,
     {"gravity_const", "ball_mass", "ball_radius", "beam_mass", "beam_length", "beam_thickness", "friction_coeff",
      "ang_offset"},
     {9.81f, 0.5f, 0.1f, 3.0f, 2.0f, 0.1f, 0.05f, 0.0f},  // ball_on_beam.py:77-87
     {0, 0, 0, 0}, {1e5f, 1e3f, 1e3f, 1e2f}, {1.0f}},  // :100-108
    {"qq-su", QQ::S, QQ::A, QQ::O, QQ::P, QQ::H, QQ::I, QQ::K,
     {"gravity_const", "motor_resistance", "motor_back_emf", "mass_rot_pole", "length_rot_pole", "damping_rot_pole",
      "mass_pend_pole", "length_pend_pole", "damping_pend_pole", "voltage_thold_neg", "voltage_thold_pos"},
     {9.81f, 8.4f, 0.042f, 0.095f, 0.085f, 5e-6f, 0.024f, 0.129f, 1e-6f, 0.0f, 0.0f},  // quanser_qube.py:54-68
     {0.0f, PI_F, 0.0f, 0.0f}, {1.0f, 1.0f, 2e-2f, 5e-3f}, {4e-3f}},  // :181-188
    {"qcp-su", Qcp::S, Qcp::A, Qcp::O, Qcp::P, Qcp::H, Qcp::I, Qcp::K,
     {"gravity_const", "cart_mass", "rail_length", "motor_efficiency", "gear_efficiency", "gear_ratio",
      "motor_inertia", "pinion_radius", "motor_resistance", "motor_back_emf", "pole_damping", "combined_damping",
      "pole_mass", "pole_length", "cart_friction_coeff", "voltage_thold_neg", "voltage_thold_pos"},
     {9.81f, 0.58f, 0.814f, 0.9f, 0.9f, 3.71f, 3.9e-7f, 6.35e-3f, 2.6f, 7.67e-3f, 0.0024f, 5.4f, 0.127f,
      0.3365f / 2, 0.02f, 0.0f, 0.0f},  // quanser_cartpole.py:111-143
     {0.0f, PI_F, 0.0f, 0.0f}, {3e-1f, 5e-1f, 5e-3f, 1e-3f}, {1e-3f}},  // :573-587
    {"qbb", Qbb::S, Qbb::A, Qbb::O, Qbb::P, Qbb::H, Qbb::I, Qbb::K,
     {"gravity_const", "ball_mass", "ball_radius", "plate_length", "arm_radius", "gear_ratio", "gear_efficiency",
      "load_inertia", "motor_inertia", "motor_back_emf", "motor_resistance", "motor_efficiency", "combined_damping",
      "ball_damping", "voltage_thold_x_pos", "voltage_thold_x_neg", "voltage_thold_y_pos", "voltage_thold_y_neg",
      "offset_th_x", "offset_th_y"},
     {9.81f, 0.003f, 0.019625f, 0.275f, 0.0254f, 70.0f, 0.9f, 5.2822e-5f, 4.6063e-7f, 0.0077f, 2.6f, 0.69f, 0.015f,
      0.05f, 0.28f, -0.10f, 0.28f, -0.074f, 0.0f, 0.0f},  // quanser_ball_balancer.py:141-143,171-202
     {0, 0, 0, 0, 0, 0, 0, 0}, {1e0f, 1e0f, 5e3f, 5e3f, 1e-2f, 1e-2f, 5e-1f, 5e-1f}, {1e-2f, 1e-2f}},  // :119-129
    // ---- the remaining pysim families ----
    {"qq-st", QQSt::S, QQSt::A, QQSt::O, QQSt::P, QQSt::H, QQSt::I, QQSt::K,
     {"gravity_const", "motor_resistance", "motor_back_emf", "mass_rot_pole", "length_rot_pole", "damping_rot_pole",
      "mass_pend_pole", "length_pend_pole", "damping_pend_pole", "voltage_thold_neg", "voltage_thold_pos"},
     {9.81f, 8.4f, 0.042f, 0.095f, 0.085f, 5e-6f, 0.024f, 0.129f, 1e-6f, 0.0f, 0.0f},
     {0.0f, PI_F, 0.0f, 0.0f}, {3.0f, 4.0f, 2.0f, 2.0f}, {5e-2f}},  // quanser_qube.py:215-222
    {"qcp-st", QcpSt::S, QcpSt::A, QcpSt::O, QcpSt::P, QcpSt::H, QcpSt::I, QcpSt::K,
     {"gravity_const", "cart_mass", "rail_length", "motor_efficiency", "gear_efficiency", "gear_ratio",
      "motor_inertia", "pinion_radius", "motor_resistance", "motor_back_emf", "pole_damping", "combined_damping",
      "pole_mass", "pole_length", "cart_friction_coeff", "voltage_thold_neg", "voltage_thold_pos"},
     {9.81f, 0.58f, 0.814f, 0.9f, 0.9f, 3.71f, 3.9e-7f, 6.35e-3f, 2.6f, 7.67e-3f, 0.0024f, 5.4f, 0.127f,
      0.3365f / 2, 0.02f, 0.0f, 0.0f},
     {0.0f, PI_F, 0.0f, 0.0f}, {5e-0f, 1e1f, 1e-2f, 1e-2f}, {1e-3f}},  // quanser_cartpole.py:494-504
    {"pend", Pend::S, Pend::A, Pend::O, Pend::P, Pend::H, Pend::I, Pend::K,
     {"gravity_const", "pole_mass", "pole_length", "pole_damping", "torque_thold"},
     {9.81f, 1.0f, 1.0f, 0.05f, 3.5f},  // pendulum.py:94-101
     {PI_F, 0.0f}, {1e-0f, 1e-3f}, {1e-2f}},  // :82-87
    {"bob-d", BobD::S, BobD::A, BobD::O, BobD::P, BobD::H, BobD::I, BobD::K,
     {"gravity_const", "ball_mass", "ball_radius", "beam_mass", "beam_length", "beam_thickness", "friction_coeff",
      "ang_offset"},
     {9.81f, 0.5f, 0.1f, 3.0f, 2.0f, 0.1f, 0.05f, 0.0f},
     {0, 0, 0, 0}, {1e5f, 1e3f, 1e3f, 1e2f}, {1.0f}}};

struct vs_env {
    int type = 0;
    int device = 0;
    Task task{};
    DrSpecs dr{};                 // host copy of the live randomizer
    DrSpecs* d_specs = nullptr;   // device scratch for vs_sample_params
    float* d_pbuf = nullptr;      // DomainRandWrapperBuffer parameter sets
    float* d_ring = nullptr;      // ActDelayWrapper ring (Pipe::ring)
    int rollout_variant = -1;     // vs_set_rollout_variant: -1 automatic, 0 k_rollout, 1 k_rollout_ws
    int n_cu = 256;               // compute units of the device (256 on MI355X)
    bool auto_reset = false;
    uint64_t ar_seed = 0;
    bool uniform = true;
    hipStream_t own_stream = nullptr, stream = nullptr;
    Dev d{};
    int traj_cap = 0;
    uint64_t epoch = 0;  // absolute step index of the action stream of vs_step_random
    std::string err;
    std::vector<void*> allocs;
    void* stage = nullptr;
    size_t stage_bytes = 0;
    void* stage_mask = nullptr;
    unsigned long long* d_counter = nullptr;
};

static thread_local std::string g_create_err;

static int fail(vs_handle h, int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    else
        snprintf(buf, sizeof buf, "%s", what);
    if (h) h->err = buf;
    else g_create_err = buf;
    return code;
}

#define HIPCHK(h, x)                                              \
    do {                                                          \
        hipError_t e_ = (x);                                      \
        if (e_ != hipSuccess) return fail(h, VS_ERR_HIP, #x, e_); \
    } while (0)

#define DISPATCH_ENV(type, ...)                                  \
    switch (type) {                                              \
        case VS_ENV_OMO: { using E = Omo; __VA_ARGS__; } break;    \
        case VS_ENV_BOB: { using E = Bob; __VA_ARGS__; } break;    \
        case VS_ENV_QQ_SU: { using E = QQ; __VA_ARGS__; } break;   \
        case VS_ENV_QCP_SU: { using E = Qcp; __VA_ARGS__; } break; \
        case VS_ENV_QBB: { using E = Qbb; __VA_ARGS__; } break;    \
        case VS_ENV_QQ_ST: { using E = QQSt; __VA_ARGS__; } break; \
        case VS_ENV_QCP_ST: { using E = QcpSt; __VA_ARGS__; } break; \
        case VS_ENV_PEND: { using E = Pend; __VA_ARGS__; } break;  \
        case VS_ENV_BOB_D: { using E = BobD; __VA_ARGS__; } break; \
        default: break;                                          \
    }

static inline dim3 grid_for(int ld) { return dim3((unsigned)((ld + BLOCK - 1) / BLOCK)); }

static bool is_device_ptr(const void* p) {
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear: plain host memory is reported as an error
        return false;
    }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

template <class T>
static int dalloc(vs_handle h, T** p, size_t count) {
    void* q = nullptr;
    HIPCHK(h, hipMalloc(&q, count * sizeof(T) > 0 ? count * sizeof(T) : 4));
    HIPCHK(h, hipMemsetAsync(q, 0, count * sizeof(T) > 0 ? count * sizeof(T) : 4, h->stream));
    h->allocs.push_back(q);
    *p = (T*)q;
    return VS_OK;
}

// stage a host SoA [rows][pitch] into device memory [rows][ld]; device inputs are used in place
static int stage_rows(vs_handle h, const float* src, int rows, int64_t pitch, const float** out, long* out_pitch) {
    if (is_device_ptr(src)) {
        *out = src;
        *out_pitch = (long)pitch;
        return VS_OK;
    }
    size_t need = (size_t)rows * h->d.ld * sizeof(float);
    if (need > h->stage_bytes) {
        if (h->stage) HIPCHK(h, hipFree(h->stage));
        h->stage = nullptr;
        h->stage_bytes = 0;
        HIPCHK(h, hipMalloc(&h->stage, need));
        h->stage_bytes = need;
    }
    HIPCHK(h, hipMemcpy2DAsync(h->stage, (size_t)h->d.ld * 4, src, (size_t)pitch * 4, (size_t)h->d.n * 4, rows,
                               hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // control path: the caller's host buffer may be a temporary
    *out = (const float*)h->stage;
    *out_pitch = h->d.ld;
    return VS_OK;
}

static int stage_mask(vs_handle h, const uint8_t* mask, const uint8_t** out) {
    if (!mask) { *out = nullptr; return VS_OK; }
    if (is_device_ptr(mask)) { *out = mask; return VS_OK; }
    if (!h->stage_mask) HIPCHK(h, hipMalloc(&h->stage_mask, (size_t)h->d.ld));
    HIPCHK(h, hipMemcpyAsync(h->stage_mask, mask, (size_t)h->d.n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *out = (const uint8_t*)h->stage_mask;
    return VS_OK;
}

static int check_specs(vs_handle h, const vs_dp_spec* specs, int n, DrSpecs* out) {
    const EnvInfo& ei = ENV_INFO[h->type];
    if (n < 0 || n > MAXP || (n > 0 && !specs)) return fail(h, VS_ERR_ARG, "bad domain-parameter spec list");
    out->n = n;
    for (int q = 0; q < n; ++q) {
        if (specs[q].param_index < 0 || specs[q].param_index >= ei.P) return fail(h, VS_ERR_ARG, "spec param_index out of range");
        const int kind = specs[q].kind;
        if (kind != VS_DP_NORMAL && kind != VS_DP_UNIFORM && kind != VS_DP_BERNOULLI) return fail(h, VS_ERR_ARG, "spec kind must be VS_DP_NORMAL, VS_DP_UNIFORM or VS_DP_BERNOULLI");
        if (kind != VS_DP_BERNOULLI && !(specs[q].spread >= 0.f)) return fail(h, VS_ERR_ARG, "spec spread must be >= 0");
        if (kind == VS_DP_BERNOULLI && !(specs[q].aux >= 0.f && specs[q].aux <= 1.f)) return fail(h, VS_ERR_ARG, "spec aux (prob_1) must be in [0, 1]");
        out->s[q] = specs[q];
    }
    return VS_OK;
}

template <class E>
static void launch_step(vs_handle h, const float* act, long es, long ds) {
    dim3 g = grid_for(h->d.ld), b(BLOCK);
    bool uni = h->uniform && h->dr.n == 0 && h->d.pbuf_n == 0;
#define LS(U, AR, PI) hipLaunchKernelGGL((k_step<E, U, AR, PI>), g, b, 0, h->stream, h->task, h->d, act, es, ds, h->ar_seed)
    if (h->d.pipe.act_on || h->d.pipe.obs_on) {  // the wrapper pipeline: per-env-constant variant only
        if (h->auto_reset) LS(false, true, true); else LS(false, false, true);
    } else if (h->auto_reset) { if (uni) LS(true, true, false); else LS(false, true, false); }
    else { if (uni) LS(true, false, false); else LS(false, false, false); }
#undef LS
}

// The wave-specialised kernel pays while k_rollout would leave a SIMD with a single wave (up to 256 envs per compute unit),
// for the families whose step splits into two comparable halves (E::WS_PAYS), and needs constants that do not change
// inside the launch.  VS_ROLLOUT_VARIANT=plain|ws overrides for every handle (experiments).
template <class E>
static bool use_ws(vs_handle h) {
    if (E::FINAL == FINAL_STATE_TIME) return false;
    if (h->d.pipe.act_on || h->d.pipe.obs_on) return false;
    const bool live = h->dr.n > 0 || h->d.pbuf_n > 0;
    if (live && E::REWARD_SIDE_USES_CONSTS) return false;
    if (h->rollout_variant >= 0) return h->rollout_variant == 1;
    if (live && !E::WS_WITH_LIVE_DR) return false;
    static const char* force = getenv("VS_ROLLOUT_VARIANT");
    if (force && force[0] == 'p') return false;
    if (force && force[0] == 'w') return true;
    // one 256-env workgroup per CU at most: a CU that gets a second one runs four waves per SIMD and the launch waits for it
    // (73 728 envs with records: 100 us against k_rollout's 79 us; at 65 536: 55 against 69)
    return E::WS_PAYS && h->d.ld <= (int64_t)WS_ENVS * h->n_cu;
}

template <class E>
static void launch_rollout(vs_handle h, int k, uint64_t seed, uint64_t ep, bool rec) {
    bool uni = h->uniform && h->dr.n == 0;
    if constexpr (E::FINAL != FINAL_STATE_TIME) {
        if (use_ws<E>(h)) {
            dim3 g((unsigned)(h->d.ld / WS_ENVS)), b(WS_BLOCK);
            // R = 4 steps per exchange (measured on the headline config: R = 1 / 2 / 4 -> 68.7 / 64.8 / 62.0 us per 100 steps)
#define LW(U, AR, R) hipLaunchKernelGGL((k_rollout_ws<E, U, AR, R, 4>), g, b, 0, h->stream, h->task, h->d, k, seed, h->ar_seed, ep)
            if (uni) {
                if (h->auto_reset) { if (rec) LW(true, true, true); else LW(true, true, false); }
                else { if (rec) LW(true, false, true); else LW(true, false, false); }
            } else {
                if (h->auto_reset) { if (rec) LW(false, true, true); else LW(false, true, false); }
                else { if (rec) LW(false, false, true); else LW(false, false, false); }
            }
#undef LW
            return;
        }
    }
    dim3 g = grid_for(h->d.ld), b(BLOCK);
#define LR(U, AR, R) hipLaunchKernelGGL((k_rollout<E, U, AR, R, false>), g, b, 0, h->stream, h->task, h->d, k, seed, h->ar_seed, ep)
#define LP(AR, R) hipLaunchKernelGGL((k_rollout<E, false, AR, R, true>), g, b, 0, h->stream, h->task, h->d, k, seed, h->ar_seed, ep)
    if (h->d.pipe.act_on || h->d.pipe.obs_on) {
        if (h->auto_reset) { if (rec) LP(true, true); else LP(true, false); }
        else { if (rec) LP(false, true); else LP(false, false); }
    } else if (uni) {
        if (h->auto_reset) { if (rec) LR(true, true, true); else LR(true, true, false); }
        else { if (rec) LR(true, false, true); else LR(true, false, false); }
    } else {
        if (h->auto_reset) { if (rec) LR(false, true, true); else LR(false, true, false); }
        else { if (rec) LR(false, false, true); else LR(false, false, false); }
    }
#undef LR
#undef LP
}

struct vs_mixed {
    int n = 0;
    vs_handle sub[MAX_SEG]{};
    Segs host{};
    Segs* dev = nullptr;
    int total_blocks = 0;
    std::string err;
};

static int mixed_upload(vs_mixed* m, const float* const* acts, const int64_t* env_strides, const int64_t* dim_strides,
                        int k_steps) {
    int blocks = 0;
    vs_handle h0 = m->sub[0];
    for (int q = 0; q < m->n; ++q) {
        vs_handle h = m->sub[q];
        if (h->d.pipe.act_on || h->d.pipe.obs_on) {
            m->err = "mixed batch: a segment carries an action/observation pipeline (single-family handles only)";
            return VS_ERR_STATE;
        }
        Seg& sg = m->host.s[q];
        blocks += (h->d.ld + BLOCK - 1) / BLOCK;
        sg.type = h->type;
        sg.block_end = blocks;
        sg.T = h->task;
        sg.d = h->d;
        sg.act = acts ? acts[q] : nullptr;
        sg.env_stride = env_strides ? (long)env_strides[q] : 0;
        sg.dim_stride = dim_strides ? (long)dim_strides[q] : 0;
        sg.reset_seed = h->ar_seed;
        sg.epoch0 = h->epoch;
        h->epoch += (uint64_t)k_steps;
    }
    m->host.n = m->n;
    m->total_blocks = blocks;
    // the segment table travels through device memory (kernel arguments are capped at 4 KB); hipMemcpyAsync from the
    // pageable host copy is ordered on the stream before the launch that reads it
    hipError_t e = hipMemcpyAsync(m->dev, &m->host, sizeof(Segs), hipMemcpyHostToDevice, h0->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h0->stream);  // host.s is rewritten by the next call
    if (e != hipSuccess) { m->err = std::string("mixed_upload: ") + hipGetErrorString(e); return VS_ERR_HIP; }
    return VS_OK;
}

extern "C" {

int vs_version(void) { return 130; }

int vs_traj_layout(int t, int* F, int* nq, int* h2, int* h1) {
    if (t < 0 || t >= VS_ENV_COUNT) return VS_ERR_ARG;
    const int f = ENV_INFO[t].O + ENV_INFO[t].A + 1;
    if (F) *F = f;
    if (nq) *nq = f / 4;
    if (h2) *h2 = (f % 4) >= 2 ? 1 : 0;
    if (h1) *h1 = f % 2;
    return VS_OK;
}

int vs_env_dims(int t, int* S, int* A, int* O, int* P, int* H, int* I, int* K) {
    if (t < 0 || t >= VS_ENV_COUNT) return VS_ERR_ARG;
    const EnvInfo& e = ENV_INFO[t];
    if (S) *S = e.S;
    if (A) *A = e.A;
    if (O) *O = e.O;
    if (P) *P = e.P;
    if (H) *H = e.H;
    if (I) *I = e.I;
    if (K) *K = e.K;
    return VS_OK;
}

const char* vs_env_name(int t) { return (t < 0 || t >= VS_ENV_COUNT) ? nullptr : ENV_INFO[t].name; }

const char* vs_param_name(int t, int i) {
    if (t < 0 || t >= VS_ENV_COUNT || i < 0 || i >= ENV_INFO[t].P) return nullptr;
    return ENV_INFO[t].pnames[i];
}

int vs_nominal_params(int t, int flags, float* out) {
    if (t < 0 || t >= VS_ENV_COUNT || !out) return VS_ERR_ARG;
    for (int k = 0; k < ENV_INFO[t].P; ++k) out[k] = ENV_INFO[t].nominal[k];
    if ((t == VS_ENV_QCP_SU || t == VS_ENV_QCP_ST) && (flags & VS_FLAG_LONG_POLE)) {  // get_nominal_domain_param(long=True), :113-118
        out[12] = 0.23f;
        out[13] = 0.641f / 2;
    }
    return VS_OK;
}

const char* vs_last_error(vs_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int vs_create(int env_type, int64_t n_envs, double dt, int64_t max_steps, int device_id, const vs_task_cfg* cfg,
              vs_handle* out) {
    if (!out) return fail(nullptr, VS_ERR_ARG, "vs_create: out is NULL");
    *out = nullptr;
    if (env_type < 0 || env_type >= VS_ENV_COUNT) return fail(nullptr, VS_ERR_ARG, "vs_create: unknown env_type");
    if (n_envs < 1 || n_envs > (1LL << 30)) return fail(nullptr, VS_ERR_ARG, "vs_create: n_envs must be in [1, 2^30]");
    if (!(dt >= 0.0)) return fail(nullptr, VS_ERR_ARG, "vs_create: dt must be >= 0");  // Env.__init__ base.py:58-59
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(nullptr, VS_ERR_HIP, "vs_create: no HIP device available (libvecsim has no CPU fallback)", e);
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, VS_ERR_ARG, "vs_create: bad device_id");
    vs_handle h = new (std::nothrow) vs_env();
    if (!h) return fail(nullptr, VS_ERR_HIP, "vs_create: out of host memory");
    h->type = env_type;
    h->device = device_id;
    {
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cu > 0) h->n_cu = cu;
    }
    const EnvInfo& ei = ENV_INFO[env_type];
    Task& T = h->task;
    bool defaults = !cfg || cfg->use_defaults;
    for (int j = 0; j < MAXS; ++j) {
        T.des[j] = defaults ? ei.des[j] : cfg->state_des[j];
        T.qd[j] = defaults ? ei.qd[j] : cfg->q_diag[j];
    }
    for (int j = 0; j < MAXA; ++j) T.rd[j] = defaults ? ei.rd[j] : cfg->r_diag[j];
    T.dt = (float)dt;
    T.max_steps = (max_steps <= 0 || max_steps >= INT_MAX) ? INT_MAX : (int)max_steps;
    // without a cfg the ctor defaults of the reference apply: QCartPoleStabSim(long=True, simple_dynamics=True)
    T.flags = cfg ? cfg->flags : (env_type == VS_ENV_QCP_ST ? (VS_FLAG_LONG_POLE | VS_FLAG_SIMPLE_DYNAMICS) : 0);
    T.wild_init = cfg ? cfg->wild_init : 0;
    for (int j = 0; j < MAXS; ++j) T.init_fixed[j] = cfg ? cfg->init_state[j] : 0.f;
    int rc = VS_OK;
#define CK(x) do { rc = (x); if (rc != VS_OK) { g_create_err = h->err; vs_destroy(h); return rc; } } while (0)
#define HK(x) do { hipError_t e2 = (x); if (e2 != hipSuccess) { fail(nullptr, VS_ERR_HIP, #x, e2); vs_destroy(h); return VS_ERR_HIP; } } while (0)
    HK(hipSetDevice(device_id));
    // a BLOCKING stream: it orders itself with the legacy default stream, which is where torch (and most callers) run
    // unless told otherwise -- zero-copy views of the handle's buffers can then be read by default-stream work without
    // an explicit sync.  Callers on other non-blocking streams hand theirs over with vs_set_stream.
    HK(hipStreamCreateWithFlags(&h->own_stream, hipStreamDefault));
    h->stream = h->own_stream;
    Dev& d = h->d;
    d.n = (int)n_envs;
    d.ld = (int)(((n_envs + BLOCK - 1) / BLOCK) * BLOCK);
    size_t ld = d.ld;
    CK(dalloc(h, &d.state, ei.S * ld));
    CK(dalloc(h, &d.hidden, (ei.H > 0 ? ei.H : 1) * ld));
    CK(dalloc(h, &d.obs, ei.O * ld));
    CK(dalloc(h, &d.rew, ld));
    CK(dalloc(h, &d.ret, ld));
    CK(dalloc(h, &d.consts, ei.K * ld));
    CK(dalloc(h, &d.params, ei.P * ld));
    CK(dalloc(h, &d.consts_uni, (size_t)MAXK));
    CK(dalloc(h, &d.done, ld));
    CK(dalloc(h, &d.failed, ld));
    CK(dalloc(h, &d.err, ld));
    CK(dalloc(h, &d.yielded, ld));
    CK(dalloc(h, &d.step, ld));
    CK(dalloc(h, &d.ep_idx, ld));
    CK(dalloc(h, &d.es_count, ld));
    CK(dalloc(h, &d.es_retsum, ld));
    CK(dalloc(h, &d.es_lensum, ld));
    CK(dalloc(h, &h->d_specs, (size_t)1));
    d.ep_cap = (unsigned)(ld < (1u << 16) ? (1u << 16) : ld);
    CK(dalloc(h, &d.ep_ret, (size_t)d.ep_cap));
    CK(dalloc(h, &d.ep_len, (size_t)d.ep_cap));
    CK(dalloc(h, &d.ep_env, (size_t)d.ep_cap));
    CK(dalloc(h, &d.ep_count, (size_t)1));
    CK(dalloc(h, &h->d_counter, (size_t)1));
    float nominal[MAXP];
    vs_nominal_params(env_type, T.flags, nominal);
    CK(vs_set_params_uniform(h, nominal));
    CK(vs_reset(h, nullptr, 0, 0, nullptr, 0));
    HK(hipStreamSynchronize(h->stream));
#undef CK
#undef HK
    *out = h;
    return VS_OK;
}

int vs_destroy(vs_handle h) {
    if (!h) return VS_OK;
    (void)hipSetDevice(h->device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->stage) (void)hipFree(h->stage);
    if (h->stage_mask) (void)hipFree(h->stage_mask);
    if (h->d_pbuf) (void)hipFree(h->d_pbuf);
    if (h->d_ring) (void)hipFree(h->d_ring);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return VS_OK;
}

int vs_set_stream(vs_handle h, void* s) {
    if (!h) return VS_ERR_ARG;
    // the caller orders work across streams (events / torch stream semantics); a capturing stream must not be synced
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &st) != hipSuccess) (void)hipGetLastError();
    if (st == hipStreamCaptureStatusNone) HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = s ? (hipStream_t)s : h->own_stream;
    return VS_OK;
}

int vs_sync(vs_handle h) {
    if (!h) return VS_ERR_ARG;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return VS_OK;
}

int64_t vs_n_envs(vs_handle h) { return h ? h->d.n : -1; }
int64_t vs_ld(vs_handle h) { return h ? h->d.ld : -1; }

int vs_set_params(vs_handle h, const float* params_soa, int64_t pitch, const uint8_t* mask) {
    if (!h || !params_soa) return fail(h, VS_ERR_ARG, "vs_set_params: NULL argument");
    if (pitch < h->d.n) return fail(h, VS_ERR_ARG, "vs_set_params: pitch < n_envs");
    HIPCHK(h, hipSetDevice(h->device));
    const float* src; long sp; const uint8_t* m;
    int rc = stage_rows(h, params_soa, ENV_INFO[h->type].P, pitch, &src, &sp);
    if (rc) return rc;
    rc = stage_mask(h, mask, &m);
    if (rc) return rc;
    h->uniform = false;
    DISPATCH_ENV(h->type, hipLaunchKernelGGL(k_set_params<E>, grid_for(h->d.ld), dim3(BLOCK), 0, h->stream, h->task,
                                              h->d, src, sp, 0, m));
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_set_params_uniform(vs_handle h, const float* params) {
    if (!h || !params) return fail(h, VS_ERR_ARG, "vs_set_params_uniform: NULL argument");
    HIPCHK(h, hipSetDevice(h->device));
    int P = ENV_INFO[h->type].P;
    if (h->stage_bytes < (size_t)MAXP * 4) {
        if (h->stage) HIPCHK(h, hipFree(h->stage));
        h->stage_bytes = 0;
        HIPCHK(h, hipMalloc(&h->stage, (size_t)MAXP * 4));
        h->stage_bytes = (size_t)MAXP * 4;
    }
    HIPCHK(h, hipMemcpyAsync(h->stage, params, (size_t)P * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // `params` may be a temporary of the caller
    h->uniform = true;
    DISPATCH_ENV(h->type, hipLaunchKernelGGL(k_set_params<E>, grid_for(h->d.ld), dim3(BLOCK), 0, h->stream, h->task,
                                              h->d, (const float*)h->stage, 0L, 1, (const uint8_t*)nullptr));
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_sample_params(vs_handle h, const vs_dp_spec* specs, int n_specs, uint64_t seed, const uint8_t* mask) {
    if (!h) return VS_ERR_ARG;
    DrSpecs dr;
    int rc = check_specs(h, specs, n_specs, &dr);
    if (rc) return rc;
    if (n_specs == 0) return VS_OK;
    HIPCHK(h, hipSetDevice(h->device));
    const uint8_t* m;
    rc = stage_mask(h, mask, &m);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_specs, &dr, sizeof(DrSpecs), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // `dr` lives on this stack frame
    h->uniform = false;
    DISPATCH_ENV(h->type, hipLaunchKernelGGL(k_sample_params<E>, grid_for(h->d.ld), dim3(BLOCK), 0, h->stream,
                                              h->task, h->d, (const DrSpecs*)h->d_specs, seed, m));
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_set_randomizer(vs_handle h, const vs_dp_spec* specs, int n_specs) {
    if (!h) return VS_ERR_ARG;
    DrSpecs dr;
    int rc = check_specs(h, specs, n_specs, &dr);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    if (dr.n > 0 && h->d.pbuf_n > 0) return fail(h, VS_ERR_STATE, "vs_set_randomizer: a parameter buffer is set");
    h->dr = dr;
    h->d.drv = dr;  // travels with every launch as part of the kernel arguments
    h->d.dr_n = dr.n;
    if (n_specs > 0) h->uniform = false;
    return VS_OK;
}

int vs_set_param_buffer(vs_handle h, const float* params_soa, int n_sets, int selection) {
    if (!h || n_sets < 0 || (n_sets > 0 && !params_soa) || selection < 0 || selection > 1)
        return fail(h, VS_ERR_ARG, "vs_set_param_buffer: bad argument");
    if (n_sets > 0 && h->dr.n > 0) return fail(h, VS_ERR_STATE, "vs_set_param_buffer: a live randomizer is set");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->d_pbuf) { HIPCHK(h, hipFree(h->d_pbuf)); h->d_pbuf = nullptr; }
    h->d.pbuf = nullptr;
    h->d.pbuf_n = 0;
    h->d.pbuf_mode = selection;
    if (n_sets == 0) return VS_OK;
    size_t bytes = (size_t)ENV_INFO[h->type].P * n_sets * sizeof(float);
    HIPCHK(h, hipMalloc((void**)&h->d_pbuf, bytes));
    HIPCHK(h, hipMemcpy(h->d_pbuf, params_soa, bytes, hipMemcpyHostToDevice));
    h->d.pbuf = h->d_pbuf;
    h->d.pbuf_n = n_sets;
    h->uniform = false;
    return VS_OK;
}

int vs_set_act_norm(vs_handle h, int on) {
    if (!h) return VS_ERR_ARG;
    if (on) h->task.flags |= VS_FLAG_ACT_NORM;
    else h->task.flags &= ~VS_FLAG_ACT_NORM;
    return VS_OK;
}

int vs_set_act_pipeline(vs_handle h, int delay, const float* noise_mean, const float* noise_std, int noise_normed,
                        int noise_after_delay, uint64_t seed) {
    if (!h) return VS_ERR_ARG;
    if (delay < 0 || delay > VS_MAX_ACT_DELAY) return fail(h, VS_ERR_ARG, "vs_set_act_pipeline: delay must be in [0, VS_MAX_ACT_DELAY]");
    const EnvInfo& ei = ENV_INFO[h->type];
    Pipe& p = h->d.pipe;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (delay != p.delay) {
        if (h->d_ring) { HIPCHK(h, hipFree(h->d_ring)); h->d_ring = nullptr; }
        if (delay > 0) {
            size_t bytes = (size_t)delay * ei.A * h->d.ld * sizeof(float);
            HIPCHK(h, hipMalloc((void**)&h->d_ring, bytes));
            // on the handle's stream: a null-stream memset is not ordered with kernels on a non-blocking stream
            HIPCHK(h, hipMemsetAsync(h->d_ring, 0, bytes, h->stream));
        }
        p.ring = h->d_ring;
        p.delay = delay;
    }
    p.act_noise = 0;
    for (int j = 0; j < MAXA; ++j) {
        p.a_mean[j] = (noise_mean && j < ei.A) ? noise_mean[j] : 0.f;
        p.a_std[j] = (noise_std && j < ei.A) ? noise_std[j] : 0.f;
        if (p.a_std[j] < 0.f || p.a_std[j] != p.a_std[j]) return fail(h, VS_ERR_ARG, "vs_set_act_pipeline: noise_std must be >= 0");
        if (p.a_mean[j] != 0.f || p.a_std[j] != 0.f) p.act_noise = 1;
    }
    p.noise_normed = noise_normed != 0;
    p.noise_after_delay = noise_after_delay != 0;
    p.seed = seed;
    p.act_on = p.delay > 0 || p.act_noise;
    return VS_OK;
}

int vs_set_obs_pipeline(vs_handle h, const float* scale, const float* shift, const float* noise_std, uint64_t seed) {
    if (!h) return VS_ERR_ARG;
    const EnvInfo& ei = ENV_INFO[h->type];
    Pipe& p = h->d.pipe;
    p.obs_noise = 0;
    bool ident = true;
    for (int j = 0; j < MAXO; ++j) {
        p.o_scale[j] = (scale && j < ei.O) ? scale[j] : 1.f;
        p.o_shift[j] = (shift && j < ei.O) ? shift[j] : 0.f;
        p.o_std[j] = (noise_std && j < ei.O) ? noise_std[j] : 0.f;
        if (p.o_std[j] < 0.f || p.o_std[j] != p.o_std[j]) return fail(h, VS_ERR_ARG, "vs_set_obs_pipeline: noise_std must be >= 0");
        if (p.o_std[j] != 0.f) p.obs_noise = 1;
        if (p.o_scale[j] != 1.f || p.o_shift[j] != 0.f) ident = false;
    }
    p.seed = seed;
    p.obs_on = !ident || p.obs_noise;
    // VS_OBS is the wrapped observation from now on
    HIPCHK(h, hipSetDevice(h->device));
    DISPATCH_ENV(h->type, hipLaunchKernelGGL(k_observe<E>, grid_for(h->d.ld), dim3(BLOCK), 0, h->stream, h->d));
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_reset(vs_handle h, const float* init_state, int64_t pitch, int full, const uint8_t* mask, uint64_t seed) {
    if (!h) return VS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    const EnvInfo& ei = ENV_INFO[h->type];
    const float* src = nullptr; long sp = 0; const uint8_t* m;
    int rc;
    if (init_state) {
        if (pitch < h->d.n) return fail(h, VS_ERR_ARG, "vs_reset: pitch < n_envs");
        rc = stage_rows(h, init_state, full ? ei.S : ei.I, pitch, &src, &sp);
        if (rc) return rc;
    }
    rc = stage_mask(h, mask, &m);
    if (rc) return rc;
    DISPATCH_ENV(h->type, hipLaunchKernelGGL(k_reset<E>, grid_for(h->d.ld), dim3(BLOCK), 0, h->stream, h->task, h->d,
                                              src, sp, full, m, seed));
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_set_max_steps(vs_handle h, int64_t max_steps) {
    if (!h) return VS_ERR_ARG;
    h->task.max_steps = (max_steps <= 0 || max_steps >= INT_MAX) ? INT_MAX : (int)max_steps;
    return VS_OK;
}

int vs_set_dt(vs_handle h, double dt) {
    if (!h) return VS_ERR_ARG;
    if (!(dt >= 0.0)) return fail(h, VS_ERR_ARG, "vs_set_dt: dt must be >= 0");
    h->task.dt = (float)dt;  // no derived constant depends on the step size
    return VS_OK;
}

int vs_set_index_offset(vs_handle h, uint32_t first_global_index) {
    if (!h) return VS_ERR_ARG;
    h->d.idx0 = first_global_index;
    return VS_OK;
}

int vs_set_auto_reset(vs_handle h, int on, uint64_t seed) {
    if (!h) return VS_ERR_ARG;
    h->auto_reset = on != 0;
    h->ar_seed = seed;
    return VS_OK;
}

int vs_step(vs_handle h, const float* actions, int64_t env_stride, int64_t dim_stride) {
    if (!h || !actions) return fail(h, VS_ERR_ARG, "vs_step: NULL argument");
    // while the stream is being captured into a hipGraph only the launch itself may be issued (pointer queries and
    // device switches invalidate the capture); the pointer was validated by the eager warm-up call
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &st) != hipSuccess) (void)hipGetLastError();
    if (st == hipStreamCaptureStatusNone) {
        if (!is_device_ptr(actions)) return fail(h, VS_ERR_ARG, "vs_step: actions must be device memory");
        HIPCHK(h, hipSetDevice(h->device));
    }
    DISPATCH_ENV(h->type, launch_step<E>(h, actions, (long)env_stride, (long)dim_stride));
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_seek_random(vs_handle h, uint64_t step_index) {
    if (!h) return VS_ERR_ARG;
    h->epoch = step_index;
    return VS_OK;
}

int vs_step_jac(vs_handle h, const float* actions, int64_t env_stride, int64_t dim_stride) {
    if (!h || !actions) return fail(h, VS_ERR_ARG, "vs_step_jac: NULL argument");
    if (!is_device_ptr(actions)) return fail(h, VS_ERR_ARG, "vs_step_jac: actions must be device memory");
    if (h->auto_reset) return fail(h, VS_ERR_STATE, "vs_step_jac: switch auto-reset off (the Jacobian of a reset is meaningless)");
    if (h->d.pipe.act_on || h->d.pipe.obs_on)
        return fail(h, VS_ERR_STATE, "vs_step_jac: remove the action/observation pipeline (Jacobians are those of the bare env)");
    HIPCHK(h, hipSetDevice(h->device));
    const EnvInfo& ei = ENV_INFO[h->type];
    if (!h->d.jac_s) {
        size_t ni = (size_t)(ei.S + ei.A), ld = h->d.ld;
        int rc;
        if ((rc = dalloc(h, &h->d.jac_s, ei.S * ni * ld))) return rc;
        if ((rc = dalloc(h, &h->d.jac_r, ni * ld))) return rc;
        if ((rc = dalloc(h, &h->d.jac_o, ei.O * ni * ld))) return rc;
    }
    bool uni = h->uniform && h->dr.n == 0 && h->d.pbuf_n == 0;
    dim3 g = grid_for(h->d.ld), b(BLOCK);
    if (uni) { DISPATCH_ENV(h->type, hipLaunchKernelGGL((k_step_jac<E, true>), g, b, 0, h->stream, h->task, h->d, actions, (long)env_stride, (long)dim_stride)); }
    else { DISPATCH_ENV(h->type, hipLaunchKernelGGL((k_step_jac<E, false>), g, b, 0, h->stream, h->task, h->d, actions, (long)env_stride, (long)dim_stride)); }
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_set_traj_capacity(vs_handle h, int t_max) {
    if (!h || t_max < 0) return fail(h, VS_ERR_ARG, "vs_set_traj_capacity: bad argument");
    if (t_max <= h->traj_cap) return VS_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const EnvInfo& ei = ENV_INFO[h->type];
    Dev& d = h->d;
    size_t ld = d.ld;
    // old buffers stay in h->allocs until destroy (capacity only grows a handful of times)
    int rc;
    if ((rc = dalloc(h, &d.traj_rec, (size_t)t_max * (ei.O + ei.A + 1) * ld))) return rc;
    if ((rc = dalloc(h, &d.traj_done, (size_t)t_max * ld))) return rc;
    h->traj_cap = t_max;
    return VS_OK;
}

int vs_set_rollout_variant(vs_handle h, int variant) {
    if (!h || variant < -1 || variant > 1) return fail(h, VS_ERR_ARG, "vs_set_rollout_variant: -1 (automatic), 0 or 1");
    h->rollout_variant = variant;
    return VS_OK;
}

int vs_rollout_variant(vs_handle h) {
    if (!h) return VS_ERR_ARG;
    int ws = 0;
    DISPATCH_ENV(h->type, ws = use_ws<E>(h) ? 1 : 0);
    return ws;
}

int vs_set_traj_offset(vs_handle h, int t0) {
    if (!h || t0 < 0) return fail(h, VS_ERR_ARG, "vs_set_traj_offset: bad argument");
    h->d.traj_t0 = t0;
    return VS_OK;
}

int vs_step_random(vs_handle h, uint64_t seed, int k_steps, int record) {
    if (!h || k_steps < 1) return fail(h, VS_ERR_ARG, "vs_step_random: bad argument");
    if (record && h->d.traj_t0 + k_steps > h->traj_cap) return fail(h, VS_ERR_STATE, "vs_step_random: traj offset + k_steps exceeds vs_set_traj_capacity");
    HIPCHK(h, hipSetDevice(h->device));
    uint64_t ep = h->epoch;
    h->epoch += (uint64_t)k_steps;
    DISPATCH_ENV(h->type, launch_rollout<E>(h, k_steps, seed, ep, record != 0));
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_set_episode_log(vs_handle h, int on) {
    if (!h) return VS_ERR_ARG;
    h->d.log_episodes = on != 0;
    return VS_OK;
}

int vs_mixed_create(const vs_handle* handles, int n, vs_mixed_handle* out) {
    if (!handles || !out || n < 1 || n > MAX_SEG) return fail(nullptr, VS_ERR_ARG, "vs_mixed_create: need 1..5 handles");
    *out = nullptr;
    for (int q = 0; q < n; ++q) {
        if (!handles[q]) return fail(nullptr, VS_ERR_ARG, "vs_mixed_create: NULL handle");
        if (handles[q]->device != handles[0]->device) return fail(nullptr, VS_ERR_ARG, "vs_mixed_create: handles on different devices");
        if (handles[q]->auto_reset != handles[0]->auto_reset) return fail(nullptr, VS_ERR_ARG, "vs_mixed_create: handles differ in auto-reset");
    }
    vs_mixed* m = new (std::nothrow) vs_mixed();
    if (!m) return fail(nullptr, VS_ERR_HIP, "vs_mixed_create: out of host memory");
    m->n = n;
    for (int q = 0; q < n; ++q) {
        m->sub[q] = handles[q];
        handles[q]->stream = handles[0]->stream;  // one launch, one stream
    }
    if (hipSetDevice(handles[0]->device) != hipSuccess || hipMalloc((void**)&m->dev, sizeof(Segs)) != hipSuccess) {
        delete m;
        return fail(nullptr, VS_ERR_HIP, "vs_mixed_create: hipMalloc failed");
    }
    *out = m;
    return VS_OK;
}

int vs_mixed_destroy(vs_mixed_handle m) {
    if (!m) return VS_OK;
    if (m->dev) (void)hipFree(m->dev);
    delete m;
    return VS_OK;
}

const char* vs_mixed_last_error(vs_mixed_handle m) { return m ? m->err.c_str() : ""; }

int vs_mixed_step_random(vs_mixed_handle m, uint64_t seed, int k_steps, int record) {
    if (!m || k_steps < 1) return VS_ERR_ARG;
    for (int q = 0; q < m->n; ++q) {
        if (record && m->sub[q]->d.traj_t0 + k_steps > m->sub[q]->traj_cap) { m->err = "vs_mixed_step_random: k_steps exceeds a segment's vs_set_traj_capacity"; return VS_ERR_STATE; }
        if (m->sub[q]->auto_reset != m->sub[0]->auto_reset) { m->err = "vs_mixed_step_random: segments differ in auto-reset"; return VS_ERR_STATE; }
    }
    if (hipSetDevice(m->sub[0]->device) != hipSuccess) return VS_ERR_HIP;
    int rc = mixed_upload(m, nullptr, nullptr, nullptr, k_steps);
    if (rc) return rc;
    dim3 g((unsigned)m->total_blocks), b(BLOCK);
    hipStream_t st = m->sub[0]->stream;
    bool ar = m->sub[0]->auto_reset;
    if (ar) { if (record) hipLaunchKernelGGL((k_rollout_mixed<true, true>), g, b, 0, st, (const Segs*)m->dev, k_steps, seed);
              else hipLaunchKernelGGL((k_rollout_mixed<true, false>), g, b, 0, st, (const Segs*)m->dev, k_steps, seed); }
    else { if (record) hipLaunchKernelGGL((k_rollout_mixed<false, true>), g, b, 0, st, (const Segs*)m->dev, k_steps, seed);
           else hipLaunchKernelGGL((k_rollout_mixed<false, false>), g, b, 0, st, (const Segs*)m->dev, k_steps, seed); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { m->err = hipGetErrorString(e); return VS_ERR_HIP; }
    return VS_OK;
}

int vs_mixed_step(vs_mixed_handle m, const float* const* actions, const int64_t* env_strides, const int64_t* dim_strides) {
    if (!m || !actions || !env_strides || !dim_strides) return VS_ERR_ARG;
    for (int q = 0; q < m->n; ++q)
        if (!actions[q] || !is_device_ptr(actions[q])) { m->err = "vs_mixed_step: actions must be device memory"; return VS_ERR_ARG; }
    if (hipSetDevice(m->sub[0]->device) != hipSuccess) return VS_ERR_HIP;
    int rc = mixed_upload(m, actions, env_strides, dim_strides, 0);
    if (rc) return rc;
    dim3 g((unsigned)m->total_blocks), b(BLOCK);
    hipStream_t st = m->sub[0]->stream;
    if (m->sub[0]->auto_reset) hipLaunchKernelGGL((k_step_mixed<true>), g, b, 0, st, (const Segs*)m->dev);
    else hipLaunchKernelGGL((k_step_mixed<false>), g, b, 0, st, (const Segs*)m->dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { m->err = hipGetErrorString(e); return VS_ERR_HIP; }
    return VS_OK;
}

int vs_mixed_time_random(vs_mixed_handle m, uint64_t seed, int k_steps, int record, int iters, float* avg_ms) {
    if (!m || !avg_ms || iters < 1) return VS_ERR_ARG;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return VS_ERR_HIP;
    int rc = vs_mixed_step_random(m, seed, k_steps, record);
    hipStream_t st = m->sub[0]->stream;
    if (rc == VS_OK) {
        (void)hipEventRecord(e0, st);
        for (int it = 0; it < iters && rc == VS_OK; ++it) rc = vs_mixed_step_random(m, seed, k_steps, record);
        (void)hipEventRecord(e1, st);
        float ms = 0.f;
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = VS_ERR_HIP;
        *avg_ms = ms / (float)iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int vs_clear_episodes(vs_handle h) {
    if (!h) return VS_ERR_ARG;
    HIPCHK(h, hipMemsetAsync(h->d.ep_count, 0, sizeof(unsigned), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d.es_count, 0, (size_t)h->d.ld * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d.es_retsum, 0, (size_t)h->d.ld * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d.es_lensum, 0, (size_t)h->d.ld * 4, h->stream));
    return VS_OK;
}

static bool buf_info(vs_handle h, int which, void** p, size_t* bytes) {
    const EnvInfo& ei = ENV_INFO[h->type];
    Dev& d = h->d;
    size_t ld = d.ld;
    switch (which) {
        case VS_STATE: *p = d.state; *bytes = ei.S * ld * 4; return true;
        case VS_OBS: *p = d.obs; *bytes = ei.O * ld * 4; return true;
        case VS_REW: *p = d.rew; *bytes = ld * 4; return true;
        case VS_DONE: *p = d.done; *bytes = ld; return true;
        case VS_HIDDEN: *p = d.hidden; *bytes = (size_t)ei.H * ld * 4; return true;
        case VS_STEPCOUNT: *p = d.step; *bytes = ld * 4; return true;
        case VS_ERRFLAG: *p = d.err; *bytes = ld; return true;
        case VS_RETURNS: *p = d.ret; *bytes = ld * 4; return true;
        case VS_PARAMS: *p = d.params; *bytes = ei.P * ld * 4; return true;
        case VS_CONSTS: *p = d.consts; *bytes = ei.K * ld * 4; return true;
        case VS_EP_RETURNS: *p = d.ep_ret; *bytes = (size_t)d.ep_cap * 4; return true;
        case VS_EP_LENGTHS: *p = d.ep_len; *bytes = (size_t)d.ep_cap * 4; return true;
        case VS_EP_ENVIDX: *p = d.ep_env; *bytes = (size_t)d.ep_cap * 4; return true;
        case VS_EP_COUNT: *p = d.ep_count; *bytes = 4; return true;
        case VS_TRAJ_REC: *p = d.traj_rec; *bytes = (size_t)h->traj_cap * (ei.O + ei.A + 1) * ld * 4; return true;
        case VS_TRAJ_DONE: *p = d.traj_done; *bytes = (size_t)h->traj_cap * ld; return true;
        case VS_FAILED: *p = d.failed; *bytes = ld; return true;
        case VS_EPSTAT_COUNT: *p = d.es_count; *bytes = ld * 4; return true;
        case VS_EPSTAT_RETSUM: *p = d.es_retsum; *bytes = ld * 4; return true;
        case VS_EPSTAT_LENSUM: *p = d.es_lensum; *bytes = ld * 4; return true;
        case VS_JAC_STATE: *p = d.jac_s; *bytes = d.jac_s ? (size_t)ei.S * (ei.S + ei.A) * ld * 4 : 0; return true;
        case VS_JAC_REW: *p = d.jac_r; *bytes = d.jac_r ? (size_t)(ei.S + ei.A) * ld * 4 : 0; return true;
        case VS_JAC_OBS: *p = d.jac_o; *bytes = d.jac_o ? (size_t)ei.O * (ei.S + ei.A) * ld * 4 : 0; return true;
        default: return false;
    }
}

void* vs_get(vs_handle h, int which) {
    if (!h) return nullptr;
    void* p; size_t b;
    if (!buf_info(h, which, &p, &b)) { fail(h, VS_ERR_ARG, "vs_get: unknown buffer"); return nullptr; }
    return p;
}

int vs_copy_to_host(vs_handle h, int which, void* dst) {
    if (!h || !dst) return fail(h, VS_ERR_ARG, "vs_copy_to_host: NULL argument");
    void* p; size_t b;
    if (!buf_info(h, which, &p, &b)) return fail(h, VS_ERR_ARG, "vs_copy_to_host: unknown buffer");
    if (b == 0) return VS_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(dst, p, b, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return VS_OK;
}

int vs_copy_from_host(vs_handle h, int which, const void* src) {
    if (!h || !src) return fail(h, VS_ERR_ARG, "vs_copy_from_host: NULL argument");
    if (which != VS_STATE && which != VS_HIDDEN && which != VS_STEPCOUNT)
        return fail(h, VS_ERR_ARG, "vs_copy_from_host: only VS_STATE / VS_HIDDEN / VS_STEPCOUNT are assignable");
    void* p; size_t b;
    buf_info(h, which, &p, &b);
    if (b == 0) return VS_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(p, src, b, hipMemcpyHostToDevice, h->stream));
    if (which == VS_STATE) {
        DISPATCH_ENV(h->type, hipLaunchKernelGGL(k_observe<E>, grid_for(h->d.ld), dim3(BLOCK), 0, h->stream, h->d));
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return VS_OK;
}

int64_t vs_error_count(vs_handle h) {
    if (!h) return -1;
    if (hipSetDevice(h->device) != hipSuccess) return -1;
    if (hipMemsetAsync(h->d_counter, 0, 8, h->stream) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_count_err, grid_for(h->d.ld), dim3(BLOCK), 0, h->stream, h->d.err, h->d.n, h->d_counter);
    unsigned long long v = 0;
    if (hipMemcpyAsync(&v, h->d_counter, 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return -1;
    return (int64_t)v;
}

int vs_time_step_kernel(vs_handle h, int mode, const float* actions, int64_t env_stride, int64_t dim_stride,
                        int k_steps, int record, int iters, float* avg_ms) {
    if (!h || !avg_ms || iters < 1) return fail(h, VS_ERR_ARG, "vs_time_step_kernel: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    int rc = VS_OK;
    // events on the stream the kernels are launched on; one warm launch first
    rc = mode == 0 ? vs_step(h, actions, env_stride, dim_stride) : vs_step_random(h, 1234, k_steps, record);
    if (rc == VS_OK) {
        (void)hipEventRecord(e0, h->stream);
        for (int it = 0; it < iters && rc == VS_OK; ++it)
            rc = mode == 0 ? vs_step(h, actions, env_stride, dim_stride) : vs_step_random(h, 1234, k_steps, record);
        (void)hipEventRecord(e1, h->stream);
        hipError_t e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) rc = fail(h, VS_ERR_HIP, "vs_time_step_kernel: event timing", e);
        *avg_ms = ms / (float)iters;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int vs_membw_probe(int device_id, int64_t bytes, int iters, float* gbps) {
    if (!gbps || bytes < (1 << 20) || iters < 1) return VS_ERR_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return VS_ERR_HIP;
    void *a = nullptr, *b = nullptr;
    size_t n4 = (size_t)bytes / 16;
    if (hipMalloc(&a, n4 * 16) != hipSuccess) return VS_ERR_HIP;
    if (hipMalloc(&b, n4 * 16) != hipSuccess) { (void)hipFree(a); return VS_ERR_HIP; }
    (void)hipMemset(a, 1, n4 * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_copy4, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_copy4, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4);
    (void)hipEventRecord(e1, 0);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    if (e != hipSuccess || ms <= 0.f) return VS_ERR_HIP;
    *gbps = (float)(2.0 * (double)(n4 * 16) * iters / (ms * 1e-3) / 1e9);  // read + write
    return VS_OK;
}

int vs_memwrite_probe(int device_id, int64_t bytes, int iters, float* gbps) {
    if (!gbps || bytes < (1 << 20) || iters < 1) return VS_ERR_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return VS_ERR_HIP;
    void* a = nullptr;
    size_t n4 = (size_t)bytes / 16;
    if (hipMalloc(&a, n4 * 16) != hipSuccess) return VS_ERR_HIP;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_fill4, dim3(2048), dim3(256), 0, 0, (float4*)a, n4, 0.f);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_fill4, dim3(2048), dim3(256), 0, 0, (float4*)a, n4, (float)i);
    (void)hipEventRecord(e1, 0);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    if (e != hipSuccess || ms <= 0.f) return VS_ERR_HIP;
    *gbps = (float)((double)(n4 * 16) * iters / (ms * 1e-3) / 1e9);  // write only
    return VS_OK;
}

}  // extern "C"
