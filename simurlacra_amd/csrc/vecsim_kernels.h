// vecsim_kernels.h -- libvecsim: the HIP kernels (gfx950 / CDNA4) and their launchers.
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
//                                     LDS: what runs while k_rollout would leave the SIMDs with a single wave
//   k_*_mixed       vs_mixed_*        several families in one launch (one workgroup = one family)
//   k_step_jac      vs_step_jac       step + Jacobians by forward-mode dual numbers
//   k_reset / k_set_params / k_sample_params / k_observe   control path
// Variants carrying the wrapper pipeline (action noise / delay, observation normalisation / noise) are separate
// instantiations (template parameter PIPE): the default kernels do not pay for it.
//
// Translation units (built in parallel by build.py and linked into ONE libvecsim.so):
//   vecsim.hip         the C-ABI of include/vecsim.h (host code, non-template kernels)
//   vecsim_family.hip  compiled once per env family (-DVS_FAMILY=n): the kernels of that family and their launchers
//   vecsim_mixed.hip   the mixed-batch kernels (every family's body behind one workgroup-uniform switch)
#pragma once
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
    uint32_t* traj_done; // done flags of the recorded steps, one BIT per env and step: word [t / 32][env], bit t % 32
    int traj_t0;  // record row offset of the next recording vs_step_random
    int traj_rows;  // capacity of the record buffers in rows
    int* rec_row;   // device-side row counter of vs_step_record(row < 0), advanced by k_bump_row
    float *jac_s, *jac_r, *jac_o;  // step Jacobians (vs_step_jac), allocated on first use
    unsigned long long* dbg;       // diagnostic builds only (-DVS_WS_STAMP): per-wave cycle sums, [ld / 64][3 roles][4]
    int n, ld;
};

#ifdef VS_WS_STAMP  // diagnostic builds only: where the two waves of k_rollout_ws spend their cycles
#define VS_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define VS_STAMP(var)
#endif

// The lane index as a RARE block (reset, redraw, refill, the epilogue of a kernel) must see it: a value the optimiser has to
// take as defined right here.  Every address such a block computes (VS_PARAMS / VS_CONSTS rows of the lane, the stock entry)
// is loop-invariant, so LICM hoists the arithmetic out of the step loop and the results -- two VGPRs per 64-bit row address,
// 33 rows for the cartpole's redraw -- stay live across the whole hot loop.  That, not the rare code's own needs, is what pushed
// the auto-reset kernels of round 2 to their register caps and into scratch (cartpole: 256 VGPRs + 35 spills; with the index
// laundered 131 and none; QQube headline kernel 168 + 8 -> 117).  An empty asm costs no instruction.
__device__ __forceinline__ int cold_lane(int i) {
    asm volatile("" : "+v"(i));
    return i;
}
#ifdef VS_NO_COLD_LANE  // diagnostic builds only: the round-2 code generation
#define VS_COLD(i) (i)
#else
#define VS_COLD(i) cold_lane(i)
#endif

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
// tr: E::observe_p of the pre-step state if the caller holds it in registers (the trig observe() shares with the dynamics)
template <class E, class R>
__device__ __forceinline__ StepOutT<R> step_one(const Task& T, const float* c, R* s, R* h, const R* a_raw, int& step,
                                                bool& yielded, const R* tr, const Dev* dp = nullptr, int lane = 0,
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
    E::dynamics(T, c, s, h, a, tr);
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

// Non-temporal access for the single-step kernel at large batch sizes (NT): one step of 16.7 M envs touches 2 GB once per
// launch -- nothing of it is in a cache when the next launch comes, and keeping it out of the caches' way is worth 27 % there
// (k_step at 16.7 M QQube envs, same box: 0.57 -> 0.73 of HBM); up to ~2 M envs the working set lives in the 256 MB Infinity
// Cache from one launch to the next and the default policy is the right one (STEP_NT_MIN_ENVS).
template <bool NT, class V>
__device__ __forceinline__ V ld_nt(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, class V>
__device__ __forceinline__ void st_nt(V v, V* p) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <class E, bool UNI, bool NT = false>
__device__ __forceinline__ void load_consts(const Dev& d, int i, float* c, int first, int last) {
#pragma unroll
    for (int k = 0; k < E::K; ++k)
        if (k >= first && k < last) c[k] = UNI ? d.consts_uni[k] : ld_nt<NT>(&d.consts[(size_t)k * d.ld + i]);
}

// DomainRandomizer.randomize for one lane (domain_parameter.py:104-132): draw -> clamp; writes the raw params
__device__ __forceinline__ float draw_one_param(const vs_dp_spec& sp, Rng& g) {
    float v;
    if (sp.kind == VS_DP_NORMAL) v = sp.mean + sp.spread * g.normal();
    else if (sp.kind == VS_DP_UNIFORM) v = g.uniform(sp.mean - sp.spread, sp.mean + sp.spread);
    else v = g.u01() < sp.aux ? sp.spread : sp.mean;  // Bernoulli(prob_1): val_1 with probability prob_1, else val_0
    v = fminf(fmaxf(v, sp.clip_lo), sp.clip_up);
    if (sp.roundint) v = rintf(v);  // torch.round: half to even
    return v;
}
template <class E>
__device__ __forceinline__ void draw_params(const DrSpecs* dr, Rng& g, float* p) {
    int n = dr->n;
    for (int q = 0; q < n; ++q) {
        const vs_dp_spec sp = dr->s[q];
        const float v = draw_one_param(sp, g);
#pragma unroll
        for (int k = 0; k < E::P; ++k)
            if (k == sp.param_index) p[k] = v;
    }
}

// lds_params (k_rollout_ws under a live randomizer): the lane's parameter rows in LDS, [P][lds_pitch] -- the rows a redraw does
// not touch are the same there as in VS_PARAMS, the randomised ones are overwritten by the draws below whatever they hold: the
// redraw then starts without waiting for P global loads (~3 000 cycles on the wave that runs it for one or two of its lanes)
template <class E>
__device__ __forceinline__ void redraw_lane_params(const Task& T, const Dev& d, const DrSpecs* dr, int i, uint64_t seed,
                                                   uint64_t epi, float* c, const float* lds_params = nullptr, int lds_pitch = 0,
                                                   int lds_lane = 0) {
    float p[E::P];
#pragma unroll
    for (int k = 0; k < E::P; ++k) p[k] = lds_params ? lds_params[k * lds_pitch + lds_lane] : d.params[(size_t)k * d.ld + i];
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
// with_dr / with_pbuf are compile-time constants at every call site (inlined): the kernels of a handle without a live randomizer /
// without a parameter buffer do not carry that block
template <class E>
__device__ __forceinline__ void reset_lane_sampled(const Task& T, const Dev& d, bool with_dr, int i, uint64_t seed,
                                                   uint64_t epi, float* c, float* s, float* h,
                                                   const float* lds_params = nullptr, int lds_pitch = 0, int lds_lane = 0,
                                                   bool with_pbuf = true) {
    if (with_dr && d.dr_n > 0) redraw_lane_params<E>(T, d, &d.drv, i, seed, epi, c, lds_params, lds_pitch, lds_lane);
    if (with_dr && with_pbuf && d.pbuf_n > 0) {
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
    float init[E::I];
#ifdef VS_ABLATE_RESET  // diagnostic builds only: what the reset path costs (a reset lane restarts from a fixed state)
#pragma unroll
    for (int j = 0; j < E::I; ++j) init[j] = 0.01f * (float)(j + 1);
#else
    Rng g(seed, d.idx0 + (uint32_t)i, RNG_INIT, epi);
    E::sample_init(T, c, g, init);
#endif
    E::state_from_init(init, s);
    E::init_hidden(T, c, nullptr, s, h, false);
}

// completed-episode append: the lanes of a wave that finished take consecutive slots of the ring.  The ranking is the
// canonical wave idiom -- one returning atomic by the first finishing lane for the whole wave's count, its result broadcast
// with v_readfirstlane, ranks from v_mbcnt over the ballot mask -- with nothing routed through ds_bpermute (an earlier
// __shfl-based form lost entries on MI355X in a layout-dependent way: all lanes of a wave ended up with rank 0).
__device__ __forceinline__ void append_episode(const Dev& d, bool fin, int i, float ret, int len) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(fin);
    if (m == 0ull) return;
    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    unsigned base = 0;
    if (fin && rank == 0u) base = atomicAdd(d.ep_count, (unsigned)__popcll(m));
    // the first finishing lane holds the base: the lowest set bit of m; read it from that lane into an SGPR
    base = (unsigned)__builtin_amdgcn_readlane((int)base, __ffsll((long long)m) - 1);
    if (fin) {
        unsigned slot = (base + rank) % d.ep_cap;
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
// DRK: the redraw of a live randomizer / parameter buffer is compiled in (see k_rollout_ws); the kernels of a handle that has
// neither carry none of it
template <class E, bool UNI, bool DRK = !UNI>
__device__ __forceinline__ void auto_reset(const Task& T, const Dev& d, bool fin, int i, uint64_t seed, float* c,
                                           float* s, float* h, int& step, float& ret, bool& yielded, EpStat& es) {
    if (__builtin_amdgcn_ballot_w64(fin) == 0ull) return;
    i = VS_COLD(i);  // (see cold_lane: the addresses of this block are not to be carried through the caller's step loop)
    if (d.log_episodes) append_episode(d, fin, i, ret, step);
    if (fin) {
        es.count += 1u;
        es.retsum += ret;
        es.lensum += step;
        load_consts<E, UNI>(d, i, c, E::KS, E::K);  // reset-only constants
        reset_lane_sampled<E>(T, d, DRK, i, seed, (uint64_t)es.epi, c, s, h);
        es.epi += 1u;
        step = 0;
        ret = 0.f;
        yielded = false;
    }
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
    // NT: non-temporal stores (the record stream is written once and read by another kernel much later)
    template <int Q0 = 0, bool NT = false>
    __device__ __forceinline__ static void store(float* __restrict__ row, size_t ld, int i, const float* v) {
        static_assert(Q0 <= NQ, "first plane");
        typedef float f4 __attribute__((ext_vector_type(4)));
        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int q = Q0; q < NQ; ++q) {
            f4 x = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            f4* p = reinterpret_cast<f4*>(row + (size_t)q * 4 * ld) + i;
            if (NT) __builtin_nontemporal_store(x, p);
            else *p = x;
        }
        if (H2) {
            f2 x = {v[4 * NQ], v[4 * NQ + 1]};
            f2* p = reinterpret_cast<f2*>(row + (size_t)NQ * 4 * ld) + i;
            if (NT) __builtin_nontemporal_store(x, p);
            else *p = x;
        }
        if (H1) {
            float* p = row + ((size_t)NQ * 4 + H2 * 2) * ld + i;
            if (NT) __builtin_nontemporal_store(v[F - 1], p);
            else *p = v[F - 1];
        }
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
// Record modes (vs_set_record_mode): REC = 1 the F1 floats above -- what an on-policy algorithm reads; REC = 2 appends what
// rollout() also keeps per step (P/sampling/rollout.py:237-258): the state BEFORE the step, the applied action
// env.limit_act(act) and the hidden state before the step (qcp: th_ddot):
//   [obs (O) | act (A) | rew | state (S) | act_app (A) | hidden (H)],  F2 = F1 + S + A + H
template <class E, int REC>
struct Rec {
    static constexpr int F = REC == 2 ? E::O + E::A + 1 + E::S + E::A + E::H : E::O + E::A + 1;
};
// s_pre / a_app / h_pre are read for REC == 2 only
// Q0: first 4-float plane to store (1: plane 0 is written by another wave, see k_rollout_ws's generator wave)
template <class E, int REC, int Q0 = 0>
__device__ __forceinline__ void store_record(float* __restrict__ row, size_t ld, int i, const float* ob, const float* a,
                                             float rew, const float* s_pre, const float* a_app, const float* h_pre) {
    constexpr int F = Rec<E, REC>::F;
    float v[F];
#pragma unroll
    for (int j = 0; j < E::O; ++j) v[j] = ob[j];
#pragma unroll
    for (int j = 0; j < E::A; ++j) v[E::O + j] = a[j];
    v[E::O + E::A] = rew;
    if (REC == 2) {
        constexpr int B = E::O + E::A + 1;
#pragma unroll
        for (int j = 0; j < E::S; ++j) v[B + j] = s_pre[j];
#pragma unroll
        for (int j = 0; j < E::A; ++j) v[B + E::S + j] = a_app[j];
#pragma unroll
        for (int j = 0; j < E::H; ++j) v[B + E::S + E::A + j] = h_pre[j];
    }
    // non-temporal: +2 .. 6 % on every recording kernel (QQube 65 536 envs 1.63e11 -> 1.69e11, 1 M envs 1.11e11 -> 1.18e11)
    Planes<F>::template store<Q0, true>(row, ld, i, v);
}

// done flags of the recorded steps: one bit per env and step, 32 steps to a word, words [t / 32][ld] -- a lane keeps the
// word of the running 32-step window in a register and a wave stores it with ONE coalesced dword store per 32 steps
// (a byte per env and step was a 64-B partial-line store per wave and step next to the two 1-KB record stores).
// Row offsets (vs_set_traj_offset) need not be multiples of 32: the first word is completed, not overwritten.
struct DoneBits {
    uint32_t w;
    __device__ __forceinline__ static uint32_t* word(const Dev& d, int i, size_t row) {
        return d.traj_done + (row >> 5) * (size_t)d.ld + i;
    }
    __device__ __forceinline__ void begin(const Dev& d, int i, size_t row0) {
        const unsigned b = (unsigned)(row0 & 31u);  // wave-uniform
        w = b ? *word(d, i, row0) & ((1u << b) - 1u) : 0u;
    }
    // row = absolute record row of this step; last = it is the last recorded step of the launch (both wave-uniform)
    __device__ __forceinline__ void put(const Dev& d, int i, size_t row, bool done, bool last) {
        const unsigned b = (unsigned)(row & 31u);
        w |= (done ? 1u : 0u) << b;
        if (b == 31u || last) {
            *word(d, i, row) = w;
            w = 0u;
        }
    }
};

// env.limit_act(act) of the OUTERMOST env, what rollout() records as the applied action (rollout.py:244; Env.limit_act
// P/environments/base.py:215-222): the projection onto the act space the policy sees -- [-1, 1] under ActNormWrapper
template <class E>
__device__ __forceinline__ void applied_action(const Task& T, const float* c, const float* alo, const float* ahi,
                                               const float* a, float* a_app) {
    const bool nrm = (T.flags & VS_FLAG_ACT_NORM) != 0;
    float lo[E::A], hi[E::A];
#pragma unroll
    for (int j = 0; j < E::A; ++j) {
        lo[j] = nrm ? -1.0f : alo[j];
        hi[j] = nrm ? 1.0f : ahi[j];
    }
    E::limit_act(c, lo, hi, a, a_app);
}

// ------------------------------------------------------------------------------------------------------- step kernel
// vs_step: one fused SimPyEnv.step per lane.  AR = auto-reset of finished lanes inside the same launch.
// No host-side counter enters the kernel: replaying a captured hipGraph of vs_step launches is safe.
// PIPE: the wrapper pipeline (struct Pipe) is compiled in; the default kernels do not carry it.
// REC (vs_step_record): the step also writes its record -- what rollout() keeps of a step taken with the caller's policy
// (rollout.py:237-258): the observation the policy saw (VS_OBS as the previous step / the reset left it), the policy's
// action, the reward, and in mode 2 the state and hidden state before the step and env.limit_act(act) -- into row `row` of
// the VS_TRAJ_* buffers, or, for row < 0, into the row the handle's device-side counter names (Dev::rec_row: a captured
// hipGraph replays with the counter advanced by k_bump_row between the steps).  Rows beyond the capacity are not written.
template <class E, bool UNI, bool AR, bool PIPE = false, int REC = 0, bool DRK = !UNI, bool NT = false>
__device__ __forceinline__ void step_body(const Task& T, const Dev& d, const float* __restrict__ act, long env_stride,
                                          long dim_stride, uint64_t seed, int block, int row = 0) {
    int i = block * BLOCK + threadIdx.x;
    if (i >= d.ld) return;
    const size_t ld = d.ld;
    float c[E::K], s[E::S], h[E::H > 0 ? E::H : 1], a[E::A];
    load_consts<E, UNI, NT>(d, i, c, 0, E::KS);
#pragma unroll
    for (int j = 0; j < E::S; ++j) s[j] = ld_nt<NT>(&d.state[j * ld + i]);
#pragma unroll
    for (int j = 0; j < E::H; ++j) h[j] = ld_nt<NT>(&d.hidden[j * ld + i]);
    bool valid = i < d.n;
#pragma unroll
    for (int j = 0; j < E::A; ++j) a[j] = valid ? ld_nt<NT>(&act[(size_t)i * env_stride + (size_t)j * dim_stride]) : 0.f;
    int step = ld_nt<NT>(&d.step[i]);
    bool yielded = E::FINAL != FINAL_NONE ? d.yielded[i] != 0 : false;
    const bool noisy = PIPE && (d.pipe.act_noise | d.pipe.obs_noise);  // wave-uniform
    uint32_t epi = noisy ? d.ep_idx[i] : 0u;
    if (!AR && (T.flags & VS_FLAG_FREEZE_DONE)) {  // wave-uniform flag
        // rollout() stops at done (rollout.py:185): a lane whose episode has ended keeps its state, observation, step
        // counter and done flag; its reward reads 0 and nothing it is fed can raise the NaN flag
        if (d.done[i] != 0) {
            d.rew[i] = 0.f;
            return;
        }
    }
    float ob_pre[E::O], s_pre[E::S], h_pre[E::H > 0 ? E::H : 1], a_app[E::A];
    if (REC) {
        if (row < 0) row = *d.rec_row;
#pragma unroll
        for (int j = 0; j < E::O; ++j) ob_pre[j] = d.obs[j * ld + i];
    }
    if (REC == 2) {
#pragma unroll
        for (int j = 0; j < E::S; ++j) s_pre[j] = s[j];
#pragma unroll
        for (int j = 0; j < E::H; ++j) h_pre[j] = h[j];
        float alo[E::A], ahi[E::A];
        E::act_bounds(c, alo, ahi);
        applied_action<E>(T, c, alo, ahi, a, a_app);
    }

    StepOut o = step_one<E, float>(T, c, s, h, a, step, yielded, (const float*)nullptr, PIPE ? &d : (const Dev*)nullptr,
                                   i, epi);
    if (REC) {
        if (row < d.traj_rows) {  // (wave-uniform)
            store_record<E, REC>(d.traj_rec + (size_t)row * Rec<E, REC>::F * ld, ld, i, ob_pre, a, o.rew, s_pre, a_app, h_pre);
            uint32_t* w = DoneBits::word(d, i, (size_t)row);  // one step per launch: complete the lane's word of this 32-row window
            const uint32_t bit = 1u << ((unsigned)row & 31u);
            *w = (*w & ~bit) | (o.done ? bit : 0u);
        }
    }

    // (vs_set_lean_step: exactly what SimPyEnv.step returns -- no running return, no failed byte; wave-uniform flag)
    const bool lean = (T.flags & VS_FLAG_LEAN_STEP) != 0;
    float ret = 0.f;
    if (!lean) ret = d.ret[i] + o.rew;
    st_nt<NT>(o.rew, &d.rew[i]);
    d.done[i] = o.done;
    if (!lean) d.failed[i] = o.failed;
    if (o.err && valid) d.err[i] = 1;  // sticky, write-only

    if (AR) {
        bool fin = o.done && valid;
        if (__builtin_amdgcn_ballot_w64(fin) != 0ull) {  // single-step kernel: the per-env counters are touched by finishing lanes only
            EpStat es{0u, 0u, 0.f, 0};
            if (fin) es = EpStat{d.ep_idx[i], d.es_count[i], d.es_retsum[i], d.es_lensum[i]};
            auto_reset<E, UNI, DRK>(T, d, fin, i, seed, c, s, h, step, ret, yielded, es);
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
    for (int j = 0; j < E::S; ++j) st_nt<NT>(s[j], &d.state[j * ld + i]);
#pragma unroll
    for (int j = 0; j < E::H; ++j) st_nt<NT>(h[j], &d.hidden[j * ld + i]);
#pragma unroll
    for (int j = 0; j < E::O; ++j) st_nt<NT>(ob[j], &d.obs[j * ld + i]);
    st_nt<NT>(step, &d.step[i]);
    if (!lean) d.ret[i] = ret;
    if (E::FINAL != FINAL_NONE) d.yielded[i] = yielded;
}

template <class E, bool UNI, bool AR, bool PIPE, int REC, bool DRK = false, bool NT = false>
__global__ __launch_bounds__(BLOCK) void k_step(Task T, Dev d, const float* __restrict__ act, long env_stride,
                                                long dim_stride, uint64_t seed, int row) {
    step_body<E, UNI, AR, PIPE, REC, DRK, NT>(T, d, act, env_stride, dim_stride, seed, (int)blockIdx.x, row);
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

// ---------------------------------------------------------------------------------------------------- rollout kernel
// vs_step_random: rollout() with DummyPolicy (rollout.py:185-239, dummy.py:77-84) -- k env steps per launch, state,
// hidden state and constants stay in registers; only the per-step records stream to HBM when REC.
// Without auto-reset a finished lane freezes (rollout stops at done).
// Actions: Philox4x32-10 keyed by `seed`, counter (env, RNG_ACT, absolute step / SPB); one block feeds SPB = 4 / A
// consecutive steps (the block boundary is wave-uniform because it depends on the launch-global step index only).
template <class E, bool UNI, bool AR, int REC, bool PIPE = false, bool DRK = !UNI>
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
    DoneBits db;
    if (REC) {
        E::observe(s, ob);
        db.begin(d, i, rec0);
    }
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
        float s_pre[E::S], h_pre[E::H > 0 ? E::H : 1], a_app[E::A];  // REC == 2: what rollout() keeps besides obs / act / rew
        if (REC) {
            if (PIPE && d.pipe.obs_on) {  // wave-uniform
                pipe_obs<E>(d, i, es.epi, step, ob, ow);
            } else {
#pragma unroll
                for (int j = 0; j < E::O; ++j) ow[j] = ob[j];
            }
        }
        if (REC == 2) {
#pragma unroll
            for (int j = 0; j < E::S; ++j) s_pre[j] = s[j];
#pragma unroll
            for (int j = 0; j < E::H; ++j) h_pre[j] = h[j];
            applied_action<E>(T, c, alo, ahi, a, a_app);
        }
        if (!frozen) {
            StepOut o = step_one<E, float>(T, c, s, h, a, step, yielded,
                                           (REC && E::TRIG > 0) ? (const float*)(ob + E::TRIG_AT) : (const float*)nullptr,
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
            store_record<E, REC>(d.traj_rec + (rec0 + (size_t)t) * Rec<E, REC>::F * ld, ld, i, ow, a, rew, s_pre, a_app, h_pre);
            db.put(d, i, rec0 + (size_t)t, done, t == k_steps - 1);
        }
        bool fin = done && valid && !frozen;
        if (AR) {
            auto_reset<E, UNI, DRK>(T, d, fin, i, reset_seed, c, s, h, step, ret, yielded, es);
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
    i = VS_COLD(i);  // the epilogue's addresses are the prologue's: recomputed here instead of carried through the loop
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

template <class E, bool UNI, bool AR, int REC, bool PIPE, bool DRK = false>
__global__ __launch_bounds__(BLOCK) void k_rollout(Task T, Dev d, int k_steps, uint64_t seed, uint64_t reset_seed,
                                                   uint64_t epoch0) {
    rollout_body<E, UNI, AR, REC, PIPE, DRK>(T, d, k_steps, seed, reset_seed, epoch0, (int)blockIdx.x);
}

// LDS-only workgroup barrier of the multi-wave kernels below
enum : unsigned { WSF_DONE = 1u, WSF_FAILED = 2u, WSF_FROZEN = 4u, WSF_FIN = 8u };

__device__ __forceinline__ void ws_barrier() {
#ifdef VS_WS_NOSYNC  // diagnostic builds only (timing without the exchange; results are wrong)
    return;
#endif
    // LDS traffic only: wait for this wave's LDS ops (lgkmcnt(0)), not for its global stores (a __syncthreads() would also
    // drain vmcnt and stall the C wave on its record stores in every phase)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// -------------------------------------------------------------------------------------------- policy inside the kernel
// vs_step_policy: rollout() with a feed-forward network policy (P/policies/feed_back/fnn.py:43-160 FNN.forward, called at
// rollout.py:203-219) evaluated INSIDE the fused rollout kernel -- k env steps per launch with a real policy in the loop and
// no launch, no host round trip and no HBM traffic between the policy and the step.
//   act = W_out f(... f(W_1 x + b_1) ...) + b_out  [output_nonlin]  [+ std * N(0, 1): NormalActNoiseExplStrat]
//   x   = the observation rows the policy sees (ObsPartialWrapper: Fnn::obs_idx), optionally in the fork's featurisation
//         [o_0, sin o_1, cos o_1, o_2 ..] (FNNPolicy.forward, fnn.py:219-222)
// Shape of the work on CDNA4: one workgroup = 64 envs x 8 (or 4, fnn_waves) waves, and TWO lane mappings.
//   * The network runs "lane = hidden unit": lane j of every wave keeps unit j's weight rows of all layers in VGPRs for
//     the whole launch (weights stationary: no weight traffic at all inside the step loop), and a wave evaluates the
//     network for its share (8 or 16) of the workgroup's envs one after the other.  An env's activation vector is uniform over
//     the lanes: it is read from LDS as broadcast ds_read_b128 (one address for the whole wave: no bank conflicts), four
//     inputs per instruction, and a layer of 64 x 64 is 16 LDS reads + 64 FMAs per env.  The narrow output layer is a
//     product per lane and a DPP wave reduction.
//   * The envs run "lane = env" on wave 0, with the very step code of k_rollout (same records, same reset path); the
//     two mappings meet in LDS twice per step: observations [env][input] in, actions [env][A] out, one LDS-only barrier each.
// fp32 FMAs on the vector ALU, not MFMA: the fp32 MFMA rate equals the vector rate on this chip, and bf16 MFMA would change
// the policy's numerics (its torch reference is fp32).  A first version with "lane = env" in the network too -- weights as
// scalar loads, units split over the waves -- spent its time waiting for SMEM (7.3 us per step at <= 16 384 envs).
constexpr int FNN_MAXH = 4;   // hidden layers
constexpr int FNN_W = 64;     // padded width of a hidden layer = lanes of a wave
constexpr int FNN_XS = 12;    // padded width of the input row (in_dim <= MAXO + 1 = 9): three 16-B reads
// waves per 64 envs: 8 (two per SIMD, 256 VGPRs each) while one 64-wide hidden-to-hidden weight row set fits beside the env
// state; 4 (one per SIMD, 512 VGPRs) for three and four hidden layers, whose 128 / 192 weight registers would spill
__host__ __device__ constexpr int fnn_waves(int n_hidden) { return n_hidden <= 2 ? 8 : 4; }
enum FnnNonlin { FNN_ID = 0, FNN_TANH = 1, FNN_RELU = 2, FNN_SIGMOID = 3 };
struct Fnn {
    const float* w;  // device, packed by vs_set_policy_fnn: per hidden layer l  Wt_l [in_l][64] (row k = input k, unit-
                     // contiguous, zero padded) at off_w[l] and b_l [64] at off_b[l]; output layer  Wo [A][64] at
                     // off_w[n_hidden], bo [A] at off_b[n_hidden]
    int n_hidden, in_dim, out_dim;
    int hidden[FNN_MAXH];
    int hid_nonlin[FNN_MAXH], out_nonlin;
    int feat;                 // 1: the fork's [o_0, sin o_1, cos o_1, o_2 ..] featurisation (in_dim = visible obs + 1)
    int n_vis;                // number of observation rows the policy sees
    int obs_idx[MAXO];        // ... and which they are
    int ident;                // obs_idx is 0 .. O-1: the policy sees the whole observation
    int noisy;                // any noise_std > 0
    float noise_std[MAXA];
    int off_w[FNN_MAXH + 1], off_b[FNN_MAXH + 1];
};

// tanh through v_exp_f32 / v_rcp_f32: (1 - t) / (1 + t), t = exp(-2|x|); absolute error < 2e-7 (the relative error grows
// below |x| ~ 1e-3, where tanh x ~ x is tiny next to the biases it is added to); NaN stays NaN
__device__ __forceinline__ float tanh_fast(float x) {
    float t = __builtin_amdgcn_exp2f(-2.885390081777927f * fabsf(x));  // exp(-2|x|)
    float r = (1.0f - t) * rcp_fast(1.0f + t);
    return copysignf(r, x);
}
__device__ __forceinline__ float fnn_nonlin(int kind, float x) {  // kind is wave-uniform
    if (kind == FNN_TANH) return tanh_fast(x);
    if (kind == FNN_RELU) return fmaxf(x, 0.f);
    if (kind == FNN_SIGMOID) return rcp_fast(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
    return x;
}

// sum over the 64 lanes of a wave, valid in lane 63: DPP only (quad swaps, row mirrors, row broadcasts) -- nothing through
// the LDS crossbar
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
#define VS_DPP_ADD(ctrl, rmask)                                                                                             \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false))
    VS_DPP_ADD(0xB1, 0xF);   // quad_perm [1, 0, 3, 2]
    VS_DPP_ADD(0x4E, 0xF);   // quad_perm [2, 3, 0, 1]
    VS_DPP_ADD(0x141, 0xF);  // row_half_mirror
    VS_DPP_ADD(0x140, 0xF);  // row_mirror: every lane of a row of 16 holds the row's sum
    VS_DPP_ADD(0x142, 0xA);  // row_bcast15 into rows 1 and 3
    VS_DPP_ADD(0x143, 0xC);  // row_bcast31 into rows 2 and 3
#undef VS_DPP_ADD
    return v;
}

// between an LDS store and the load of what ANOTHER lane of the same wave stored: the hardware runs a wave's LDS instructions in
// order, but to the compiler the two addresses differ and the load may move up -- a wavefront-scope fence pins the order
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// NE = envs per workgroup: 64 (wave 0 owns them) or 256 (waves 0 .. 3 own 64 each, on a SIMD each).  The weight registers allow
// one workgroup per compute unit at a time either way; the larger one spends a smaller share of a step in its serial part
// (env step and barriers) and is the shape for batches beyond 128 envs per compute unit.
// MF: the hidden layers on the matrix cores (v_mfma_f32_32x32x2_f32, fp32 in and out: the policy's numerics stay fp32), for
// 256-env workgroups of 8 waves and one or two hidden layers.  A wave evaluates the network for 32 envs as D = W x X with
// rows = units (two tiles of 32), columns = its 32 envs, K = the layer's inputs two at a time:
//   * A operand (weights): lane l holds W[unit 32u + l % 32][input 2kk + l / 32] -- stationary in VGPRs for the launch;
//   * B operand, first layer: the env's input row from LDS, lane l reads x[env l % 32][2kk + l / 32];
//   * accumulator register r of lane l is D[unit 8 (r / 4) + 4 (l / 32) + r % 4][env l % 32] (scratch/ubench/mfma_layout.hip):
//     taken as the B operand of K-step r of the NEXT layer it supplies inputs (8 (r / 4) + r % 4) and (.. + 4) -- the sum over K
//     does not care about the order, so the next layer's weights are simply loaded in that order and the activations never
//     leave the registers between two layers (bias = the accumulator's initial value, nonlinearity in place);
//   * output layer: 32 products per lane against the weights in accumulator order, the two half-waves added through LDS.
// 64 x 64 hidden-to-hidden: 64 MFMAs of 64 cycles per wave and 32 envs instead of 16 broadcast LDS reads + 64 FMAs per env.
typedef float fnn_acc __attribute__((ext_vector_type(16)));
template <class E, bool AR, int REC, int NHID, int NE, bool MF = false>
__global__ __launch_bounds__(64 * fnn_waves(NHID)) void k_rollout_fnn(Task T, Dev d, Fnn P, int k_steps, uint64_t reset_seed,
                                                                      uint64_t noise_seed) {
    static_assert(NHID >= 1 && NHID <= FNN_MAXH, "hidden layers");
    static_assert(NE == 64 || NE == 256, "envs per workgroup");
    static_assert(NE / 64 <= fnn_waves(NHID), "one wave per 64 envs owns them");
    static_assert(!MF || (NE == 256 && NHID <= 2), "matrix-core path: 8 waves x 32 envs, one or two hidden layers");
    constexpr bool UNI = false;               // per-env constants: Dev::consts is always kept (k_set_params), read once per launch
    constexpr int EPW = NE / fnn_waves(NHID);  // envs a wave evaluates the network for
    __shared__ __attribute__((aligned(16))) float l_x[NE * FNN_XS];  // what the policy sees: [env][input]
    __shared__ __attribute__((aligned(16))) float l_h[MF ? 4 : NE * FNN_W];   // activations of the running layer: [env][unit]
    __shared__ float l_a[NE * MAXA];                                 // the network's output: [env][A]
    __shared__ __attribute__((aligned(16))) float l_bias[MF ? NHID * FNN_W : 4];   // MF: hidden biases and output weights, read
    __shared__ __attribute__((aligned(16))) float l_wo[MF ? MAXA * FNN_W : 4];     //     in accumulator order every step
    __shared__ float l_part[MF ? 64 * fnn_waves(NHID) * 2 : 4];                    //     half-wave partial sums of the output layer
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const bool envw = wave < NE / 64;         // this wave owns 64 of the workgroup's envs
    const int er = (envw ? wave : 0) * 64 + lane;  // (env waves: the lane's env inside the workgroup ...
    const int i = blockIdx.x * NE + er;            //  ... and in the batch)
    const size_t ld = d.ld;
    const size_t rec0 = (size_t)d.traj_t0;
    const bool valid = i < d.n;

    // ---- unit `lane` of every layer: its weight rows, for the whole launch
    float w1[FNN_XS], wh[NHID > 1 ? NHID - 1 : 1][FNN_W], bh[NHID], wo[E::A];
    float a1[2][FNN_XS / 2], a2[2][2][16];  // MF: the A operands of the two layers (see above)
    const int half = lane >> 5, col = lane & 31;
    if constexpr (!MF) {
#pragma unroll
        for (int k = 0; k < FNN_XS; ++k) w1[k] = k < P.in_dim ? P.w[P.off_w[0] + k * FNN_W + lane] : 0.f;
#pragma unroll
        for (int l = 1; l < NHID; ++l) {
#pragma unroll
            for (int k = 0; k < FNN_W; ++k) wh[l - 1][k] = P.w[P.off_w[l] + k * FNN_W + lane];
        }
#pragma unroll
        for (int l = 0; l < NHID; ++l) bh[l] = P.w[P.off_b[l] + lane];
#pragma unroll
        for (int j = 0; j < E::A; ++j) wo[j] = P.w[P.off_w[NHID] + j * FNN_W + lane];
    } else {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int kk = 0; kk < FNN_XS / 2; ++kk) {
                const int k = 2 * kk + half;
                a1[u][kk] = k < P.in_dim ? P.w[P.off_w[0] + k * FNN_W + 32 * u + col] : 0.f;
            }
        }
        if constexpr (NHID > 1) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int up = 0; up < 2; ++up) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int k = 32 * up + 8 * (r / 4) + 4 * half + (r % 4);
                        a2[u][up][r] = P.w[P.off_w[1] + k * FNN_W + 32 * u + col];
                    }
                }
            }
        }
        // biases and output weights into LDS (the first barrier of the step loop comes before their first use)
        if (threadIdx.x < FNN_W) {
#pragma unroll
            for (int l = 0; l < NHID; ++l) l_bias[l * FNN_W + threadIdx.x] = P.w[P.off_b[l] + threadIdx.x];
#pragma unroll
            for (int j = 0; j < E::A; ++j) l_wo[j * FNN_W + threadIdx.x] = P.w[P.off_w[NHID] + j * FNN_W + threadIdx.x];
        }
    }

    // ---- env state of the lane: wave 0 only (the other waves never touch it)
    float c[E::K], s[E::S], h[E::H > 0 ? E::H : 1], a[E::A], ob[E::O];
    float alo[E::A], ahi[E::A];
    int step = 0;
    float ret = 0.f, rew = 0.f;
    bool yielded = false, frozen = false, done = false, failed = false;
    EpStat es{0u, 0u, 0.f, 0};
    DoneBits db;
    db.w = 0u;
    if (envw) {
        load_consts<E, UNI>(d, i, c, 0, E::KS);
#pragma unroll
        for (int j = 0; j < E::S; ++j) s[j] = d.state[j * ld + i];
#pragma unroll
        for (int j = 0; j < E::H; ++j) h[j] = d.hidden[j * ld + i];
        step = d.step[i];
        ret = d.ret[i];
        yielded = E::FINAL != FINAL_NONE ? d.yielded[i] != 0 : false;
        frozen = !AR && d.done[i] != 0;
        rew = d.rew[i];
        done = d.done[i] != 0, failed = d.failed[i] != 0;
        es = EpStat{d.ep_idx[i], d.es_count[i], d.es_retsum[i], d.es_lensum[i]};
        E::act_bounds(c, alo, ahi);
        E::observe(s, ob);
        if (REC) db.begin(d, i, rec0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // (see rollout_body: nothing pending at the loop header)

    for (int t = 0; t < k_steps; ++t) {
        // ---- what the policy sees of obs_t: wave 0 writes its env's input row
        if (envw) {
            float x[FNN_XS];
            if (P.ident) {
#pragma unroll
                for (int k = 0; k < MAXO; ++k) x[k] = k < E::O ? ob[k] : 0.f;
            } else {
#pragma unroll
                for (int k = 0; k < MAXO; ++k) {
                    float v = 0.f;
#pragma unroll
                    for (int j = 0; j < E::O; ++j) v = (k < P.n_vis && P.obs_idx[k] == j) ? ob[j] : v;  // wave-uniform selects
                    x[k] = v;
                }
            }
#pragma unroll
            for (int k = MAXO; k < FNN_XS; ++k) x[k] = 0.f;
            if (P.feat) {  // [o_0, sin o_1, cos o_1, o_2 ..]
                float sn, cs;
                sincos_fast(x[1], &sn, &cs);
#pragma unroll
                for (int k = MAXO; k >= 3; --k) x[k] = x[k - 1];
                x[1] = sn, x[2] = cs;
            }
            Planes<FNN_XS>::store(l_x, 1, er * (FNN_XS / 4), x);  // three 16-B stores: the env's row
        }
        ws_barrier();
        // ---- the network for this wave's envs: lane = unit.  Layer by layer over the wave's EPW envs, in groups of QU envs
        // whose code is one straight-line block (four at a time for one and two hidden layers, two for deeper networks, in a rolled loop:
        // code size).  Every LDS read below is one address for the whole wave (the env index is
        // wave-uniform).  The nonlinearity kind is wave-uniform too: its switch sits outside the group.
        constexpr int QU = NHID <= 2 ? 4 : 2;
        if constexpr (MF) {
            const int e0 = wave * 32;  // this wave's 32 envs
            // (a layer of at most 32 units has an all-zero second tile: zero weights and biases, and the next layer's weights
            // for it are zero too -- it is skipped, wave-uniformly)
            const bool wide0 = P.hidden[0] > 32, wide1 = NHID > 1 && P.hidden[NHID > 1 ? 1 : 0] > 32;
            auto nonlin16 = [&](int kind, fnn_acc* v, bool wide) __attribute__((always_inline)) {  // kind, wide: wave-uniform
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u == 1 && !wide) continue;
                    if (kind == FNN_TANH) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            // 1 - 2 / (1 + exp(2x)) with the bare v_rcp_f32 (1 ulp): mul, exp, add, rcp, fma -- 64 of these per
                            // env step and lane, on the same issue budget as the MFMAs (two instructions less than the
                            // (1 - t) / (1 + t) form of tanh_fast; exp -> inf / 0 gives +-1, NaN stays NaN; same 1e-7 absolute error)
                            const float t = __builtin_amdgcn_exp2f(2.885390081777927f * v[u][r]);
                            v[u][r] = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + t), 1.0f);
                        }
                    } else if (kind != FNN_ID) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[u][r] = fnn_nonlin(kind, v[u][r]);
                    }
                }
            };
            auto bias_init = [&](int l, fnn_acc* v) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b4 = *reinterpret_cast<const float4*>(l_bias + l * FNN_W + 32 * u + 8 * g + 4 * half);
                        v[u][4 * g] = b4.x, v[u][4 * g + 1] = b4.y, v[u][4 * g + 2] = b4.z, v[u][4 * g + 3] = b4.w;
                    }
            };
            fnn_acc h1[2], h2[2];
            bias_init(0, h1);
            float xb1[FNN_XS / 2];
#pragma unroll
            for (int kk = 0; kk < FNN_XS / 2; ++kk) xb1[kk] = l_x[(e0 + col) * FNN_XS + 2 * kk + half];
#pragma unroll
            for (int kk = 0; kk < FNN_XS / 2; ++kk) {
                if (2 * kk >= P.in_dim) continue;  // (wave-uniform: the padding inputs)
                h1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0][kk], xb1[kk], h1[0], 0, 0, 0);
                if (wide0) h1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1][kk], xb1[kk], h1[1], 0, 0, 0);
            }
            nonlin16(P.hid_nonlin[0], h1, wide0);
            if constexpr (NHID > 1) {
                bias_init(1, h2);
#pragma unroll
                for (int up = 0; up < 2; ++up) {
                    if (up == 1 && !wide0) continue;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float xb = h1[up][r];
                        h2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0][up][r], xb, h2[0], 0, 0, 0);
                        if (wide1) h2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[1][up][r], xb, h2[1], 0, 0, 0);
                    }
                }
                nonlin16(P.hid_nonlin[1], h2, wide1);
            }
            fnn_acc* const hl = NHID > 1 ? h2 : h1;
            const bool widel = NHID > 1 ? wide1 : wide0;
#pragma unroll
            for (int j = 0; j < E::A; ++j) {
                float part = 0.f;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (u == 1 && !widel) continue;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 w4 = *reinterpret_cast<const float4*>(l_wo + j * FNN_W + 32 * u + 8 * g + 4 * half);
                        part = fmaf(w4.x, hl[u][4 * g], part);
                        part = fmaf(w4.y, hl[u][4 * g + 1], part);
                        part = fmaf(w4.z, hl[u][4 * g + 2], part);
                        part = fmaf(w4.w, hl[u][4 * g + 3], part);
                    }
                }
                // the other half-wave holds the other 32 units of the same env: add through LDS (one wave, in order)
                l_part[(wave * 64 + lane) * 2 + (j & 1)] = part;
                wave_lds_fence();
                const float other = l_part[(wave * 64 + (lane ^ 32)) * 2 + (j & 1)];
                if (half == 0) l_a[(e0 + col) * MAXA + j] = part + other;
            }
        }
        auto with_kind = [&](int kind, auto&& body) __attribute__((always_inline)) {
            if (kind == FNN_TANH) body(std::integral_constant<int, FNN_TANH>{});
            else if (kind == FNN_RELU) body(std::integral_constant<int, FNN_RELU>{});
            else if (kind == FNN_SIGMOID) body(std::integral_constant<int, FNN_SIGMOID>{});
            else body(std::integral_constant<int, FNN_ID>{});
        };
        // what becomes of unit `lane`'s activation of env e: the next layer's input row (in place: the row has been read
        // completely by then), or -- behind the last hidden layer -- the output layer: a product per unit and a wave reduction
        // (padding units carry zero weights)
        auto emit = [&](auto last, int e, float hv) __attribute__((always_inline)) {
            if constexpr (decltype(last)::value) {
#pragma unroll
                for (int j = 0; j < E::A; ++j) {
                    const float sum = wave_sum_to_lane63(wo[j] * hv);
                    if (lane == 63) l_a[e * MAXA + j] = sum;
                }
            } else {
                l_h[e * FNN_W + lane] = hv;
            }
        };
        if constexpr (!MF) with_kind(P.hid_nonlin[0], [&](auto kind) __attribute__((always_inline)) {
#pragma unroll 1
            for (int g = 0; g < EPW / QU; ++g) {
                const int e0 = wave * EPW + g * QU;
                float x[3][FNN_XS];  // input rows two envs ahead of the arithmetic
                Planes<FNN_XS>::load(l_x, 1, e0 * (FNN_XS / 4), x[0]);
                Planes<FNN_XS>::load(l_x, 1, (e0 + 1) * (FNN_XS / 4), x[1]);
#pragma unroll
                for (int q = 0; q < QU; ++q) {
                    if (q + 2 < QU) Planes<FNN_XS>::load(l_x, 1, (e0 + q + 2) * (FNN_XS / 4), x[(q + 2) % 3]);
                    float acc = bh[0];
#pragma unroll
                    for (int k = 0; k < FNN_XS; ++k) acc = fmaf(w1[k], x[q % 3][k], acc);
                    emit(std::integral_constant<bool, NHID == 1>{}, e0 + q, fnn_nonlin(decltype(kind)::value, acc));
                }
            }
        });
        auto hidden_layer = [&](auto lc) __attribute__((always_inline)) {
            constexpr int l = decltype(lc)::value;
            with_kind(P.hid_nonlin[l], [&](auto kind) __attribute__((always_inline)) {
                // a row of 64 inputs is four quarters of four 16-B reads; the reads of quarter n + 2 are issued before the 16
                // FMAs of quarter n (a ring of three register buffers), across the envs of a group: the LDS latency hides
                // behind the arithmetic
#pragma unroll 1
                for (int g = 0; g < EPW / QU; ++g) {
                    const int e0 = wave * EPW + g * QU;
                    constexpr int NQ4 = 4 * QU;
                    float4 buf[3][4];
                    auto issue = [&](int n, float4* dst) __attribute__((always_inline)) {
                        const float4* row = reinterpret_cast<const float4*>(l_h + (e0 + n / 4) * FNN_W) + (n & 3) * 4;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) dst[kk] = row[kk];
                    };
                    issue(0, buf[0]);
                    issue(1, buf[1]);
                    float acc = bh[l];
#pragma unroll
                    for (int n = 0; n < NQ4; ++n) {
                        if (n + 2 < NQ4) issue(n + 2, buf[(n + 2) % 3]);
                        const float4* x4 = buf[n % 3];
                        const int k0 = (n & 3) * 16;
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            acc = fmaf(wh[l - 1][k0 + 4 * kk], x4[kk].x, acc);
                            acc = fmaf(wh[l - 1][k0 + 4 * kk + 1], x4[kk].y, acc);
                            acc = fmaf(wh[l - 1][k0 + 4 * kk + 2], x4[kk].z, acc);
                            acc = fmaf(wh[l - 1][k0 + 4 * kk + 3], x4[kk].w, acc);
                        }
                        if ((n & 3) == 3) {
                            emit(std::integral_constant<bool, l == NHID - 1>{}, e0 + n / 4, fnn_nonlin(decltype(kind)::value, acc));
                            acc = bh[l];
                        }
                    }
                }
            });
        };
        if constexpr (!MF && NHID > 1) hidden_layer(std::integral_constant<int, 1>{});
        if constexpr (!MF && NHID > 2) hidden_layer(std::integral_constant<int, 2>{});
        if constexpr (!MF && NHID > 3) hidden_layer(std::integral_constant<int, 3>{});
        ws_barrier();
        if (!envw) continue;
        // ---- the policy's action of this lane's env (+ exploration noise)
#pragma unroll
        for (int j = 0; j < E::A; ++j) a[j] = fnn_nonlin(P.out_nonlin, l_a[er * MAXA + j] + P.w[P.off_b[NHID] + j]);
        if (P.noisy) {
            // NormalActNoiseExplStrat: + std * N(0, 1), keyed like the wrapper noise by (env, episode, step)
            uint4 b = Rng::philox(noise_seed, d.idx0 + (uint32_t)i, RNG_POLICY_NOISE, ((uint64_t)es.epi << 32) | (uint32_t)step);
            float z[2];
            Rng::box_muller(b.x, b.y, z[0], z[1]);
#pragma unroll
            for (int j = 0; j < E::A; ++j) a[j] = fmaf(P.noise_std[j], z[j], a[j]);
        }
        // ---- the env step: rollout_body's, statement for statement
        float s_pre[E::S], h_pre[E::H > 0 ? E::H : 1], a_app[E::A], ow[E::O];
#pragma unroll
        for (int j = 0; j < E::O; ++j) ow[j] = ob[j];
        if (REC == 2) {
#pragma unroll
            for (int j = 0; j < E::S; ++j) s_pre[j] = s[j];
#pragma unroll
            for (int j = 0; j < E::H; ++j) h_pre[j] = h[j];
            applied_action<E>(T, c, alo, ahi, a, a_app);
        }
        if (!frozen) {
            StepOut o = step_one<E, float>(T, c, s, h, a, step, yielded,
                                           E::TRIG > 0 ? (const float*)(ob + E::TRIG_AT) : (const float*)nullptr);
            rew = o.rew;
            done = o.done;
            failed = o.failed;
            ret += o.rew;
            if (o.err && valid) d.err[i] = 1;
        } else {
            rew = 0.f;
        }
        if (REC) {
            store_record<E, REC>(d.traj_rec + (rec0 + (size_t)t) * Rec<E, REC>::F * ld, ld, i, ow, a, rew, s_pre, a_app, h_pre);
            db.put(d, i, rec0 + (size_t)t, done, t == k_steps - 1);
        }
        bool fin = done && valid && !frozen;
        if (AR) {
            auto_reset<E, UNI>(T, d, fin, i, reset_seed, c, s, h, step, ret, yielded, es);
            E::act_bounds(c, alo, ahi);
        } else {
            if (fin) {
                es.count += 1u;
                es.retsum += ret;
                es.lensum += step;
            }
            if (d.log_episodes) append_episode(d, fin, i, ret, step);
            frozen |= done;
        }
        E::observe(s, ob);
    }
    if (!envw) return;
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
// NE = envs per workgroup (a workgroup is NE / 64 P waves followed by NE / 64 C waves):
//   256  512 threads; wave w and w + 4 share a SIMD, so every SIMD of the CU holds one P and one C wave -- the shape for
//        one workgroup per compute unit (65 536 envs on 256 CUs)
//    64  128 threads, one wave of each role on SIMDs of their own -- for batches that cannot fill the chip with 256-env
//        workgroups (4 096 envs are 16 of those, on 16 of 256 CUs, but 64 of these)
// The message P -> C of one step is  s_t | E::observe_p(s_t) (records only) | h_t (REC == 2) | a_t (DP) | flags : the C wave
// finishes the observation (E::observe_c: for QQube the sin / cos of theta, which the dynamics never need) and P keeps only
// the trig its next step reuses.
// DP ("draw on P"): the policy's action generator runs on the physics wave and the action travels in the message instead of
// through l_act -- for the families whose C wave is the longer one once it records (QQube: its share of observe() plus the
// record stores outweigh the Philox block per four steps; measured with the per-role cycle stamps of -DVS_WS_STAMP).
// Occupancy is part of the design: amdgpu_waves_per_eu(min, max) -- min: the register budget (three roles: three waves per
// SIMD); max (0 = none): a family with E::WS_ALONE runs its 64-env workgroups ONE wave per SIMD (the kernel descriptor is
// padded to 257 VGPRs): its physics wave is by far the long one and loses 15 % when the hardware, free to do so since the
// kernel needs only 136 registers, puts both waves of a workgroup on one SIMD and leaves another idle
// (scratch/ubench/place2.hip: 512 workgroups of 128 threads land as [PC][--][P-][-C] per compute unit, [P][C][P][C] when padded).
// DRK ("domain randomisation compiled in"): the handle has a live randomizer or a parameter buffer, i.e. a reset inside the
// launch redraws the lane's domain parameters.  The kernels without it (the headline's, every nominal-parameter batch) carry
// none of that code -- the redraw (Philox, Box-Muller, the parameter select chains, _calc_constants) was ~1 000 instructions in
// each of the eight inlined reset blocks of a kernel that never executes it.
//   DRK = 0 none, 1 a live randomizer (DomainRandWrapperLive), 2 a parameter buffer (DomainRandWrapperBuffer; or both)
template <class E, bool UNI, bool AR, int REC, int WS_R, int NE, bool DP, int NR = 2, int DRK = 0>
__global__ __launch_bounds__(NR * NE)
__attribute__((amdgpu_waves_per_eu(NR == 3 ? 3 : E::WS_MIN_WAVES, (NE == 64 && NR == 2 && E::WS_ALONE) ? 1 : 0))) void k_rollout_ws(Task T, Dev d, int k_steps, uint64_t seed, uint64_t reset_seed,
                                                        uint64_t epoch0) {
    static_assert(DRK == 0 || (!UNI && AR), "a randomizer needs per-env constants and resets inside the launch");
    static_assert(DRK >= 0 && DRK <= 2, "none | randomizer | buffer");
    static_assert(E::FINAL != FINAL_STATE_TIME, "needs the post-step state on the reward side");
    static_assert(NE == 64 || NE == 128 || NE == 256, "envs per workgroup");
    static_assert(NR == 2 || (NR == 3 && !DP), "two roles, or three with the generator wave drawing the actions");
    // GOBS: the generator wave (NR == 3) also finishes the observation and stores the record's first plane (the first four
    // observation floats); the C wave then stores the remaining planes only
    constexpr bool G3 = NR == 3;
#ifdef VS_G_NOOBS  // diagnostic builds only
    constexpr bool GOBS = false;
#else
    constexpr bool GOBS = G3 && REC != 0 && E::O >= 4;
#endif
    // DP draws a batch of actions ahead with the action bounds of that moment: fine, because the bounds change only with a
    // redraw of the domain parameters at a reset, and a family whose bounds depend on them (REWARD_SIDE_USES_CONSTS) never
    // runs this kernel under live randomisation (Launch<E>::variant)
    constexpr int HM = REC == 2 ? E::H : 0;                      // hidden state travels for the full records only
    constexpr int AM = DP ? E::A : 0;                            // the action travels in the message when P draws it
    constexpr int M0 = E::S + (REC ? E::TRIG : 0) + HM + AM + 1;  // message of one step
    constexpr int M = M0 % 4 == 3 ? M0 + 1 : M0;                 // 4k + 3 floats would be three LDS ops for the tail; pad to a quad
    constexpr int TR0 = E::S, HM0 = E::S + (REC ? E::TRIG : 0), AM0 = HM0 + HM;  // offsets inside the message
    constexpr int NT = E::TRIG > 0 ? E::TRIG : 1, NH = E::H > 0 ? E::H : 1;
    __shared__ __attribute__((aligned(16))) float l_msg[2][WS_R][M * NE];
    // PC ("prep on C"): the C wave, which draws the actions, also runs ActNorm -> clip -> dead zone on them and hands the P
    // wave the voltages that reach the dynamics, next to the raw action it keeps for its own reward / record
    // 0: no, 1: ActNorm -> clip, 2: + dead zone (see EnvDefaults::WS_PREP_C / WS_PREP_G64)
#ifdef VS_PREP_G256  // (experiment: also in the 256-env shape -- QQube at 65 536 envs - 2 %, oscillator - 1.5 %, pendulum + 1 %: there the generator
                    // wave is the busiest of the three already)
    constexpr int PCL = DP ? 0 : (E::WS_PREP_C > 0 ? E::WS_PREP_C : ((NR == 3 && DRK == 0) ? E::WS_PREP_G64 : 0));
#else
    constexpr int PCL = DP ? 0 : (E::WS_PREP_C > 0 ? E::WS_PREP_C : ((NR == 3 && NE == 64 && DRK == 0) ? E::WS_PREP_G64 : 0));
#endif
    constexpr bool PC = PCL > 0;
    static_assert(PCL != 2 || E::REWARD_SIDE_USES_CONSTS || DRK == 0, "a dead zone off the physics wave reads per-env constants: they must not change inside the launch");
    constexpr int AW = PC ? 2 * E::A : E::A;  // floats per step in l_act: [u | a] or [a]
    // two buffers (the C wave draws batch b + 1 after it has worked batch b - 1 off), three when a wave of its own draws:
    // in phase b the G wave writes batch b + 1 while P reads batch b and C still reads the raw actions of batch b - 1
    __shared__ __attribute__((aligned(16))) float l_act[G3 ? 3 : (DP ? 1 : 2)][DP ? 1 : WS_R][DP ? 4 : AW * NE];
    auto ab = [](int bb) __attribute__((always_inline)) { return DP ? 0 : (G3 ? bb % 3 : (bb & 1)); };
    // The reset stock (auto-reset only): the C wave keeps, per lane, the init-space sample of the lane's NEXT episode (and
    // the trig of that state) ready in LDS, tagged with the episode counter it was drawn for.  A resetting lane of the P
    // wave takes it with a handful of LDS reads instead of running Philox + sample_init + observe_p for the one or two
    // lanes of the wave that reset; C refills the consumed entries every WS_REFILL batches, for all of them at once (many
    // lanes per pass instead of one pass per event).  A lane that resets again before its entry was refilled (tag !=
    // counter), and every reset under live randomisation (the init space may depend on the redrawn parameters), draws for
    // itself as before -- the values are the same either way (same Philox counters).
    constexpr bool STOCK = AR;
    constexpr int SKW = E::I + (REC ? E::TRIG : 0);
#ifdef VS_WS_REFILL  // (experiments)
    constexpr int WS_REFILL = VS_WS_REFILL;
#else
    constexpr int WS_REFILL = 8;
#endif
    __shared__ float l_stock[STOCK ? SKW * NE : 1];
    __shared__ uint32_t l_stag[STOCK ? NE : 1];
    __shared__ uint32_t l_epi[(STOCK && G3) ? NE : 1];  // G3: the P wave publishes a lane's episode counter at its resets
    // Under a live randomizer (DomainRandWrapperLive: the parameters are redrawn at every reset) the stock entry also holds the
    // NEXT episode's domain parameters and the constants _calc_constants derives from them: the redraw (Philox + Box-Muller per
    // parameter + calc_consts: ~11 000 cycles when the P wave runs it for the one or two lanes of its 64 that reset, measured
    // with -DVS_WS_STAMP) leaves the P wave's critical path like the init-space sample did.  Dynamic LDS, sized by the
    // launcher only when a randomizer is set:
    //   [P][NE] the parameters of the entry (a redraw overwrites the randomised ones in place; the others are the lane's
    //   parameters at the start of the launch and never change) | [K][NE] the constants calc_consts derives from them
    extern __shared__ float l_dyn[];
    // (wave-uniform; constants at compile time for DRK = 0, and DRK = 1 says there is no parameter buffer)
    const bool dr_stock = DRK == 1 ? (STOCK && !UNI && d.dr_n > 0) : (DRK == 2 && STOCK && !UNI && d.dr_n > 0 && d.pbuf_n == 0);
    const bool stock_on = STOCK && (DRK == 0 || (DRK == 1 ? true : (d.pbuf_n == 0 && (d.dr_n == 0 || dr_stock))));
    float* const l_npar = l_dyn;
    float* const l_ncon = l_dyn + E::P * NE;
    const int wave = threadIdx.x >> 6;
    const int role = wave / (NE / 64);  // 0 P, 1 C, 2 G (NR == 3)
    const int le = threadIdx.x & (NE - 1);  // env slot inside the workgroup
    const int i = blockIdx.x * NE + le;
    const int le_ = le, i_ = i;             // (for the rare blocks that shadow the two with laundered copies)
    const size_t ld = d.ld;
    const bool valid = i < d.n;
    const int nb = (k_steps + WS_R - 1) / WS_R;
    float c[E::K];
    load_consts<E, UNI>(d, i, c, 0, E::KS);
    float alo[E::A], ahi[E::A];
    E::act_bounds(c, alo, ahi);

    // The action stream (that of k_rollout): step ta takes words (ta % SPB) * A .. of block Philox(seed; env, RNG_ACT,
    // ta / SPB).  A batch of WS_R steps spans NBLK blocks from a phase ph = epoch0 % SPB that is the same for every batch:
    // with the last block of the previous batch carried over, every batch computes exactly NBLK new blocks,
    // unconditionally -- no step-dependent branch inside the draw.  Run by the C wave, or by the P wave when DP.
    constexpr unsigned SPB = 4 / E::A;
    static_assert(WS_R % SPB == 0, "a batch is a whole number of Philox blocks");
    constexpr int NBLK = WS_R / SPB;
    const unsigned ph = (unsigned)(epoch0 % SPB);
    const uint64_t blk0 = epoch0 / SPB;
    uint4 carry = make_uint4(0, 0, 0, 0);
    // the actions of batch bb: act_space.sample_uniform() per step (all WS_R of them: a ragged last batch ignores the rest)
    auto draw_batch = [&](int bb, float (*a_out)[AW]) __attribute__((always_inline)) {
        uint32_t w[(NBLK + 1) * 4];
        w[0] = carry.x, w[1] = carry.y, w[2] = carry.z, w[3] = carry.w;
#pragma unroll
        for (int q = 1; q <= NBLK; ++q) {
            carry = Rng::philox(seed, d.idx0 + (uint32_t)i, RNG_ACT, blk0 + (uint64_t)bb * NBLK + (uint64_t)q);
            w[4 * q] = carry.x, w[4 * q + 1] = carry.y, w[4 * q + 2] = carry.z, w[4 * q + 3] = carry.w;
        }
#pragma unroll
        for (int r = 0; r < WS_R; ++r) {
#pragma unroll
            for (int j = 0; j < E::A; ++j) {
                // word (ph + r) * A + j of w[]: ph is wave-uniform, so this is a chain of scalar-conditioned selects
                uint32_t bits = w[r * E::A + j];
#pragma unroll
                for (unsigned p2 = 1; p2 < SPB; ++p2) bits = ph == p2 ? w[(r + p2) * E::A + j] : bits;
                bool nrm = (T.flags & VS_FLAG_ACT_NORM) != 0;
                a_out[r][(PC ? E::A : 0) + j] = E::sample_action(c, nrm ? -1.0f : alo[j], nrm ? 1.0f : ahi[j], Rng::to_u01(bits), j);
            }
            if constexpr (PC) {
                // ActNorm -> limit_act -> dead zone, statement for statement what the P wave (and step_one) would do
                const float* a = a_out[r] + E::A;
                float an[E::A], ac[E::A];
                const bool nrm = (T.flags & VS_FLAG_ACT_NORM) != 0;
                float lb[E::A], ub[E::A];
                E::act_bounds(c, lb, ub);
#pragma unroll
                for (int j = 0; j < E::A; ++j) {
                    float m = lb[j] + (a[j] + 1.0f) * (ub[j] - lb[j]) * 0.5f;
                    an[j] = vsel(nrm, m, a[j]);
                }
                E::limit_act(c, alo, ahi, an, ac);
                if constexpr (PCL == 2) E::dead_zone(T, c, ac);
#pragma unroll
                for (int j = 0; j < E::A; ++j) a_out[r][j] = ac[j];
            }
        }
    };

    // The stock refill and the draw into l_act: run by the C wave, or by the G wave when there is one (NR == 3).
    uint32_t c_epi = 0u;           // the lane's episode counter as the refilling wave has seen it advance
    uint32_t c_tag = 0xFFFFFFFFu;  // the counter the lane's stock entry was drawn for (none yet)
    auto refill = [&]() __attribute__((always_inline)) {
        if (!STOCK) return;
        if (G3) c_epi = __hip_atomic_load(&l_epi[le], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const bool need = stock_on && valid && c_tag != c_epi;
        // Under a live randomizer a pass costs the wave that runs it ~9 000 cycles whether one lane needs an entry or all 64 do (every
        // lane walks through the draws, the stores are masked), and the physics wave ends up waiting for that wave: a pass for
        // one or two lanes costs more than the stock misses it prevents.  So a pass waits until WS_REFILL_MIN lanes need one
        // (the first fill of a launch always runs: every lane does).  Measured, cartpole + 7 parameters, 65 536 envs, 400 steps
        // per launch: a pass every 8 batches for any lane 9.58e10; every 32 batches 1.0e11; first fill only 1.032e11; every 8
        // batches once 8 lanes wait 1.033e11 (and 4: 1.02e11, 16: 1.033e11) -- which, unlike "never", still refills in long launches.
#ifdef VS_WS_REFILL_MIN  // (experiments)
        constexpr int REFILL_MIN = DRK == 1 ? VS_WS_REFILL_MIN : 1;
#else
        constexpr int REFILL_MIN = DRK == 1 ? 8 : 1;
#endif
        if (__builtin_popcountll(__builtin_amdgcn_ballot_w64(need)) < REFILL_MIN) return;
        const int i = VS_COLD(i_), le = VS_COLD(le_);  // shadow the kernel's: see cold_lane
        if (dr_stock) {
            // DomainRandWrapperLive.reset of episode c_epi: DomainRandomizer.randomize's draws in their order
            // (domain_parameter.py:104-132), statement for statement redraw_lane_params', straight into the entry
            Rng gp(reset_seed, d.idx0 + (uint32_t)i, RNG_PARAM, (uint64_t)c_epi);
            const int n = d.dr_n;
            for (int q = 0; q < n; ++q) {
                const vs_dp_spec sp = d.drv.s[q];
                const float v = draw_one_param(sp, gp);  // (every lane draws: only the stores are masked)
                if (need) l_npar[sp.param_index * NE + le] = v;
            }
        }
        if (need) {
            float cs[E::K];  // the constants the init space of episode c_epi is sampled with
#pragma unroll
            for (int k = 0; k < E::K; ++k) cs[k] = k < E::KS ? c[k] : 0.f;
            if (dr_stock) {
                float p[E::P];
#pragma unroll
                for (int k = 0; k < E::P; ++k) p[k] = l_npar[k * NE + le];
                E::calc_consts(T, p, cs);
#pragma unroll
                for (int k = 0; k < E::K; ++k) l_ncon[k * NE + le] = cs[k];
            }
            // SimPyEnv.reset's init_space.sample_uniform() of episode c_epi: the draw reset_lane_sampled would make
            Rng g(reset_seed, d.idx0 + (uint32_t)i, RNG_INIT, (uint64_t)c_epi);
            float init[E::I];
            E::sample_init(T, cs, g, init);
#pragma unroll
            for (int j = 0; j < E::I; ++j) l_stock[j * NE + le] = init[j];
            if (REC && E::TRIG > 0) {
                float s0[E::S], tr0[NT];
                E::state_from_init(init, s0);
                E::observe_p(s0, tr0);
#pragma unroll
                for (int j = 0; j < E::TRIG; ++j) l_stock[(E::I + j) * NE + le] = tr0[j];
            }
            __hip_atomic_store(&l_stag[le], c_epi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            c_tag = c_epi;
        }
    };
    auto draw = [&](int bb) __attribute__((always_inline)) {
        if (DP) return;
        float a_new[WS_R][AW];
        draw_batch(bb, a_new);
#pragma unroll
        for (int r = 0; r < WS_R; ++r) Planes<AW>::store(l_act[ab(bb)][DP ? 0 : r], NE, le, a_new[r]);
    };

    if (role == 0) {
        // ------------------------------------------------------------------------------------------- P wave
        // the long wave of most families and the one the others wait for at the barrier: it issues first where it shares a SIMD
        // (measured: BASELINE config 3 + 6 %, every other shape within +- 1 %)
        __builtin_amdgcn_s_setprio(3);
        float s[E::S], h[NH], tr[NT], ob[E::O];
#pragma unroll
        for (int j = 0; j < E::S; ++j) s[j] = d.state[j * ld + i];
#pragma unroll
        for (int j = 0; j < E::H; ++j) h[j] = d.hidden[j * ld + i];
        int step = d.step[i];
        uint32_t epi = d.ep_idx[i];
        bool frozen = !AR && d.done[i] != 0;
        bool done = d.done[i] != 0, failed = d.failed[i] != 0;
        bool err_acc = false;  // the NaN flag is sticky and write-only: one store after the loop instead of a branch per step
        if (REC) E::observe_p(s, tr);
        if (DP) carry = Rng::philox(seed, d.idx0 + (uint32_t)i, RNG_ACT, blk0);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        ws_barrier();  // the actions of batch 0 are in l_act[0] (!DP)
        // one step of the P side: a = the step's action(s) as l_act / draw_batch hands them over, msg = its row of l_msg
        auto p_step = [&](const float* a, float* msg) __attribute__((always_inline)) {
            {
                float v[M];
#pragma unroll
                for (int j = 0; j < E::S; ++j) v[j] = s[j];
                if (REC) {
#pragma unroll
                    for (int j = 0; j < E::TRIG; ++j) v[TR0 + j] = tr[j];
                }
#pragma unroll
                for (int j = 0; j < HM; ++j) v[HM0 + j] = h[j];
#pragma unroll
                for (int j = 0; j < AM; ++j) v[AM0 + j] = a[j];
                bool fin = false;  // (used for the reset on this side)
                if (!frozen) {
                    // the P half of step_one: ActNorm -> limit_act -> _step_dynamics -> curr_step += 1 -> is_done
                    if constexpr (PCL == 2) {
                        E::dynamics_core(T, c, s, h, a, (REC && E::TRIG > 0) ? (const float*)tr : (const float*)nullptr);
                    } else if constexpr (PCL == 1) {
                        E::dynamics(T, c, s, h, a, (REC && E::TRIG > 0) ? (const float*)tr : (const float*)nullptr);
                    } else {
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
                        E::limit_act(c, alo, ahi, an, ac);
                        E::dynamics(T, c, s, h, ac, (REC && E::TRIG > 0) ? (const float*)tr : (const float*)nullptr);
                    }
                    // the trig of the new state right behind the dynamics, in the same basic block as the bounds test and the
                    // message below: independent work for the scheduler to interleave (a lane that resets redoes it)
                    if (REC) E::observe_p(s, tr);
                    step += 1;
                    float slo[E::S], shi[E::S];
                    E::state_bounds(c, slo, shi);
                    failed = false;
#pragma unroll
                    for (int j = 0; j < E::S; ++j) {
                        if (!E::SYMMETRIC_BOX) failed |= (s[j] < slo[j]) | (s[j] > shi[j]);
                    }
                    if (E::SYMMETRIC_BOX) failed = outside_symmetric_box<E::S>(s, shi);
                    done = failed | (step >= T.max_steps);
                    // No per-step NaN test on this wave: a NaN action (clip and dead zone keep it) or state makes the state NaN
                    // in every family, a NaN state is inside no box test (never `failed`) and stays NaN until the lane is reset
                    // by its time-out or the launch ends -- the two places where the sticky flag is raised below.
                    fin = done && valid;
                }
                unsigned fl = (done ? WSF_DONE : 0u) | (E::FINAL != FINAL_NONE && failed ? WSF_FAILED : 0u) |
                              (!AR && frozen ? WSF_FROZEN : 0u);  // fin = done & !frozen & valid is recomputed by C
                v[M0 - 1] = __uint_as_float(fl);
                if (M > M0) v[M - 1] = 0.f;
                Planes<M>::store(msg, NE, le, v);
                if (AR) {
                    if (__builtin_amdgcn_ballot_w64(fin) != 0ull) {
                        VS_STAMP(sr0);
                        const int i = VS_COLD(i_), le = VS_COLD(le_);  // shadow the kernel's: see cold_lane
                        if (fin) {
#pragma unroll
                            for (int j = 0; j < E::S; ++j) err_acc |= isnan(s[j]);
                            load_consts<E, UNI>(d, i, c, E::KS, E::K);
                            bool stocked = false;
                            if (STOCK) {
                                const uint32_t tag = __hip_atomic_load(&l_stag[le], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                stocked = tag == epi;
                                if (stocked) {
                                    if (dr_stock) {
                                        // the redrawn parameters and their constants, to where redraw_lane_params puts them
#pragma unroll
                                        for (int k = 0; k < E::K; ++k) c[k] = l_ncon[k * NE + le];
#pragma unroll
                                        for (int k = 0; k < E::P; ++k) d.params[(size_t)k * ld + i] = l_npar[k * NE + le];
#pragma unroll
                                        for (int k = 0; k < E::K; ++k) d.consts[(size_t)k * ld + i] = c[k];
                                    }
                                    float init[E::I];
#pragma unroll
                                    for (int j = 0; j < E::I; ++j) init[j] = l_stock[j * NE + le];
                                    if (REC) {
#pragma unroll
                                        for (int j = 0; j < E::TRIG; ++j) tr[j] = l_stock[(E::I + j) * NE + le];
                                    }
                                    E::state_from_init(init, s);
                                    E::init_hidden(T, c, nullptr, s, h, false);
                                }
                            }
#ifdef VS_WS_STAMP  // resets served from the stock (low word) / drawn on this wave (high word), summed over the launches
                            if (d.dbg) atomicAdd(d.dbg + ((size_t)(i >> 6) * 3 + 0) * 4, stocked ? 1ull : (1ull << 32));
#endif
                            if (!stocked) {
                                // live domain randomisation redraws the lane's parameters here: allowed for the families
                                // whose C wave does not read constants (use_ws)
                                reset_lane_sampled<E>(T, d, DRK != 0, i, reset_seed, (uint64_t)epi, c, s, h,
                                                      dr_stock ? (const float*)l_npar : (const float*)nullptr, NE, le, DRK == 2);
                                if (REC) E::observe_p(s, tr);
                            }
                            epi += 1u;
                            step = 0;
                            // (after the reads of the stock entry above: the G wave rewrites an entry only once it has seen
                            // a counter beyond the entry's tag)
                            if (STOCK && G3) __hip_atomic_store(&l_epi[le], epi, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        if (!UNI) E::act_bounds(c, alo, ahi);
#ifdef VS_WS_STAMP  // cycles of the reset branch of this wave, and how often it ran (slots of the unused third role when NR == 2)
                        VS_STAMP(sr1);
                        if (d.dbg && (threadIdx.x & 63) == 0) {
                            atomicAdd(d.dbg + ((size_t)(i >> 6) * 3 + 2) * 4 + 0, sr1 - sr0);
                            atomicAdd(d.dbg + ((size_t)(i >> 6) * 3 + 2) * 4 + 1, 1ull);
                        }
#endif
                    }
                } else {
                    frozen |= done;
                }
            }
        };
        // A full batch: all WS_R steps exist -- the actions of the batch up front (one LDS round trip per batch instead of one per
        // step on the critical path), the steps unrolled: as few basic blocks as the rare reset branch allows.
        auto p_batch_full = [&](int b) __attribute__((always_inline)) {
            float a_all[WS_R][AW];
            if (DP) {
                draw_batch(b, a_all);
            } else {
#pragma unroll
                for (int r = 0; r < WS_R; ++r) Planes<AW>::load(l_act[ab(b)][DP ? 0 : r], NE, le, a_all[r]);
            }
#pragma unroll
            for (int r = 0; r < WS_R; ++r) p_step(a_all[r], l_msg[b & 1][r]);
        };
        // The ragged last batch of a launch whose step count is not a multiple of WS_R: ONE rolled copy of the step (round 2 kept a
        // second, guarded copy of the unrolled batch: with its four inlined reset blocks it doubled the code of this wave's loop
        // for a batch that runs at most once per launch)
        auto p_batch_ragged = [&](int b, int nr) __attribute__((always_inline)) {
            float a_all[WS_R][AW];
            if (DP) draw_batch(b, a_all);
#pragma unroll 1
            for (int r = 0; r < nr; ++r) {
                float a[AW];
                if (DP) {
#pragma unroll
                    for (int j = 0; j < AW; ++j) {
                        a[j] = a_all[0][j];
#pragma unroll
                        for (int q = 1; q < WS_R; ++q) a[j] = r == q ? a_all[q][j] : a[j];  // (r is wave-uniform)
                    }
                } else {
                    Planes<AW>::load(l_act[ab(b)][DP ? 0 : r], NE, le, a);
                }
                p_step(a, l_msg[b & 1][r]);
            }
        };
#ifdef VS_WS_NOP  // diagnostic: the P wave only keeps the barriers
        for (int b = 0; b < nb; ++b) ws_barrier();
        if (false)
#endif
#ifdef VS_WS_STAMP
        unsigned long long acc0 = 0, acc1 = 0, acc2 = 0;
#endif
        for (int b = 0; b < nb; ++b) {
            const int nr = min(WS_R, k_steps - b * WS_R);
            VS_STAMP(st0);
            if (nr == WS_R) p_batch_full(b);
            else p_batch_ragged(b, nr);
            VS_STAMP(st2);
            ws_barrier();
#ifdef VS_WS_STAMP
            VS_STAMP(st3);
            acc1 += st2 - st0, acc2 += st3 - st2;
#endif
        }
#ifdef VS_WS_STAMP
        if ((threadIdx.x & 63) == 0 && d.dbg) {
            unsigned long long* q = d.dbg + ((size_t)(i >> 6) * 3 + 0) * 4;
            q[1] = acc1, q[2] = acc2, q[3] = (unsigned long long)nb;  // (q[0]: the reset counters, accumulated over the launches)
        }
#endif
        const int i = VS_COLD(i_);  // (see cold_lane: the epilogue's addresses are not carried through the loop)
#pragma unroll
        for (int j = 0; j < E::S; ++j) err_acc |= isnan(s[j]);
        if (err_acc && valid) d.err[i] = 1;
        E::observe(s, ob);
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
    } else if (role == 1) {
        // ------------------------------------------------------------------------------------------- C wave
        const size_t rec0 = (size_t)d.traj_t0;
        float ret = d.ret[i];
        float rew = d.rew[i];
        bool yielded = E::FINAL != FINAL_NONE ? d.yielded[i] != 0 : false;
        EpStat es{0u, d.es_count[i], d.es_retsum[i], d.es_lensum[i]};
        int len = d.step[i];
        DoneBits db;
        if (REC) db.begin(d, i, rec0);
        if (STOCK && !G3) {
            c_epi = d.ep_idx[i];
            l_stag[le] = 0xFFFFFFFFu;
            if (dr_stock) {
#pragma unroll
                for (int k = 0; k < E::P; ++k) l_npar[k * NE + le] = d.params[(size_t)k * ld + i];
            }
        }
        if (!DP && !G3) carry = Rng::philox(seed, d.idx0 + (uint32_t)i, RNG_ACT, blk0);
        // Reward, returns and records of the steps of batch bb, in three passes so that the arithmetic of the WS_R steps --
        // independent of each other -- sits in ONE basic block (instruction-level parallelism for a wave that otherwise
        // waits on its own dependent chains), the per-step bookkeeping with its rare branches in the second, the stores last.
        auto work_off = [&](auto full_tag, int bb, int nr) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tag)::value;
            float v[WS_R][M], a[WS_R][E::A], rw[WS_R], ob[WS_R][E::O], a_app[WS_R][E::A];
#pragma unroll
            for (int r = 0; r < WS_R; ++r) {
                if (!FULL && r >= nr) continue;
                Planes<M>::load(l_msg[bb & 1][r], NE, le, v[r]);
                if (DP) {
#pragma unroll
                    for (int j = 0; j < E::A; ++j) a[r][j] = v[r][AM0 + j];
                } else if (PC) {
                    float ua[AW];
                    Planes<AW>::load(l_act[ab(bb)][DP ? 0 : r], NE, le, ua);
#pragma unroll
                    for (int j = 0; j < E::A; ++j) a[r][j] = ua[E::A + j];
                } else {
                    Planes<E::A>::load(l_act[ab(bb)][DP ? 0 : r], NE, le, a[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < WS_R; ++r) {
                if (!FULL && r >= nr) continue;
                float an[E::A];
                {
                    const bool nrm = (T.flags & VS_FLAG_ACT_NORM) != 0;
                    float lb[E::A], ub[E::A];
                    E::act_bounds(c, lb, ub);
#pragma unroll
                    for (int j = 0; j < E::A; ++j) {
                        float m = lb[j] + (a[r][j] + 1.0f) * (ub[j] - lb[j]) * 0.5f;
                        an[j] = vsel(nrm, m, a[r][j]);
                    }
                }
                rw[r] = step_reward<E, float>(T, c, v[r], an);  // pre-step state, unclipped action (Q3)
                if (REC) E::observe_c(v[r], v[r] + TR0, ob[r]);  // the rest of observe(s_t)
                if (REC == 2) applied_action<E>(T, c, alo, ahi, a[r], a_app[r]);
            }
            bool dn[WS_R];
#pragma unroll
            for (int r = 0; r < WS_R; ++r) {
                if (!FULL && r >= nr) continue;
                const unsigned fl = __float_as_uint(v[r][M0 - 1]);
                const bool was_frozen = !AR ? (fl & WSF_FROZEN) != 0u : false;
                const bool done = (fl & WSF_DONE) != 0u, failed = (fl & WSF_FAILED) != 0u;
                const bool fin = done && !was_frozen && valid;
                dn[r] = done;
                if (!was_frozen) {
                    len += 1;  // curr_step of the running episode, counted on this side too
                    rew = rw[r];
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
                rw[r] = rew;
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
                            c_epi += 1u;
                        }
                    }
                }
            }
            if (REC) {
#pragma unroll
                for (int r = 0; r < WS_R; ++r) {
                    if (!FULL && r >= nr) continue;
                    const int t = bb * WS_R + r;
                    store_record<E, REC, GOBS ? 1 : 0>(d.traj_rec + (rec0 + (size_t)t) * Rec<E, REC>::F * ld, ld, i, ob[r], a[r], rw[r],
                                                       v[r], a_app[r], v[r] + HM0);
                    db.put(d, i, rec0 + (size_t)t, dn[r], t == k_steps - 1);
                }
            }
        };
        auto work = [&](int bb) __attribute__((always_inline)) {
            const int nr = min(WS_R, k_steps - bb * WS_R);
            if (nr == WS_R) work_off(std::true_type{}, bb, nr);
            else work_off(std::false_type{}, bb, nr);
        };
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (!G3) draw(0);
        ws_barrier();
#ifdef VS_WS_NOC  // diagnostic: the C wave only keeps the barriers
        for (int b = 0; b < nb; ++b) ws_barrier();
        if (false)
#endif
#ifdef VS_WS_STAMP
        unsigned long long acc0 = 0, acc1 = 0, acc2 = 0;
#endif
        for (int b = 0; b < nb; ++b) {
            VS_STAMP(st0);
            if (b >= 1) work(b - 1);  // reads l_act[(b - 1) & 1] before draw(b + 1) overwrites the same buffer
            VS_STAMP(st1);
            if (!G3) {
                if (b + 1 < nb) draw(b + 1);
                // (b == 0 fills every lane's first entry while this wave has no batch to work off yet: a lane of P that
                // resets before its entry is there draws for itself)
                if ((b & (WS_REFILL - 1)) == 0) refill();
            }
            VS_STAMP(st2);
            ws_barrier();
#ifdef VS_WS_STAMP
            VS_STAMP(st3);
            acc0 += st1 - st0, acc1 += st2 - st1, acc2 += st3 - st2;
#endif
        }
#ifdef VS_WS_STAMP
        if ((threadIdx.x & 63) == 0 && d.dbg) {
            unsigned long long* q = d.dbg + ((size_t)(i >> 6) * 3 + 1) * 4;
            q[0] = acc0, q[1] = acc1, q[2] = acc2, q[3] = (unsigned long long)nb;
        }
#endif
        work(nb - 1);
        const int i = VS_COLD(i_);
        d.ret[i] = ret;
        d.rew[i] = rew;
        if (E::FINAL != FINAL_NONE) d.yielded[i] = yielded;
        d.es_count[i] = es.count;
        d.es_retsum[i] = es.retsum;
        d.es_lensum[i] = es.lensum;
    } else {
        // ------------------------------------------------------------------------------------------- G wave (NR == 3)
        // The generator: everything that does not wait for the physics -- the policy's actions a batch ahead (with their
        // pre-processing when E::WS_PREP_C), the reset stock -- and, GOBS, the first plane of the records: it reads the
        // messages of batch b - 1 like the C wave, finishes the observation (E::observe_c) and stores its first four floats.
        if constexpr (G3) {
            const size_t rec0 = (size_t)d.traj_t0;
            if (STOCK) {
                c_epi = d.ep_idx[i];
                l_stag[le] = 0xFFFFFFFFu;
                l_epi[le] = c_epi;
                if (dr_stock) {
#pragma unroll
                    for (int k = 0; k < E::P; ++k) l_npar[k * NE + le] = d.params[(size_t)k * ld + i];
                }
            }
            carry = Rng::philox(seed, d.idx0 + (uint32_t)i, RNG_ACT, blk0);
            auto obs_off = [&](auto full_tag, int bb, int nr) __attribute__((always_inline)) {
                constexpr bool FULL = decltype(full_tag)::value;
                // step by step (load -> observe_c -> store): this wave is not the long one, and four steps' worth of messages
                // and observations held at once pushed the kernel past the 168 VGPRs three waves per SIMD allow
#pragma unroll
                for (int r = 0; r < WS_R; ++r) {
                    if (!FULL && r >= nr) continue;
                    float v[M], ob[E::O];
                    Planes<M>::load(l_msg[bb & 1][r], NE, le, v);
                    E::observe_c(v, v + TR0, ob);
                    const int t = bb * WS_R + r;
                    float* row = d.traj_rec + (rec0 + (size_t)t) * Rec<E, REC>::F * ld;
                    Planes<4>::template store<0, true>(row, ld, i, ob);
                }
            };
            auto obs_work = [&](int bb) __attribute__((always_inline)) {
                if (!GOBS) return;
                const int nr = min(WS_R, k_steps - bb * WS_R);
                if (nr == WS_R) obs_off(std::true_type{}, bb, nr);
                else obs_off(std::false_type{}, bb, nr);
            };
            __builtin_amdgcn_s_waitcnt(0x0F70);
            draw(0);
            ws_barrier();
#ifdef VS_WS_NOG  // diagnostic: the G wave only keeps the barriers (results are wrong)
            for (int b = 0; b < nb; ++b) ws_barrier();
            if (false)
#endif
#ifdef VS_WS_STAMP
            unsigned long long acc0 = 0, acc1 = 0, acc2 = 0;
#endif
            for (int b = 0; b < nb; ++b) {
                VS_STAMP(st0);
                if (b + 1 < nb) draw(b + 1);  // into l_act[(b + 1) % 3]: P reads b % 3, C (b - 1) % 3
                if ((b & (WS_REFILL - 1)) == 0) refill();
                VS_STAMP(st1);
                if (b >= 1) obs_work(b - 1);
                VS_STAMP(st2);
                ws_barrier();
#ifdef VS_WS_STAMP
                VS_STAMP(st3);
                acc0 += st1 - st0, acc1 += st2 - st1, acc2 += st3 - st2;
#endif
            }
#ifdef VS_WS_STAMP
            if ((threadIdx.x & 63) == 0 && d.dbg) {
                unsigned long long* q = d.dbg + ((size_t)(i >> 6) * 3 + 2) * 4;
                q[0] = acc0, q[1] = acc1, q[2] = acc2, q[3] = (unsigned long long)nb;
            }
#endif
            obs_work(nb - 1);
        }
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

#ifdef VS_TU_MIXED
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

// DRK: some member handle has a live randomizer or a parameter buffer (see k_rollout_ws): only then do the bodies carry the redraw
template <bool AR, int REC, bool DRK = false>
__global__ __launch_bounds__(BLOCK) void k_rollout_mixed(const Segs* __restrict__ segs, int k_steps, uint64_t seed) {
    int b = blockIdx.x, q = 0, first = 0;
    int n = segs->n;
    while (q < n - 1 && b >= segs->s[q].block_end) first = segs->s[q++].block_end;
    const Seg& sg = segs->s[q];
    MIXED_DISPATCH(sg.type, (rollout_body<E, false, AR, REC, false, DRK>(sg.T, sg.d, k_steps, seed, sg.reset_seed, sg.epoch0, b - first)));
}

template <bool AR, bool DRK = false>
__global__ __launch_bounds__(BLOCK) void k_step_mixed(const Segs* __restrict__ segs) {
    int b = blockIdx.x, q = 0, first = 0;
    int n = segs->n;
    while (q < n - 1 && b >= segs->s[q].block_end) first = segs->s[q++].block_end;
    const Seg& sg = segs->s[q];
    MIXED_DISPATCH(sg.type, (step_body<E, false, AR, false, 0, DRK>(sg.T, sg.d, sg.act, sg.env_stride, sg.dim_stride, sg.reset_seed, b - first)));
}
#endif  // VS_TU_MIXED

// ------------------------------------------------------------------------------------------ rollouts out of the records
// vs_pack_traj: the recorded steps of the first n lanes, time-major planes [t][plane][lane], as ROLLOUTS -- rollout j = steps
// 0 .. len[j] - 1 of lane j, the rollouts one after the other (what rollout() returns per env, rollout.py:305-325, and
// StepSequence.concat makes of many, step_sequence.py:777-825) -- in ONE row-major matrix rows[total + n][F], F the record width:
//   rows[start[j] + j + t] = the record of step t of rollout j  [obs | act | rew | state | act_app | hidden]   (t < len[j])
//   rows[start[j] + j + len[j]] = the entry behind the last step: the final observation / state / hidden state (from VS_OBS /
//                                 VS_STATE / VS_HIDDEN of the lane, frozen at its done), the per-step fields zero
// so that every field of a rollout is a strided view of its len[j] (+ 1) rows.
// A transpose through LDS: a workgroup owns 64 lanes and walks through a segment of PK_SEG steps in tiles of TT steps.  Load
// side: for a given step the 64 lanes read 64 x 16 contiguous bytes per record plane (only lanes whose rollout reaches that
// step).  Store side: a lane's TT steps are ONE contiguous run of TT x F floats in the destination (QQube, full records: 832 B)
// and consecutive threads write consecutive floats of it.  Round 2 wrote six separate per-field arrays instead -- runs of 64 B
// for the one-float fields (action, reward, applied action), 8-byte stores for the 6-float observation rows -- and stayed at
// 0.46 of HBM; with one matrix every wave-level store is 256 contiguous bytes.  HBM-bound: (F read + F written) x 4 B per step.
constexpr int PK_LANES = 64, PK_SEG = 256;
template <class E, int REC>
__global__ __launch_bounds__(BLOCK) void k_pack_traj(Dev d, int n, const long long* __restrict__ len,
                                                     const long long* __restrict__ start, float* __restrict__ rows) {
    constexpr int F = Rec<E, REC>::F;
    constexpr int B = E::O + E::A + 1;
#ifdef VS_PK_TT  // (experiments)
    constexpr int TT = VS_PK_TT;
#else
    constexpr int TT = F > 16 ? 8 : 16;   // steps per tile: 64 x TT x F floats of LDS (QQube, full records: 53 KB)
#endif
    constexpr int RUN = TT * F;           // floats of a lane's TT steps in the destination
    constexpr int CA = 32;                // floats of a 128-byte line: the carry area in front of a lane's LDS row (below)
    constexpr int RS = (CA + RUN) | 1;    // odd row stride: the load side writes a column across 64 rows without bank conflicts
    static_assert(BLOCK == 256 && PK_SEG % TT == 0 && TT % 4 == 0, "four waves per workgroup");
    __shared__ float tile[PK_LANES * RS];   // [lane][carry area | TT x F floats]: 61.7 KB for the QQube's full records
    const int lane0 = blockIdx.x * PK_LANES;
    const int tid = threadIdx.x, l = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t ld = d.ld;
    // every wave holds the 64 lanes' lengths and first destination rows (start[lane] + lane) in registers, one lane each; the
    // store side fetches them with v_readlane (the lane it serves is wave-uniform)
    const int i = lane0 + l;
    const int myL = i < n ? (int)len[i] : 0;
    const long long myRow = i < n ? start[i] + (long long)i : 0;
    const int rowLo = (int)(unsigned)myRow, rowHi = (int)(myRow >> 32);
    int l_max = myL;
#pragma unroll
    for (int o = 32; o; o >>= 1) l_max = max(l_max, __shfl_xor(l_max, o));
    const int seg0 = blockIdx.y * PK_SEG;
    const int seg1 = min(seg0 + PK_SEG, l_max);
    // the records of tile k + 1 are loaded into registers BEFORE tile k's store side runs, and the two barriers of a tile wait
    // for LDS traffic only (ws_barrier: lgkmcnt, not vmcnt): the global stores of a tile stay in flight behind it
    constexpr int RPW = TT / 4;  // steps of a tile per wave on the load side
    constexpr int LPW = PK_LANES / 4;  // lanes of a tile per wave on the store side
    // (non-temporal loads of the planes and stores of the rows were measured: 3.96 -> 3.28 TB/s full-length, 2.79 -> 2.46 ragged)
    float v[RPW][F];
    auto fetch = [&](int tb) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int t = tb + wave * RPW + r;
            if (t < myL) Planes<F>::load(d.traj_rec + (size_t)t * F * ld, ld, i, v[r]);
        }
    };
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (seg0 < seg1) fetch(seg0);
    for (int tb = seg0; tb < seg1; tb += TT) {
        // ---- load side: wave w took steps tb + w * (TT / 4) .. of the tile, a lane its own record
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int ts = wave * RPW + r;
            if (tb + ts < myL) {
#pragma unroll
                for (int f = 0; f < F; ++f) tile[l * RS + CA + ts * F + f] = v[r][f];
            }
        }
        ws_barrier();
        if (tb + TT < seg1) fetch(tb + TT);  // (in flight while this tile is stored)
        // ---- store side: wave w writes for lanes 16 w .. 16 w + 15, one lane at a time (which lane, its length and its row are
        // wave-uniform), 16 bytes per thread, WHOLE 128-BYTE LINES of the destination only: a lane's steps are one contiguous
        // stream there, but a tile's piece of it (TT x F floats) neither starts nor ends on a line (rows are F x 4 bytes, F odd
        // for most families), and a line written in two halves a tile apart costs the memory side two partial writes (measured:
        // rows padded to 64 bytes -- 23 % more bytes written -- took 3 % LESS time).  So the piece behind the last line boundary
        // (< 32 floats) is carried in LDS in front of the lane's row and goes out with the next tile; a lane's last tile in this
        // workgroup's segment flushes everything.
#pragma unroll 2
        for (int j = 0; j < LPW; ++j) {
            const int ln = wave * LPW + j;
            const int Lj = __builtin_amdgcn_readlane(myL, ln);
            const int nv = min(TT, Lj - tb) * F;
            if (nv <= 0) continue;
            const long long row = ((long long)__builtin_amdgcn_readlane(rowHi, ln) << 32) | (unsigned)__builtin_amdgcn_readlane(rowLo, ln);
            float* base = rows + (size_t)(row + tb) * F;        // where the tile's first float goes
            const uintptr_t bb = (uintptr_t)base;
            const bool flush = tb + TT >= Lj || tb + TT >= seg0 + PK_SEG;
            const int relP = tb > seg0 ? -(int)((bb & 127) >> 2) : 0;               // first float not written yet (<= 0: carried)
            const int relA = (int)((long long)(((bb + 4 * (long long)relP) & ~(uintptr_t)15) - bb) >> 2);   // ... down to 16 bytes
            const int relQ = flush ? nv : (int)((((bb + 4 * (uintptr_t)nv) & ~(uintptr_t)127) - bb) >> 2);  // end of what goes out now
            const float* src = tile + ln * RS + CA;             // src[r] = float r of the tile's piece, src[-c ..] the carry
            for (int q = relA + 4 * l; q < relQ; q += 256) {
                if (q >= relP && q + 4 <= relQ) {
                    f4 x = {src[q], src[q + 1], src[q + 2], src[q + 3]};
                    *(f4*)(base + q) = x;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (q + k >= relP && q + k < relQ) base[q + k] = src[q + k];
                }
            }
            const int c = nv - relQ;   // carried into the next tile (0 after a flush)
            if (l < c) tile[ln * RS + CA - c + l] = src[relQ + l];
        }
        ws_barrier();  // the tile has been read (LDS only: the global stores stay in flight)
    }
    // ---- the entry behind a rollout's last step: the frozen lane's final observation / state (by the segment that holds it)
    if (wave == 0 && myL > 0 && myL - 1 >= seg0 && myL - 1 < seg0 + PK_SEG) {
        float* fin = rows + (size_t)(myRow + myL) * F;
#pragma unroll
        for (int f = 0; f < F; ++f) fin[f] = 0.f;
#pragma unroll
        for (int j = 0; j < E::O; ++j) fin[j] = d.obs[j * ld + i];
        if constexpr (REC == 2) {
#pragma unroll
            for (int j = 0; j < E::S; ++j) fin[B + j] = d.state[j * ld + i];
#pragma unroll
            for (int j = 0; j < E::H; ++j) fin[B + E::S + E::A + j] = d.hidden[j * ld + i];
        }
    }
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

}  // namespace vs

// ====================================================================================================================
// host side shared by the translation units
// ====================================================================================================================
struct vs_env {
    int type = 0;
    int device = 0;
    vs::Task task{};
    vs::DrSpecs dr{};                 // host copy of the live randomizer
    vs::DrSpecs* d_specs = nullptr;   // device scratch for vs_sample_params
    float* d_pbuf = nullptr;      // DomainRandWrapperBuffer parameter sets
    float* d_ring = nullptr;      // ActDelayWrapper ring (Pipe::ring)
    vs::Fnn fnn{};                // vs_set_policy_fnn: the network vs_step_policy evaluates (fnn.w == nullptr: none)
    int rollout_variant = -1;     // vs_set_rollout_variant: -1 automatic, 0 k_rollout, 1 k_rollout_ws<256>, 2 k_rollout_ws<64>, 3 / 4 the three-role kernel in 64 / 256-env workgroups
    int policy_shape = -1;        // vs_set_policy_shape: -1 automatic, 0 / 1: k_rollout_fnn in 64- / 256-env workgroups, 2: 256-env + matrix cores
    int n_cu = 256;               // compute units of the device (256 on MI355X)
    bool auto_reset = false;
    uint64_t ar_seed = 0;
    bool uniform = true;
    hipStream_t own_stream = nullptr, stream = nullptr;
    vs::Dev d{};
    int record_mode = 1;  // vs_set_record_mode: layout of the VS_TRAJ_REC rows
    int traj_cap = 0;     // rows
    uint64_t epoch = 0;   // absolute step index of the action stream of vs_step_random
    std::string err;
    std::vector<void*> allocs;
    void* stage = nullptr;
    size_t stage_bytes = 0;
    void* stage_mask = nullptr;
    unsigned long long* d_counter = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // vs_timer_start / vs_timer_stop
};

namespace vs {

static inline dim3 grid_for(int ld) { return dim3((unsigned)((ld + BLOCK - 1) / BLOCK)); }

enum RolloutVariant { RV_PLAIN = 0, RV_WS256 = 1, RV_WS64 = 2, RV_WS64G = 3, RV_WS256G = 4 };

// per-family launchers: defined (explicitly instantiated) in vecsim_family.hip, one translation unit per family
template <class E>
struct Launch {
    static void step(vs_env* h, const float* act, long es, long ds, int rec = 0, int row = 0);
    static void rollout(vs_env* h, int k, uint64_t seed, uint64_t ep, int rec);
    static void rollout_fnn(vs_env* h, int k, int rec, uint64_t noise_seed);  // vs_step_policy
    static int variant(vs_env* h);  // RolloutVariant vs_step_random would launch for the handle's configuration
    static void jac(vs_env* h, const float* act, long es, long ds);
    static void set_params(vs_env* h, const float* src, long pitch, int bcast, const uint8_t* mask);
    static void sample_params(vs_env* h, uint64_t seed, const uint8_t* mask);
    static void reset(vs_env* h, const float* init, long pitch, int full, const uint8_t* mask, uint64_t seed);
    static void observe(vs_env* h);
    static void pack_traj(vs_env* h, int n, int t_steps, const long long* len, const long long* start, float* rows);  // vs_pack_traj
};

// mixed batches: defined in vecsim_mixed.hip
void launch_rollout_mixed(const Segs* dev_segs, int total_blocks, hipStream_t st, bool ar, int rec, int k_steps, uint64_t seed, bool drk);
void launch_step_mixed(const Segs* dev_segs, int total_blocks, hipStream_t st, bool ar, bool drk);

#ifdef VS_TU_FAMILY
// Which fused kernel for a batch (measured on MI355X, profiles/r02_table_variants.txt; 256 compute units):
//   * the wave-specialised kernel pays while k_rollout would leave a SIMD with a single wave, for the families whose step
//     splits into two comparable halves, and needs constants its reward wave reads not to change inside the launch;
//   * up to 64 envs per compute unit (16 384): 64-env workgroups -- every family (4 096 QQube envs: 43.7 against 47.9 us);
//     the ball balancer up to 128 per CU (E::WS_SMALL: its two waves then have a SIMD each; its ~400 VGPRs allow no second
//     wave of that shape per SIMD);
//   * up to 256 envs per compute unit (65 536): 256-env workgroups (one per CU, a P and a C wave on every SIMD) or
//     64-env ones, whichever the family runs faster (E::WS_SHAPE_FULL); with a live randomizer the 64-env shape: a
//     resetting lane's redraw stalls one pair of waves instead of four (QQube, 7 parameters: 120 against 169 us);
//   * up to 384 envs per compute unit (98 304): 64-env workgroups still beat k_rollout's one-and-a-half waves per SIMD
//     (QQube: 72 against 86 us per 100 recorded steps) where the kernel's registers allow a third wave per SIMD (E::WS_MID:
//     not the cartpole's 224);
//   * beyond: k_rollout has two or more waves per SIMD by itself;
//   * the families with E::WS_G3 (since round 3 every family but the cartpole's stabilisation task) run THREE waves per 64 envs up to
//     256 envs per compute unit (profiles/r02_table_three_roles.txt: QQube at 65 536 envs 44.8 -> 37.7 us, at 4 096 envs 32.7 -> 27.1 us
//     per 100 steps; profiles/r03_table_families.txt: cartpole, ball-on-beam, ball balancer + 1 .. 5 %), in the shape E::WS_G3_FULL at 256.
// VS_ROLLOUT_VARIANT=plain|ws|ws64|g64|g256 overrides for every handle (experiments); vs_set_rollout_variant pins per handle.
template <class E>
int Launch<E>::variant(vs_env* h) {
    if (E::FINAL == FINAL_STATE_TIME) return RV_PLAIN;
    if (h->d.pipe.act_on || h->d.pipe.obs_on) return RV_PLAIN;
    const bool live = h->dr.n > 0 || h->d.pbuf_n > 0;
    if (live && E::REWARD_SIDE_USES_CONSTS) return RV_PLAIN;
    // the three-role shapes exist for the families they pay for (E::WS_G3); elsewhere a pin falls back to the two-role shape
    auto have = [](int v) { return E::WS_G3 ? v : (v == RV_WS64G ? (int)RV_WS64 : v == RV_WS256G ? (int)RV_WS256 : v); };
    if (h->rollout_variant >= 0) return have(h->rollout_variant);
    static const char* force = getenv("VS_ROLLOUT_VARIANT");
    if (force && force[0] == 'p') return RV_PLAIN;
    if (force && force[0] == 'w') return force[1] && force[2] == '6' ? RV_WS64 : RV_WS256;
    if (force && force[0] == 'g') return have(force[1] == '6' ? RV_WS64G : RV_WS256G);  // g64 | g256
    const int64_t ld = h->d.ld, cu = h->n_cu;
    if (E::WS_G3) {
        // three waves per 64 envs: 64-env workgroups up to 128 envs per compute unit (and under a live randomizer, where a
        // redraw then stalls one trio of waves instead of four), 256-env workgroups (one per CU, a wave of each role on every
        // SIMD) up to 256; beyond that the two-role shape has its third wave per SIMD from the envs themselves
        if (ld <= 128 * cu) return RV_WS64G;
        if (ld <= 256 * cu) return (live || E::WS_G3_FULL == 64) ? RV_WS64G : RV_WS256G;
    }
    if (ld <= E::WS_SMALL * cu) return RV_WS64;
    if (!E::WS_PAYS) return RV_PLAIN;
    if (ld <= 256 * cu) return live ? RV_WS64 : (E::WS_SHAPE_FULL == 64 ? RV_WS64 : RV_WS256);
    if (ld <= 384 * cu && E::WS_MID) return (E::WS_G3 && ld > (int64_t)E::WS_MID_G3_FROM * cu) ? RV_WS64G : RV_WS64;
    return RV_PLAIN;
}

// from 4 M envs on (0.5 GB per step: twice the Infinity Cache); measured NT against default, QQube, lean: 1 M + 10 %, 2 M - 7 %,
// 4 M + 7 %, 16.7 M + 24 %
constexpr int64_t STEP_NT_MIN_ENVS = 4 << 20;
template <class E, int REC>
static void launch_step_rec(vs_env* h, const float* act, long es, long ds, int row) {
    dim3 g = grid_for(h->d.ld), b(BLOCK);
    bool uni = h->uniform && h->dr.n == 0 && h->d.pbuf_n == 0;
    const bool drk = h->d.dr_n > 0 || h->d.pbuf_n > 0;  // a reset inside the launch redraws domain parameters
    if constexpr (REC == 0) {
        // large batches: the non-temporal instantiation (see ld_nt); VS_STEP_NT=0|1 overrides (experiments)
        static const char* nt_env = getenv("VS_STEP_NT");
        const bool nt = nt_env ? nt_env[0] == '1' : (int64_t)h->d.ld >= STEP_NT_MIN_ENVS;
        if (nt && !(h->d.pipe.act_on || h->d.pipe.obs_on) && !drk) {
#define LN(U, AR) hipLaunchKernelGGL((k_step<E, U, AR, false, 0, false, true>), g, b, 0, h->stream, h->task, h->d, act, es, ds, h->ar_seed, row)
            if (h->auto_reset) { if (uni) LN(true, true); else LN(false, true); }
            else { if (uni) LN(true, false); else LN(false, false); }
#undef LN
            return;
        }
    }
#define LS(U, AR, PI, DK) hipLaunchKernelGGL((k_step<E, U, AR, PI, REC, DK>), g, b, 0, h->stream, h->task, h->d, act, es, ds, h->ar_seed, row)
    if (h->d.pipe.act_on || h->d.pipe.obs_on) {  // the wrapper pipeline: per-env-constant variant only
        if (h->auto_reset) { if (drk) LS(false, true, true, true); else LS(false, true, true, false); } else LS(false, false, true, false);
    } else if (h->auto_reset) { if (uni) LS(true, true, false, false); else if (drk) LS(false, true, false, true); else LS(false, true, false, false); }
    else { if (uni) LS(true, false, false, false); else LS(false, false, false, false); }
#undef LS
}

template <class E>
void Launch<E>::step(vs_env* h, const float* act, long es, long ds, int rec, int row) {
    if (rec == 0) launch_step_rec<E, 0>(h, act, es, ds, 0);
    else if (rec == 1) launch_step_rec<E, 1>(h, act, es, ds, row);
    else launch_step_rec<E, 2>(h, act, es, ds, row);
}

template <class E, bool U, bool AR, int NE, int NR = 2>
static void launch_ws(vs_env* h, int k, uint64_t seed, uint64_t ep, int rec) {
    if constexpr (E::FINAL != FINAL_STATE_TIME) {
        dim3 g((unsigned)(h->d.ld / NE)), b(NR * NE);
        // dynamic LDS: the live randomizer's part of the reset stock (see the kernel), only when one is set
        const unsigned dyn = (!U && AR && h->d.dr_n > 0 && h->d.pbuf_n == 0) ? (unsigned)((E::P + E::K) * NE * sizeof(float)) : 0u;
        // the instantiation with the redraw compiled in only for a handle that has a randomizer or a parameter buffer
        const int drk = (!U && AR) ? (h->d.pbuf_n > 0 ? 2 : (h->d.dr_n > 0 ? 1 : 0)) : 0;
        // R = 4 steps per exchange (measured on the headline config: R = 1 / 2 / 4 -> 68.7 / 64.8 / 62.0 us per 100 steps)
#define LWK(REC, DRKV)                                                                                                     \
    {                                                                                                                      \
        auto kern = k_rollout_ws<E, U, AR, REC, 4, NE, (NR == 2 && E::WS_DRAW_P && REC != 0), NR, DRKV>;                    \
        static unsigned char attr_set[64] = {}; /* once per kernel and device */                                           \
        if (dyn && !attr_set[h->device & 63]) {                                                                            \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);            \
            attr_set[h->device & 63] = 1;                                                                                  \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, g, b, dyn, h->stream, h->task, h->d, k, seed, h->ar_seed, ep);                            \
    }
#define LW(REC)                                                                                                            \
    {                                                                                                                      \
        if constexpr (!U && AR) { if (drk == 2) LWK(REC, 2) else if (drk == 1) LWK(REC, 1) else LWK(REC, 0) }               \
        else LWK(REC, 0)                                                                                                   \
    }
        if (rec == 0) LW(0) else if (rec == 1) LW(1) else LW(2)
#undef LW
#undef LWK
    }
}

template <class E, bool U, bool AR, bool PI>
static void launch_plain(vs_env* h, int k, uint64_t seed, uint64_t ep, int rec) {
    dim3 g = grid_for(h->d.ld), b(BLOCK);
    const bool drk = h->d.dr_n > 0 || h->d.pbuf_n > 0;  // a reset inside the launch redraws domain parameters
#define LRK(REC, DK) hipLaunchKernelGGL((k_rollout<E, U, AR, REC, PI, DK>), g, b, 0, h->stream, h->task, h->d, k, seed, h->ar_seed, ep)
#define LR(REC) { if constexpr (!U && AR) { if (drk) LRK(REC, true); else LRK(REC, false); } else LRK(REC, false); }
    if (rec == 0) LR(0) else if (rec == 1) LR(1) else LR(2)
#undef LR
#undef LRK
}

template <class E>
void Launch<E>::rollout(vs_env* h, int k, uint64_t seed, uint64_t ep, int rec) {
    const bool uni = h->uniform && h->dr.n == 0;
    const bool ar = h->auto_reset;
    const int var = variant(h);
    if (var != RV_PLAIN) {
#define WS(NE, NR)                                                                                     \
    if (uni) { if (ar) launch_ws<E, true, true, NE, NR>(h, k, seed, ep, rec); else launch_ws<E, true, false, NE, NR>(h, k, seed, ep, rec); } \
    else { if (ar) launch_ws<E, false, true, NE, NR>(h, k, seed, ep, rec); else launch_ws<E, false, false, NE, NR>(h, k, seed, ep, rec); }
        if constexpr (E::WS_G3) {
            if (var == RV_WS64G) { WS(64, 3) return; }
            if (var == RV_WS256G) { WS(256, 3) return; }
        }
        if (var == RV_WS64 || var == RV_WS64G) { WS(64, 2) } else { WS(256, 2) }
#undef WS
        return;
    }
    if (h->d.pipe.act_on || h->d.pipe.obs_on) {
        if (ar) launch_plain<E, false, true, true>(h, k, seed, ep, rec); else launch_plain<E, false, false, true>(h, k, seed, ep, rec);
    } else if (uni) {
        if (ar) launch_plain<E, true, true, false>(h, k, seed, ep, rec); else launch_plain<E, true, false, false>(h, k, seed, ep, rec);
    } else {
        if (ar) launch_plain<E, false, true, false>(h, k, seed, ep, rec); else launch_plain<E, false, false, false>(h, k, seed, ep, rec);
    }
}

template <class E>
void Launch<E>::rollout_fnn(vs_env* h, int k, int rec, uint64_t noise_seed) {
#define LF(AR, REC, NH, NE, MF) hipLaunchKernelGGL((k_rollout_fnn<E, AR, REC, NH, NE, MF>), dim3((unsigned)(h->d.ld / NE)), dim3(64 * fnn_waves(NH)), 0, h->stream, h->task, h->d, h->fnn, k, h->ar_seed, noise_seed)
#define LFR(AR, NH, NE, MF) { if (rec == 0) LF(AR, 0, NH, NE, MF); else if (rec == 1) LF(AR, 1, NH, NE, MF); else LF(AR, 2, NH, NE, MF); }
#define LFA(NH, NE, MF) { if (h->auto_reset) LFR(true, NH, NE, MF) else LFR(false, NH, NE, MF) }
    // 256-env workgroups beyond 128 envs per compute unit, with the hidden layers on the matrix cores (one and two hidden layers;
    // measured: 65 536 QQube envs, 64 x 64 tanh).  VS_FNN_SHAPE=64|256|mfma pins the shape (experiments, tests).
    static const char* force = getenv("VS_FNN_SHAPE");
    // 0: 64-env workgroups (the network of 64 envs on 8 waves: 4.4 us per step while every workgroup has a compute unit),
    // 1: 256-env workgroups on the vector ALU, 2: 256-env workgroups on the matrix cores (7.8 us per step of up to 256 envs
    // per compute unit: the choice beyond 64 envs per compute unit; profiles/r02_table_policy_kernel.txt)
    // -- and, whatever the batch, for two hidden layers of at most 32 units each: one tile per layer, 3.6 against 4.5 us)
    const bool narrow2 = h->fnn.n_hidden == 2 && h->fnn.hidden[0] <= 32 && h->fnn.hidden[1] <= 32;
    int shape = ((int64_t)h->d.ld > 64 * (int64_t)h->n_cu || narrow2) ? 2 : 0;
    if (force) shape = force[0] == 'm' ? 2 : force[0] == '2' ? 1 : 0;
    if (h->policy_shape >= 0) shape = h->policy_shape;
    if (h->fnn.n_hidden > 2) shape = 0;
    switch (h->fnn.n_hidden) {
        case 1: if (shape == 2) LFA(1, 256, true) else if (shape == 1) LFA(1, 256, false) else LFA(1, 64, false) break;
        case 2: if (shape == 2) LFA(2, 256, true) else if (shape == 1) LFA(2, 256, false) else LFA(2, 64, false) break;
        case 3: LFA(3, 64, false) break;
        default: LFA(4, 64, false) break;
    }
#undef LFA
#undef LFR
#undef LF
}

template <class E>
void Launch<E>::jac(vs_env* h, const float* act, long es, long ds) {
    bool uni = h->uniform && h->dr.n == 0 && h->d.pbuf_n == 0;
    dim3 g = grid_for(h->d.ld), b(BLOCK);
    if (uni) hipLaunchKernelGGL((k_step_jac<E, true>), g, b, 0, h->stream, h->task, h->d, act, es, ds);
    else hipLaunchKernelGGL((k_step_jac<E, false>), g, b, 0, h->stream, h->task, h->d, act, es, ds);
}

template <class E>
void Launch<E>::set_params(vs_env* h, const float* src, long pitch, int bcast, const uint8_t* mask) {
    hipLaunchKernelGGL(k_set_params<E>, grid_for(h->d.ld), dim3(BLOCK), 0, h->stream, h->task, h->d, src, pitch, bcast, mask);
}

template <class E>
void Launch<E>::sample_params(vs_env* h, uint64_t seed, const uint8_t* mask) {
    hipLaunchKernelGGL(k_sample_params<E>, grid_for(h->d.ld), dim3(BLOCK), 0, h->stream, h->task, h->d,
                       (const DrSpecs*)h->d_specs, seed, mask);
}

template <class E>
void Launch<E>::reset(vs_env* h, const float* init, long pitch, int full, const uint8_t* mask, uint64_t seed) {
    hipLaunchKernelGGL(k_reset<E>, grid_for(h->d.ld), dim3(BLOCK), 0, h->stream, h->task, h->d, init, pitch, full, mask, seed);
}

template <class E>
void Launch<E>::observe(vs_env* h) {
    hipLaunchKernelGGL(k_observe<E>, grid_for(h->d.ld), dim3(BLOCK), 0, h->stream, h->d);
}

template <class E>
void Launch<E>::pack_traj(vs_env* h, int n, int t_steps, const long long* len, const long long* start, float* rows) {
    // 64 lanes x segments of PK_SEG steps: 65 536 lanes x 4 000 steps = 16 384 workgroups, 4 096 lanes still 1 024
    dim3 g((unsigned)((n + PK_LANES - 1) / PK_LANES), (unsigned)((t_steps + PK_SEG - 1) / PK_SEG));
    if (h->record_mode == 2) hipLaunchKernelGGL((k_pack_traj<E, 2>), g, dim3(BLOCK), 0, h->stream, h->d, n, len, start, rows);
    else hipLaunchKernelGGL((k_pack_traj<E, 1>), g, dim3(BLOCK), 0, h->stream, h->d, n, len, start, rows);
}
#endif  // VS_TU_FAMILY

}  // namespace vs
