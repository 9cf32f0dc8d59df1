// vecsim_envs.h -- per-environment device code (fp32, one environment per wavefront lane) for libvecsim.
//
// Each struct restates one Pyrado SimPyEnv family for CDNA4; reference file:line is given at every function.
// `P/` = Pyrado/pyrado/ of swami1995/SimuRLacra.  Quirk numbers (Q1..Q15) refer to SURVEY.md section 0.
//
// Layout contract shared with vecsim.hip and the Python shim:
//   params[P]  raw domain parameters in get_nominal_domain_param() order
//   consts[K]  derived constants; the first KS of them are what one step() reads, the rest are reset-only
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "vecsim_dual.h"

namespace vs {

constexpr int MAXS = 8, MAXA = 2, MAXO = 8, MAXH = 2, MAXP = 20, MAXK = 20;
constexpr double PI_D = 3.14159265358979323846;
constexpr float PI_F = (float)PI_D;
// bounds are the fp64 expressions of the reference rounded once to fp32 (not re-derived in fp32 arithmetic)
constexpr float PI_4_F = (float)(PI_D / 4.0), PI4_F = (float)(4.0 * PI_D), PI5_F = (float)(5.0 * PI_D),
                PI20_F = (float)(20.0 * PI_D), QQ_TH_MAX_F = (float)(115.0 / 180.0 * PI_D);
constexpr float TWO_PI_F = 6.28318530717958647692f;
// 2*pi = TWO_PI_HI + TWO_PI_LO with TWO_PI_HI == (float)2pi: two-term reduction keeps fmod(err, 2pi) at fp64 quality
constexpr float TWO_PI_HI = 6.28318548202514648437500f;
constexpr float TWO_PI_LO = -1.74845553146951715461910e-07f;

enum RewKind { REW_QUADR = 0, REW_EXP = 1, REW_SCALED_EXP = 2 };

// Task + ctor configuration: a by-value kernel argument (wave-uniform, lands in SGPRs / scalar loads)
struct Task {
    float des[MAXS];  // DesStateTask.state_des            P/tasks/desired_state.py:57
    float qd[MAXS];   // diag(Q)                            P/tasks/reward_functions.py:202-221
    float rd[MAXA];   // diag(R)
    float dt;
    int max_steps;  // INT_MAX == pyrado.inf
    int flags;      // VS_FLAG_*
    int wild_init;  // qcp only
    float init_fixed[MAXS];  // SingularStateSpace (pend): the fixed initial state
};

// ------------------------------------------------------------------------------------------------- small helpers
__device__ __forceinline__ float sqr(float x) { return x * x; }

// np.fmod(e, 2*pi) for |e| < ~2^20 (P/tasks/desired_state.py:149): truncated quotient, exact remainder via FMA,
// 2pi split in hi+lo so the fp32 result tracks the fp64 one to ~1 ulp of e
__device__ __forceinline__ float fmod_2pi(float e) {
    float q = truncf(e * (1.0f / TWO_PI_F));
    float r = fmaf(-q, TWO_PI_HI, e);
    r = fmaf(-q, TWO_PI_LO, r);
    // q may be off by one when e/2pi rounds across an integer: bring r back to (-2pi, 2pi) with the sign of e
    // (branch-free: work on |r'| = r * sign(e), which must end up in [0, 2pi))
    float sg = copysignf(1.0f, e);
    float m = r * sg;
    m = m < 0.f ? m + TWO_PI_F : m;
    m = m >= TWO_PI_F ? m - TWO_PI_F : m;
    return m * sg;
}

// the two sequential +-pi folds of RadiallySymmDesStateTask.step_rew (P/tasks/desired_state.py:152-153, Q4)
// e > pi  <=>  2pi - e < e  and  e < -pi  <=>  -2pi - e > e, so each fold is a min / max with the reflected value
// (identical results for every non-NaN e, ties included; a NaN state is reported through the error flag).
template <class R>
__device__ __forceinline__ R fold_pi(R e) {
    e = vmin(e, TWO_PI_F - e);
    e = vmax(e, -TWO_PI_F - e);
    return e;
}
template <int N>
__device__ __forceinline__ Dual<N> fmod_2pi(const Dual<N>& e) {  // slope 1 everywhere it is differentiable
    Dual<N> r = e;
    r.v = fmod_2pi(e.v);
    return r;
}

// exp(x) for x <= 0 on the reward path: v_exp_f32 on x*log2(e).  Rewards below fp32's normal range flush to 0 either way
// (cost > 87); above that the rounding of x*log2(e) costs <= 5e-6 relative -- inside the 2e-4 reward tolerance, and
// 2 instructions instead of the library's ~14.
__device__ __forceinline__ float exp_neg_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
template <int N>
__device__ __forceinline__ Dual<N> exp_neg_fast(const Dual<N>& x) {
    Dual<N> r;
    r.v = exp_neg_fast(x.v);
#pragma unroll
    for (int k = 0; k < N; ++k) r.d[k] = r.v * x.d[k];
    return r;
}

// sin and cos of a BOUNDED angle (every angle on the hot path is a state inside / next to its box, |x| < ~1e3).
// Rounds 1-3 (kept behind -DVS_POLY_SINCOS): 3-term Cody-Waite reduction by pi/2 with FMAs (exact products), then the Cephes
// minimax polynomials on [-pi/4, pi/4] -- 28 VALU instructions for both values against ~150 for the general-range library
// sincosf (Payne-Hanek branch included), max error against fp64 over |x| <= 100: 9e-8 abs.  Shipped since the end of round 3:
// the hardware's transcendental unit behind an exact reduction (below), 3.5e-7 abs (tests/test_gpu_parity.py).
// NaN / inf inputs give NaN in both forms (the error flag relies on that).
__device__ __forceinline__ void sincos_fast(float x, float* sn, float* cs) {
#ifndef VS_POLY_SINCOS
    // The hardware's v_sin_f32 / v_cos_f32 (argument in revolutions) behind an EXACT two-term reduction to [-pi, pi]: 5 vector + 2
    // transcendental instructions (~52 issue cycles) where the Cody-Waite reduction + two polynomials + quadrant selects below are 28
    // (~112).  Maximum absolute error 3.8e-7 over |x| <= 70 rad against the polynomials' 9e-8 (scratch/ubench/hw_sincos_err.hip; the
    // multiply by 1 / 2 pi alone, without the reduction, would add |x| x 6e-8) -- three ulp of a value near 1, a thirtieth of the
    // 1e-5 relative the state trajectories are held to; every golden-vector and oracle parity test holds at its old tolerance.
    // What it bought (end of round 3, same box): headline 1.67e11 -> 1.84e11, BASELINE config 4 (four sincos per step) 6.6e10 ->
    // 9.1e10, config 3 1.04e11 -> 1.09e11, config 2 1.69e10 -> 1.79e10.  -DVS_POLY_SINCOS keeps the polynomial form (diagnostics).
    {
        const float TWOPI_HI = 6.28318548202514648f, TWOPI_LO = -1.74845553146951715e-07f, INV_2PI = 0.159154943091895336f;
        float q = rintf(x * INV_2PI);
        float r = fmaf(-q, TWOPI_HI, x);
        r = fmaf(-q, TWOPI_LO, r);
        float rev = r * INV_2PI;
        *sn = __builtin_amdgcn_sinf(rev);
        *cs = __builtin_amdgcn_cosf(rev);
        return;
    }
#endif
    const float PIO2_HI = 1.57079637050628662109375f;       // (float)(pi/2)
    const float PIO2_MID = -4.37113882867379127e-08f;       // (float)(pi/2 - HI)
    const float PIO2_LO = -1.71512451008199912e-15f;        // (float)(pi/2 - HI - MID)
    float q = rintf(x * 0.636619772367581343f);             // x * 2/pi
    float r = fmaf(-q, PIO2_HI, x);
    r = fmaf(-q, PIO2_MID, r);
    r = fmaf(-q, PIO2_LO, r);
    int n = (int)q;
    float r2 = r * r;
    float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
    float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), r2 * r2,
                    fmaf(-0.5f, r2, 1.0f));
    float s0 = (n & 1) ? pc : ps;
    float c0 = (n & 1) ? ps : pc;
    *sn = (n & 2) ? -s0 : s0;
    *cs = ((n + 1) & 2) ? -c0 : c0;
}

// 1/x: v_rcp_f32 (1 ulp) + one Newton step; 3 instructions instead of the ~10 of an IEEE-rounded division.
// Only for well-scaled positive denominators on the hot path (determinants, inertias); <= 1 ulp from the exact value.
__device__ __forceinline__ float rcp_fast(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}

// 1/x by the hardware approximation alone (1 ulp); Dual: through rcp_fast
__device__ __forceinline__ float rcp_det(float x) { return __builtin_amdgcn_rcpf(x); }
template <int N>
__device__ __forceinline__ Dual<N> rcp_det(const Dual<N>& x);  // (defined behind rcp_fast<Dual>)

// sin and cos of a0 + dl from (sin a0, cos a0) by the addition theorems, for a SMALL increment dl (|dl| <~ 0.5: the stage
// angles of an RK step are the step's angle plus dt/2 or dt times a rate).  Truncation error of the two series below
// |dl|^7 / 7! (sine: 1.6e-8 at the 0.26 rad the cartpole's dt <= 4 ms branch allows) and |dl|^8 / 8! (cosine); 11 instructions
// against the 28 of a fresh range reduction + polynomials.
template <class R>
__device__ __forceinline__ void sincos_rot(const R& s0, const R& c0, const R& dl, R* sn, R* cs) {
    R d2 = dl * dl;
    // (sine to dl^5: the dl^7 term is below 7e-8 of sin(dl) for |dl| <= 0.26 -- an ulp -- and was one more instruction per stage)
    R sd = dl + dl * d2 * (-1.6666667163e-1f + d2 * 8.3333337680e-3f);
    R cd = 1.0f + d2 * (-0.5f + d2 * (4.1666667908e-2f + d2 * -1.3888889225e-3f));
    *sn = s0 * cd + c0 * sd;
    *cs = c0 * cd - s0 * sd;
}

template <int N>
__device__ __forceinline__ void sincos_fast(const Dual<N>& x, Dual<N>* sn, Dual<N>* cs) {
    float sv, cv;
    sincos_fast(x.v, &sv, &cv);
    sn->v = sv;
    cs->v = cv;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        sn->d[k] = cv * x.d[k];
        cs->d[k] = -sv * x.d[k];
    }
}
template <int N>
__device__ __forceinline__ Dual<N> rcp_fast(const Dual<N>& x) {
    Dual<N> r;
    r.v = rcp_fast(x.v);
#pragma unroll
    for (int k = 0; k < N; ++k) r.d[k] = -r.v * r.v * x.d[k];
    return r;
}
template <int N>
__device__ __forceinline__ Dual<N> rcp_det(const Dual<N>& x) { return rcp_fast(x); }

__device__ __forceinline__ float sgnf(float x) { return (float)(x > 0.f) - (float)(x < 0.f); }  // np.sign

// ------------------------------------------------------------------------------------------------- Philox4x32-10
struct Rng {
    uint2 key;
    uint4 ctr;
    uint4 out;
    int used;
    __device__ __forceinline__ Rng(uint64_t seed, uint32_t env, uint32_t purpose, uint64_t t) {
        key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
        ctr = make_uint4(env, purpose, (uint32_t)t, (uint32_t)(t >> 32));
        used = 4;
    }
    __device__ __forceinline__ static uint4 rounds(uint4 c, uint2 k) {
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            // one 32x32->64 multiply per word pair (v_mad_u64_u32): integer multiplies are quarter-rate on CDNA4
            uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
            uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
            c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
            k.x += 0x9E3779B9u;
            k.y += 0xBB67AE85u;
        }
        return c;
    }
    // one block of four 32-bit words for (seed, env, purpose, t)
    __device__ __forceinline__ static uint4 philox(uint64_t seed, uint32_t env, uint32_t purpose, uint64_t t) {
        return rounds(make_uint4(env, purpose, (uint32_t)t, (uint32_t)(t >> 32)),
                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    }
    __device__ __forceinline__ static float to_u01(uint32_t bits) { return (float)(bits >> 8) * (1.0f / 16777216.0f); }
    __device__ __forceinline__ void block() {
        out = rounds(ctr, key);
        ctr.y += 0x10000u;  // next block of the same (env, purpose, t) stream; purposes stay below 2^16
        used = 0;
    }
    __device__ __forceinline__ uint32_t next() {
        if (used == 4) block();
        uint32_t v = used == 0 ? out.x : used == 1 ? out.y : used == 2 ? out.z : out.w;
        ++used;
        return v;
    }
    __device__ __forceinline__ float u01() { return to_u01(next()); }  // [0, 1)
    __device__ __forceinline__ float uniform(float lo, float hi) { return lo + (hi - lo) * u01(); }
    // Box-Muller on the hardware transcendentals: v_log_f32 (log2), v_sqrt_f32 and v_cos_f32 / v_sin_f32, whose argument is
    // in revolutions -- cos(2 pi u) is ONE instruction.  The library logf / cosf / sqrtf are ~350 instructions for the same
    // draw, executed by a whole wave for the few lanes that reset (live domain randomisation was 2.4x slower with them).
    // Absolute error of a draw ~1e-6 (3e-4 for |z| < 1e-3, where log(u1) vanishes): statistics, not bits, define these draws.
    __device__ __forceinline__ static void box_muller(uint32_t b0, uint32_t b1, float& z0, float& z1) {
        float u1 = 1.0f - to_u01(b0);  // (0, 1]
        float u2 = to_u01(b1);         // [0, 1) revolutions
        float r = __builtin_amdgcn_sqrtf(-1.3862944f * __builtin_amdgcn_logf(u1));  // sqrt(-2 ln u1), ln = ln2 * log2
        z0 = r * __builtin_amdgcn_cosf(u2);
        z1 = r * __builtin_amdgcn_sinf(u2);
    }
    __device__ __forceinline__ float normal() {  // one value per two uniforms
        uint32_t b0 = next(), b1 = next();
        float z0, z1;
        box_muller(b0, b1, z0, z1);
        return z0;
    }
};
enum RngPurpose { RNG_ACT = 1, RNG_INIT = 2, RNG_PARAM = 3, RNG_ACT_NOISE = 4, RNG_OBS_NOISE = 5, RNG_POLICY_NOISE = 6 };

// defaults shared by the env structs (static members are inherited)
enum FinalKind { FINAL_NONE = 0, FINAL_CONST_MALUS = 1, FINAL_STATE_TIME = 2 };
template <int A_>
struct EnvDefaults {
    static constexpr bool SYMMETRIC_BOX = true;  // state box lo == -hi
    static constexpr int FINAL = FINAL_NONE;
    // the wave-specialised rollout kernel (vecsim.hip, k_rollout_ws) pays for this family at <= 1 wave per SIMD
    static constexpr bool WS_PAYS = true;
    // ... and its reward / record wave reads per-env constants (action bounds, c_max): then the constants must not change
    // inside a launch, i.e. no live domain randomisation.  False where bounds are fixed numbers and the reward is unscaled.
    static constexpr bool REWARD_SIDE_USES_CONSTS = true;
    // ... and the workgroup shape (envs per workgroup) it runs fastest in at one workgroup's worth of envs per compute unit
    // (65 536 envs on MI355X; measured per family, profiles/r02_table_variants.txt): the light steps prefer 64
    static constexpr int WS_SHAPE_FULL = 64;
    // ... and whether 64-env workgroups still pay between 256 and 384 envs per compute unit (needs <= 168 VGPRs)
    static constexpr bool WS_MID = true;
    // ... in which shape there: three roles (64-env workgroups: six of them fit a compute unit's LDS) or two.  Measured at the end
    // of round 3 at 81 920 / 98 304 envs (scratch/r3/mid_n.sh): oscillator, ball-on-beam (both forms) and pendulum + 3 .. 14 % with three
    // roles from 256 envs per compute unit on, the QQube (both tasks) from 320 on (+ 3 .. 7 %: mid_qq.sh); the value is the number of
    // envs per compute unit from which the three-role shape is taken
    static constexpr int WS_MID_G3_FROM = 1 << 30;
    // ... and up to how many envs per compute unit the 64-env workgroups are used whatever WS_SHAPE_FULL says
    static constexpr int WS_SMALL = 64;
    // ... and whether its reward wave, which draws the policy's actions, also pre-processes them for the physics wave, for a
    // family whose physics wave is by far the longer one: 1 = ActNorm -> clip (needs action bounds that are plain numbers:
    // the reward wave's constants may be stale under live randomisation), 2 = ActNorm -> clip -> dead zone (the family
    // then provides dead_zone() and dynamics_core(), and never runs this kernel under live randomisation)
    static constexpr int WS_PREP_C = 0;
    // ... and whether, in the three-role kernel's 64-env workgroups with NO redraw of domain parameters compiled in (DRK = 0:
    // the constants cannot change inside a launch), the generator wave pre-processes the actions it draws (same levels): at
    // small batches every wave has a SIMD to itself, a step lasts as long as the physics wave's dependent chain, and the
    // generator wave idles a third of its time
    static constexpr int WS_PREP_G64 = 0;
    // ... and which of its two waves draws the policy's actions when steps are recorded: the physics wave where the other one
    // is the longer (it finishes observe() and stores the records), see k_rollout_ws
    static constexpr bool WS_DRAW_P = false;
    // ... and whether a THIRD wave per 64 envs pays (k_rollout_ws with NR = 3: a generator wave draws the actions a batch
    // ahead, keeps the reset stock and stores the first record plane): the families whose physics and reward waves are
    // comparable once the draw is off them (measured at 1 024 .. 65 536 envs, profiles/r02_table_three_roles.txt: QQube -16 %,
    // oscillator / pendulum -10 .. -20 %; ball-on-beam, cartpole and ball balancer, whose physics wave is the long one by
    // itself, gained nothing in round 2 -- and 1 .. 5 % at the end of round 3, with 57 .. 141 instead of 119 .. 436 registers)
    static constexpr bool WS_G3 = false;
    // ... and the workgroup shape of the three-role kernel at one workgroup's worth of envs per compute unit (65 536 envs)
    static constexpr int WS_G3_FULL = 256;
    // ... and how many waves per SIMD its two-role kernel must leave room for (the register budget the compiler gets: 512 / n)
    static constexpr int WS_MIN_WAVES = 1;
    // ... and whether the two waves of its 64-env workgroups want a SIMD each (at most one wave per SIMD: see k_rollout_ws)
    static constexpr bool WS_ALONE = false;
    // Env.limit_act -> BoxSpace.project_to (P/spaces/box.py:180-184); np.clip propagates NaN (fminf/fmaxf would drop it)
    template <class R>
    __device__ static void limit_act(const float*, const float* lo, const float* hi, const R* a_raw, R* a) {
#pragma unroll
        for (int j = 0; j < A_; ++j) {
            a[j] = vmin(vmax(a_raw[j], lo[j]), hi[j]);
            if (visnan(a_raw[j])) a[j] = a_raw[j];
        }
    }
    // DummyPolicy: act_space.sample_uniform() (P/policies/feed_forward/dummy.py:77-84)
    __device__ static float sample_action(const float*, float lo, float hi, float u01, int) { return lo + (hi - lo) * u01; }
    // Trig shared between observe() and the dynamics.  A family whose observation holds sin / cos of an angle the dynamics
    // need too declares TRIG = 2 and TRIG_AT = the index of that (sin, cos) pair inside the observation:
    //   observe_p(s, tr)      the pair (what a step from the same state reuses: dynamics(..., tr))
    //   observe_c(s, tr, o)   the observation from the state and the pair (the rest of observe(): more trig for QQube)
    // observe(s, o) == observe_p + observe_c, statement for statement.  The wave-specialised rollout kernel keeps
    // observe_p on its physics wave and hands (s, tr) over; the reward / record wave finishes the observation.
    static constexpr int TRIG = 0, TRIG_AT = 0;
    template <class R>
    __device__ static void observe_p(const R*, R*) {}
};
// observe_c of the families whose observation is a copy of the state (TRIG == 0)
#define VS_OBSERVE_C_IS_OBSERVE \
    template <class R>          \
    __device__ static void observe_c(const R* s, const R*, R* o) { observe(s, o); }

// =================================================================================================== OMO
// OneMassOscillatorSim, P/environments/pysim/one_mass_oscillator.py:49-121
struct Omo : EnvDefaults<1> {
    static constexpr int WS_MID_G3_FROM = 0;
    static constexpr int WS_PREP_G64 = 1;  // 4 096 envs + 3.7 %, 32 768 + 5 %
    static constexpr int S = 2, A = 1, O = 2, H = 0, I = 2, P = 3, K = 4, KS = 4;
    static constexpr int REW = REW_QUADR, RADIAL = -1, CMAX = -1;
    static constexpr int FINAL = FINAL_CONST_MALUS;  // FinalRewTask(factor 1e3, always_negative), :75-79
    static constexpr bool WS_DRAW_P = true;  // a two-instruction physics step: the reward / record wave is the long one
    static constexpr bool WS_G3 = true;
    // (WS_PAYS: with its batch loops unrolled the split pays even for this small step: +8 % with records, +5 % without)
    enum { C_A10, C_A11, C_B1, C_AMAX };
    __device__ static void calc_consts(const Task&, const float* p, float* c) {  // _calc_constants :88-103
        float m = p[0], k = p[1], d = p[2];
        float omega = sqrtf(k / m);
        float zeta = d / (2.0f * sqrtf(m * k));
        c[C_A10] = -(omega * omega);        // A[1,0], _step_dynamics :109
        c[C_A11] = -2.0f * zeta * omega;    // A[1,1]
        c[C_B1] = 1.0f / m;                 // B[1]
        c[C_AMAX] = 1.0f * k;               // max_act = max_state[0] * k, _create_spaces :61
    }
    __device__ static void state_bounds(const float*, float* lo, float* hi) {  // :58,64
        hi[0] = 1.0f; hi[1] = 10.0f; lo[0] = -1.0f; lo[1] = -10.0f;
    }
    __device__ static void act_bounds(const float* c, float* lo, float* hi) { hi[0] = c[C_AMAX]; lo[0] = -c[C_AMAX]; }
    template <class R>
    __device__ static void dynamics(const Task& T, const float* c, R* s, R*, const R* a, const R*) {  // :105-114
        R sd0 = s[1];
        R sd1 = c[C_A10] * s[0] + c[C_A11] * s[1] + c[C_B1] * a[0];
        s[0] = s[0] + sd0 * T.dt;  // forward Euler
        s[1] = s[1] + sd1 * T.dt;
    }
    template <class R>
    __device__ static void observe(const R* s, R* o) { o[0] = s[0]; o[1] = s[1]; }
    VS_OBSERVE_C_IS_OBSERVE
    __device__ static void sample_init(const Task&, const float*, Rng& g, float* init) {  // :59-60, box.py:169-178
        init[0] = g.uniform(-0.75f, -0.65f);
        init[1] = g.uniform(-0.1f, 0.1f);
    }
    __device__ static void state_from_init(const float* init, float* s) { s[0] = init[0]; s[1] = init[1]; }
    __device__ static void init_hidden(const Task&, const float*, const float*, const float*, float*, bool) {}
};

// =================================================================================================== BoB
// BallOnBeamSim, P/environments/pysim/ball_on_beam.py:41-136
// V = 1: BallOnBeamDiscSim (ball_on_beam.py:139-161): the action space is DiscreteSpace({-max, 0, +max})
template <int V>
struct BobT : EnvDefaults<1> {
    static constexpr int WS_MID_G3_FROM = 0;
    static constexpr int S = 4, A = 1, O = 4, H = 0, I = 4, P = 8, K = 9, KS = 9;
    static constexpr int REW = REW_SCALED_EXP, RADIAL = -1;
    static constexpr int WS_SHAPE_FULL = V == 0 ? 256 : 64;  // (the discrete action's snap makes its reward wave the longer one)
    // three waves per 64 envs since round 3 (measured once the kernels had lost their register bloat and the randomizer's unused
    // redraw): 65 536 envs 2.015e11 (256-env workgroups, two roles) -> 2.10e11, 4 096 envs 1.41e10 -> 1.45e10
    static constexpr bool WS_G3 = true;
    static constexpr int WS_G3_FULL = 64;
    static constexpr int WS_PREP_G64 = 1;  // 4 096 / 32 768 envs + 6.5 %, 65 536 + 1.5 %
    // DiscreteSpace.project_to (P/spaces/discrete.py:104-131): an action that is np.isclose to one of the elements is
    // kept as it is, anything else snaps to the closest element (argmin: the first of two equally close ones)
    template <class R>
    __device__ static void limit_act(const float* c, const float* lo, const float* hi, const R* a_raw, R* a) {
        if (V == 0) { EnvDefaults<1>::limit_act(c, lo, hi, a_raw, a); return; }
        float x = val(a_raw[0]);
        float e[3] = {lo[0], (lo[0] + hi[0]) * 0.5f, hi[0]};
        float tol = 1e-8f + 1e-5f * fabsf(x);  // np.isclose(eles, cand): atol + rtol * |cand|
        bool close = false;
        int best = 0;
        float bd = fabsf(x - e[0]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float dk = fabsf(x - e[k]);
            close |= dk <= tol;
            if (dk < bd) { bd = dk; best = k; }
        }
        a[0] = vsel(close || isnan(x), a_raw[0], R(e[best]));
    }
    __device__ static float sample_action(const float* c, float lo, float hi, float u01, int) {
        if (V == 0) return lo + (hi - lo) * u01;
        int k = min((int)(u01 * 3.0f), 2);  // np.random.randint(3)
        return k == 0 ? lo : (k == 1 ? (lo + hi) * 0.5f : hi);
    }
    enum { C_MG, C_M, C_FRICT, C_OFF, C_INV_ZETA_BALL, C_J_BEAM, C_XMAX, C_AMAX, C_CMAX };
    static constexpr int CMAX = C_CMAX;
    __device__ static void calc_consts(const Task& T, const float* p, float* c) {  // _calc_constants :89-98
        float g = p[0], m_ball = p[1], r_ball = p[2], m_beam = p[3], l_beam = p[4], d_beam = p[5];
        float J_ball = 2.0f / 5 * m_ball * r_ball * r_ball;
        c[C_J_BEAM] = 1.0f / 12 * m_beam * (l_beam * l_beam + d_beam * d_beam);
        c[C_INV_ZETA_BALL] = 1.0f / (m_ball + J_ball / (r_ball * r_ball));  // 1 / zeta_ball
        c[C_MG] = m_ball * g;
        c[C_M] = m_ball;
        c[C_FRICT] = p[6];
        c[C_OFF] = p[7];
        c[C_XMAX] = l_beam / 2.0f;               // _create_spaces :54
        c[C_AMAX] = l_beam / 2.0f * g * 3.0f;    // :55
        // ScaledExpQuadrErrRewFcn.reset (reward_functions.py:284-297), recomputed per env (Q11)
        float smax[4] = {c[C_XMAX], PI_4_F, 10.0f, PI_F};
        float mc = 0.f;
        for (int j = 0; j < 4; ++j) mc += smax[j] * (T.qd[j] * smax[j]);
        mc += c[C_AMAX] * (T.rd[0] * c[C_AMAX]);
        c[C_CMAX] = 9.210340371976182f / mc;  // -ln(1e-4)
    }
    __device__ static void state_bounds(const float* c, float* lo, float* hi) {
        hi[0] = c[C_XMAX]; hi[1] = PI_4_F; hi[2] = 10.0f; hi[3] = PI_F;
        for (int j = 0; j < 4; ++j) lo[j] = -hi[j];
    }
    __device__ static void act_bounds(const float* c, float* lo, float* hi) { hi[0] = c[C_AMAX]; lo[0] = -c[C_AMAX]; }
    template <class R>
    __device__ static void dynamics(const Task& T, const float* c, R* s, R*, const R* act, const R*) {  // :110-129
        R x = s[0], a = s[1] + c[C_OFF], x_dot = s[2], a_dot = s[3];
        R sa, ca;
        sincos_fast(a, &sa, &ca);
        R zeta_beam = c[C_M] * x * x + c[C_J_BEAM];
        R x_ddot = (-c[C_FRICT] * x_dot + c[C_M] * x * a_dot * a_dot - c[C_MG] * sa) * c[C_INV_ZETA_BALL];
        R a_ddot = (act[0] - 2.0f * c[C_M] * x * x_dot * a_dot - c[C_MG] * ca * x) * rcp_fast(zeta_beam);
        s[2] += x_ddot * T.dt;  // symplectic Euler: velocity first
        s[3] += a_ddot * T.dt;
        s[0] += s[2] * T.dt;
        s[1] += s[3] * T.dt;
    }
    template <class R>
    __device__ static void observe(const R* s, R* o) {
        for (int j = 0; j < 4; ++j) o[j] = s[j];
    }
    VS_OBSERVE_C_IS_OBSERVE
    __device__ static void sample_init(const Task&, const float* c, Rng& g, float* init) {  // :60-73, compound.py:84-87
        bool right = (g.next() & 1u) != 0u;  // np.random.randint(2)
        float l2 = c[C_XMAX];
        float x = g.uniform(0.7f * l2, 0.8f * l2);
        init[0] = right ? x : -x;  // box 0: [-0.8, -0.7] l/2, box 1: [0.7, 0.8] l/2
        init[1] = g.uniform(-5.0f / 180.0f * PI_F, 5.0f / 180.0f * PI_F);
        init[2] = g.uniform(-0.2f, 0.2f);
        init[3] = g.uniform(-0.02f * PI_F, 0.02f * PI_F);
    }
    __device__ static void state_from_init(const float* init, float* s) {
        for (int j = 0; j < 4; ++j) s[j] = init[j];
    }
    __device__ static void init_hidden(const Task&, const float*, const float*, const float*, float*, bool) {}
};
using Bob = BobT<0>;
using BobD = BobT<1>;

// =================================================================================================== QQube
// QQubeSwingUpSim, P/environments/pysim/quanser_qube.py:41-188
// V = 1: QQubeStabSim (quanser_qube.py:191-222): same dynamics and spaces, init space around the upright pendulum
template <int V>
struct QQT : EnvDefaults<1> {
    static constexpr int WS_MID_G3_FROM = 320;
    static constexpr int S = 4, A = 1, O = 6, H = 0, I = 4, P = 11, K = 11, KS = 11;
    static constexpr int REW = REW_EXP, RADIAL = 1, CMAX = -1;
    static constexpr bool REWARD_SIDE_USES_CONSTS = false;
    static constexpr bool WS_DRAW_P = true;
    static constexpr bool WS_G3 = true;
    static constexpr int WS_MIN_WAVES = 3;  // its two-role kernel runs three waves per SIMD between 256 and 384 envs per CU (WS_MID)
    enum { C_C0, C_C1, C_C2, C_C3, C_C4, C_KM, C_RM, C_DR, C_DP, C_TH_NEG, C_TH_POS };
    __device__ static void calc_consts(const Task&, const float* p, float* c) {  // _calc_constants :70-87
        float g = p[0], Rm = p[1], km = p[2], mr = p[3], Lr = p[4], Dr = p[5], mp = p[6], Lp = p[7], Dp = p[8];
        float Jr = mr * Lr * Lr / 12.0f;
        float Jp = mp * Lp * Lp / 12.0f;
        c[C_C0] = Jr + mp * Lr * Lr;
        c[C_C1] = 0.25f * mp * Lp * Lp;
        c[C_C2] = 0.5f * mp * Lp * Lr;
        c[C_C3] = Jp + c[C_C1];
        c[C_C4] = 0.5f * mp * Lp * g;
        c[C_KM] = km; c[C_RM] = 1.0f / Rm; c[C_DR] = Dr; c[C_DP] = Dp;
        c[C_TH_NEG] = p[9]; c[C_TH_POS] = p[10];
    }
    __device__ static void state_bounds(const float*, float* lo, float* hi) {  // _create_spaces :169
        hi[0] = QQ_TH_MAX_F; hi[1] = PI4_F; hi[2] = PI20_F; hi[3] = PI20_F;
        for (int j = 0; j < 4; ++j) lo[j] = -hi[j];
    }
    __device__ static void act_bounds(const float*, float* lo, float* hi) { hi[0] = 4.5f; lo[0] = -4.5f; }  // MAX_ACT_QQ
    // tr: (sin, cos)(alpha) of the PRE-step state when the caller has them in registers (fused rollout), else nullptr
    // BASELINE config 2 (4 096 envs): 98 -> 85 vector instructions on the physics wave's chain, 260 -> 241 ns per step (+ 8 %; 32 768 envs + 4 %)
    static constexpr int WS_PREP_G64 = 2;
    template <class R>
    __device__ static void dead_zone(const Task&, const float* c, R* a) {  // _step_dynamics :130-131
        if (c[C_TH_NEG] <= a[0] && a[0] <= c[C_TH_POS]) a[0] = 0.f;
    }
    template <class R>
    __device__ static void dynamics(const Task& T, const float* c, R* s, R* h, const R* act, const R* tr) {
        R u[1] = {act[0]};
        dead_zone(T, c, u);
        dynamics_core(T, c, s, h, u, tr);
    }
    // the step behind the dead zone (ua: the voltage that reaches the motor)
    template <class R>
    __device__ static void dynamics_core(const Task& T, const float* c, R* s, R*, const R* ua, const R* tr) {
        R u = ua[0];
        // _dyn :89-125, evaluated once: the reference's "RK4" re-evaluates _dyn at self.state in every stage (Q1), so
        // k_j differ only in their position-derivative slots and the update collapses to
        //   v' = v + dt a,  p' = p + dt v + dt^2/2 a        (closed form verified against the oracle: max abs diff 0)
        R thd = s[2], ald = s[3];
        R sin_al, cos_al;
        if (tr) { sin_al = tr[0]; cos_al = tr[1]; }  // observe() already holds sin/cos(alpha) of this state
        else sincos_fast(s[1], &sin_al, &cos_al);
        R sin_2al = 2.0f * sin_al * cos_al;
        R a = c[C_C0] + c[C_C1] * sin_al * sin_al;
        R b = c[C_C2] * cos_al;
        R cc = c[C_C3];
        R det = a * cc - b * b;
        R trq = c[C_KM] * (u - c[C_KM] * thd) * c[C_RM];  // C_RM holds 1 / Rm
        R c0 = c[C_C1] * sin_2al * thd * ald - c[C_C2] * sin_al * ald * ald;
        R c1 = -0.5f * c[C_C1] * sin_2al * thd * thd + c[C_C4] * sin_al;
        R x = trq - c[C_DR] * thd - c0;
        R y = -c[C_DP] * ald - c1;
        R inv_det = rcp_fast(det);
        R thdd = (cc * x - b * y) * inv_det;
        R aldd = (a * y - b * x) * inv_det;
        R dt = T.dt, hdt2 = 0.5f * dt * dt;
        s[0] = s[0] + dt * thd + hdt2 * thdd;
        s[1] = s[1] + dt * ald + hdt2 * aldd;
        s[2] = thd + dt * thdd;
        s[3] = ald + dt * aldd;
    }
    template <class R>
    __device__ static void observe(const R* s, R* o) {  // :148-149
        sincos_fast(s[0], &o[0], &o[1]);
        sincos_fast(s[1], &o[2], &o[3]);
        o[4] = s[2];
        o[5] = s[3];
    }
    static constexpr int TRIG = 2, TRIG_AT = 2;
    template <class R>
    __device__ static void observe_p(const R* s, R* tr) { sincos_fast(s[1], &tr[0], &tr[1]); }
    template <class R>
    __device__ static void observe_c(const R* s, const R* tr, R* o) {
        sincos_fast(s[0], &o[0], &o[1]);
        o[2] = tr[0];
        o[3] = tr[1];
        o[4] = s[2];
        o[5] = s[3];
    }
    __device__ static void sample_init(const Task&, const float*, Rng& g, float* init) {  // :170
        const float d2r = PI_F / 180.0f;
        if (V == 1) {  // :207-208: theta +-5 deg, alpha 175..185 deg, zero velocities
            init[0] = g.uniform(-5.0f * d2r, 5.0f * d2r);
            init[1] = g.uniform(175.0f * d2r, 185.0f * d2r);
            init[2] = init[3] = 0.f;
            return;
        }
        init[0] = g.uniform(-2.0f * d2r, 2.0f * d2r);
        init[1] = g.uniform(-1.0f * d2r, 1.0f * d2r);
        init[2] = g.uniform(-0.5f * d2r, 0.5f * d2r);
        init[3] = g.uniform(-0.5f * d2r, 0.5f * d2r);
    }
    __device__ static void state_from_init(const float* init, float* s) {
        for (int j = 0; j < 4; ++j) s[j] = init[j];
    }
    __device__ static void init_hidden(const Task&, const float*, const float*, const float*, float*, bool) {}
};
using QQ = QQT<0>;
using QQSt = QQT<1>;

// =================================================================================================== QCartPole
// QCartPoleSwingUpSim, P/environments/pysim/quanser_cartpole.py:45-230, 507-587; rk4 591-655
// V = 1: QCartPoleStabSim (quanser_cartpole.py:441-504): the pole has to stay within +-15 deg of upright; quadratic
// reward; FinalRewTask(state_dependent, time_dependent)
template <int V>
struct QcpT : EnvDefaults<1> {
    static constexpr int S = 4, A = 1, O = 5, H = 1, I = 4, P = 17, K = 16, KS = 16;
    static constexpr int REW = V == 1 ? REW_QUADR : REW_EXP, RADIAL = 1, CMAX = -1;
    static constexpr bool REWARD_SIDE_USES_CONSTS = false;
    // (64-env workgroups at 65 536 envs since round 3: with the physics wave issuing first (s_setprio) 247.8 against 255.9 us per
    // 400 recorded steps in 256-env workgroups -- the inherited default; round 2 had measured the two the other way round)
    static constexpr bool WS_MID = false;
    static constexpr int WS_PREP_C = 1;  // a chain of four dependent _dynamics evaluations: every instruction off it counts
    // three waves per 64 envs since round 3 (the swing-up task; the stabilisation task's final reward keeps it on k_rollout): with
    // 125 instead of 256 registers the shape fits, and the generator wave takes the draws and the reset stock -- under a live
    // randomizer the stock's refill passes (~9 000 cycles each) -- off the reward wave the physics wave waits for:
    // 65 536 envs + 7 randomised parameters 1.033e11 -> 1.046e11, without a randomizer 1.109e11 -> 1.125e11 (64-env workgroups)
    static constexpr bool WS_G3 = V == 0;
    static constexpr int WS_G3_FULL = 64;
    static constexpr int WS_MIN_WAVES = 2;  // 65 536 envs in 64-env workgroups are two waves per SIMD: at most 256 VGPRs
    static constexpr bool SYMMETRIC_BOX = V == 0;
    static constexpr int FINAL = V == 1 ? FINAL_STATE_TIME : FINAL_NONE;
    enum { C_KA, C_ETA_M, C_KB, C_MTG, C_MPL2, C_MU, C_M00, C_MPL, C_M11, C_BEQ, C_BP, C_MPLG, C_TH_NEG, C_TH_POS,
           C_XMAX, C_XDMAX };
    __device__ static void calc_consts(const Task&, const float* p, float* c) {  // _calc_constants :145-155 + _dynamics
        float g = p[0], m_c = p[1], l_rail = p[2], eta_m = p[3], eta_g = p[4], K_g = p[5], J_m = p[6], r_mp = p[7],
              R_m = p[8], k_m = p[9], B_p = p[10], B_eq = p[11], m_p = p[12], l_p = p[13], mu_c = p[14];
        float J_pole = l_p * l_p * m_p / 3.0f;
        float J_eq = m_c + (eta_g * K_g * K_g * J_m) / (r_mp * r_mp);
        c[C_KA] = (eta_g * K_g * eta_m * k_m) / (R_m * r_mp);  // f_act prefactor :195
        c[C_ETA_M] = eta_m;
        c[C_KB] = K_g * k_m / r_mp;
        c[C_MTG] = (m_c + m_p) * g;        // f_normal :202
        c[C_MPL2] = m_p * l_p / 2.0f;
        c[C_MU] = mu_c;
        c[C_M00] = m_p + J_eq;             // mass matrix :211-216
        c[C_MPL] = m_p * l_p;
        c[C_M11] = J_pole + m_p * l_p * l_p;
        c[C_BEQ] = B_eq;
        c[C_BP] = B_p;
        c[C_MPLG] = m_p * l_p * g;
        c[C_TH_NEG] = p[15];
        c[C_TH_POS] = p[16];
        c[C_XMAX] = l_rail / 2.0f - 0.15f;  // _create_spaces :546-551, _x_buffer :77
        c[C_XDMAX] = l_rail;
    }
    __device__ static void state_bounds(const float* c, float* lo, float* hi) {
        if (V == 1) {  // :478-483, stab_thold = 15 deg
            hi[0] = c[C_XMAX]; lo[0] = -c[C_XMAX];
            lo[1] = (float)(PI_D - 15.0 / 180.0 * PI_D); hi[1] = (float)(PI_D + 15.0 / 180.0 * PI_D);
            hi[2] = c[C_XDMAX]; lo[2] = -c[C_XDMAX];
            hi[3] = (float)(2.0 * PI_D); lo[3] = -hi[3];
            return;
        }
        hi[0] = c[C_XMAX]; hi[1] = PI4_F; hi[2] = c[C_XDMAX]; hi[3] = PI20_F;
        for (int j = 0; j < 4; ++j) lo[j] = -hi[j];
    }
    __device__ static void act_bounds(const float*, float* lo, float* hi) { hi[0] = 6.0f; lo[0] = -6.0f; }  // MAX_ACT_QCP
    // one evaluation of QCartPoleSim._dynamics (:166-230) on the augmented state y = [x, th, x_dot, th_dot], action u
    template <class R>
    __device__ __forceinline__ static void f_dyn(const Task& T, const float* c, const R* y, R u, R thdd_prev, R* k, R& thdd_out, const R* tr) {
        R th = y[1], x_dot = y[2], th_dot = y[3];
        R sin_th, cos_th;
        if (tr) { sin_th = tr[0]; cos_th = tr[1]; }
        else sincos_fast(th, &sin_th, &cos_th);
        bool simple = (T.flags & 1) != 0;
        if (!simple && c[C_TH_NEG] <= u && u <= c[C_TH_POS]) u = 0.f;  // dead zone :188-192
        R f_act = c[C_KA] * (c[C_ETA_M] * u - c[C_KB] * x_dot);
        R f_tot = f_act;
        // Coulomb friction :199-208, f_c = 0 if f_normal < 0 else mu_c f_normal sign(x_dot).  This function is evaluated four times
        // per step on the ONE wave the cartpole's kernels wait for (~270 dependent vector instructions per step):
        // every instruction here counts four times.  Round 3 (float path; the Jacobian kernel's Dual path keeps the plain form):
        // the product with sign(x_dot) as a sign-bit flip, zero where x_dot == 0 or f_normal < 0 (exactly v * np.sign(x): - 3
        // instructions), and the whole term computed unconditionally and dropped by a select for simple_dynamics (a wave-uniform
        // branch per stage cut the step into eight basic blocks)
        if constexpr (std::is_same<R, float>::value) {
            float f_normal = c[C_MTG] - c[C_MPL2] * (sin_th * thdd_prev + cos_th * th_dot * th_dot);
            float t = c[C_MU] * f_normal;
            float f_c = __uint_as_float(__float_as_uint(t) ^ (__float_as_uint(x_dot) & 0x80000000u));  // t * sign(x_dot), x_dot != 0
            f_c = (f_normal < 0.f || x_dot == 0.f) ? 0.f : f_c;
            f_tot = simple ? f_act : f_act - f_c;
        } else if (!simple) {
            R f_normal = c[C_MTG] - c[C_MPL2] * (sin_th * thdd_prev + cos_th * th_dot * th_dot);
            R f_c = vsel(f_normal < 0.f, R(0.f), c[C_MU] * f_normal * sgnf(val(x_dot)));
            f_tot = f_act - f_c;
        }
        R M01 = c[C_MPL] * cos_th;
        R r0 = f_tot - c[C_BEQ] * x_dot - c[C_MPL] * sin_th * th_dot * th_dot;
        R r1 = -c[C_BP] * th_dot - c[C_MPLG] * sin_th;
        // np.linalg.solve on the SPD 2x2 -> closed form; the determinant's reciprocal by the bare v_rcp_f32 (1 ulp: the Newton step
        // of rcp_fast was two more instructions per stage for the last half ulp of an fp32 result compared at 1e-5)
        R inv_det = rcp_det(c[C_M00] * c[C_M11] - M01 * M01);
        R x_ddot = (c[C_M11] * r0 - M01 * r1) * inv_det;
        R th_ddot = (c[C_M00] * r1 - M01 * r0) * inv_det;
        k[0] = x_dot + x_ddot * T.dt;  // already Euler-advanced velocities as position derivative (Q6, :227-230)
        k[1] = th_dot + th_ddot * T.dt;
        k[2] = x_ddot;
        k[3] = th_ddot;
        thdd_out = th_ddot;
    }
    // stages 2 .. 4 of the rk4 below and its final combination; ROT: sin / cos of the stage angles by rotation from the
    // step's own (tr0).  Everything between the first stage and the new state sits inside ONE side of the caller's
    // wave-uniform branch: the stage vectors stay in registers (arrays that crossed the branch went to scratch)
    template <bool ROT, class R>
    __device__ __forceinline__ static void rk_tail(const Task& T, const float* c, R* s, R* h, R u, const R* k1, R a1,
                                                   const R* tr0) {
        R dt = T.dt, dt2 = dt / 2.0f;
        R y[4], trs[2], k2[4], k3[4], k4[4], a2, a3, a4;
        for (int j = 0; j < 4; ++j) y[j] = s[j] + dt2 * k1[j];
        if (ROT) sincos_rot(tr0[0], tr0[1], dt2 * k1[1], &trs[0], &trs[1]);
        f_dyn(T, c, y, u, a1, k2, a2, ROT ? (const R*)trs : (const R*)nullptr);
        for (int j = 0; j < 4; ++j) y[j] = s[j] + dt2 * k2[j];
        if (ROT) sincos_rot(tr0[0], tr0[1], dt2 * k2[1], &trs[0], &trs[1]);
        f_dyn(T, c, y, u, a2, k3, a3, ROT ? (const R*)trs : (const R*)nullptr);
        for (int j = 0; j < 4; ++j) y[j] = s[j] + dt * k3[j];
        if (ROT) sincos_rot(tr0[0], tr0[1], dt * k3[1], &trs[0], &trs[1]);
        f_dyn(T, c, y, u, a3, k4, a4, ROT ? (const R*)trs : (const R*)nullptr);
        for (int j = 0; j < 4; ++j) s[j] = s[j] + dt / 6.0f * (k1[j] + 2.0f * k2[j] + 2.0f * k3[j] + k4[j]);
        h[0] = (a1 + a2 + a3 + a4) / 4.0f;  // mean of the stage th_ddots (:652)
    }
    template <class R>
    __device__ __forceinline__ static void dynamics(const Task& T, const float* c, R* s, R* h, const R* act, const R* tr) {
        // (force-inlined: with two copies of the later stages the inliner left it a real call -- the constants and the state
        // then lived in scratch and the kernel ran three times slower)
        // rk4 (:591-655) over [x, th, x_dot, th_dot, u]; u has zero derivative; th_ddot chained through the stages.
        // The four stage vectors live in VGPRs (16 floats per lane): there is no cross-lane reuse to stage in LDS.
        R u = act[0];
        R k1[4], a1;
        // sin / cos of the pole angle: of the step's own angle once (tr: the caller holds it from observe()), of the three later
        // stage angles by rotation from it -- they differ from it by dt/2 or dt times a pole rate, at most ~0.26 rad for
        // dt <= 4 ms inside (and well beyond) the 20 pi rad/s state box; coarser steps take a fresh sincos per stage
        R tr0[2];
        if (tr) { tr0[0] = tr[0]; tr0[1] = tr[1]; }
        else sincos_fast(s[1], &tr0[0], &tr0[1]);
        f_dyn(T, c, s, u, h[0], k1, a1, (const R*)tr0);
        if (T.dt <= 0.004f) rk_tail<true>(T, c, s, h, u, k1, a1, tr0);  // wave-uniform: dt is a kernel argument
        else rk_tail<false>(T, c, s, h, u, k1, a1, tr0);
    }
    template <class R>
    __device__ static void observe(const R* s, R* o) {  // :107-108
        o[0] = s[0];
        sincos_fast(s[1], &o[1], &o[2]);
        o[3] = s[2];
        o[4] = s[3];
    }
    static constexpr int TRIG = 2, TRIG_AT = 1;
    template <class R>
    __device__ static void observe_p(const R* s, R* tr) { sincos_fast(s[1], &tr[0], &tr[1]); }
    template <class R>
    __device__ static void observe_c(const R* s, const R* tr, R* o) {
        o[0] = s[0];
        o[1] = tr[0];
        o[2] = tr[1];
        o[3] = s[2];
        o[4] = s[3];
    }
    __device__ static void sample_init(const Task& T, const float*, Rng& g, float* init) {  // :552-560
        if (V == 1) {  // :485-490, max_init_th_offset = 8 deg
            init[0] = g.uniform(-0.02f, 0.02f);
            init[1] = g.uniform((float)(PI_D - 8.0 / 180.0 * PI_D), (float)(PI_D + 8.0 / 180.0 * PI_D));
            init[2] = g.uniform(-0.02f, 0.02f);
            init[3] = g.uniform((float)(-5.0 / 180.0 * PI_D), (float)(5.0 / 180.0 * PI_D));
            return;
        }
        float hi[4];
        if (T.wild_init == 0) { hi[0] = 0.25f; hi[1] = PI_F; hi[2] = 0.8f; hi[3] = PI_F; }
        else if (T.wild_init == 1) { hi[0] = 0.02f; hi[1] = 2.0f / 180.0f * PI_F; hi[2] = 0.f; hi[3] = 1.0f / 180.0f * PI_F; }
        else { hi[0] = 0.02f; hi[1] = PI_F; hi[2] = 0.f; hi[3] = 1.0f / 180.0f * PI_F; }
        for (int j = 0; j < 4; ++j) init[j] = g.uniform(-hi[j], hi[j]);
    }
    __device__ static void state_from_init(const float* init, float* s) {
        for (int j = 0; j < 4; ++j) s[j] = init[j];
    }
    __device__ static void init_hidden(const Task&, const float*, const float*, const float*, float* h, bool) {
        h[0] = 0.f;  // reset(): self._th_ddot = 0.0 (:103)
    }
};
using Qcp = QcpT<0>;
using QcpSt = QcpT<1>;

// =================================================================================================== Pendulum
// PendulumSim, P/environments/pysim/pendulum.py:43-117
struct Pend : EnvDefaults<1> {
    static constexpr int WS_MID_G3_FROM = 0;
    static constexpr int WS_PREP_G64 = 1;  // 4 096 envs + 4 %
    static constexpr int S = 2, A = 1, O = 3, H = 0, I = 2, P = 5, K = 4, KS = 4;
    // idcs=[1] in the reference (pendulum.py:87): the 2pi modulo is applied to the theta_dot error
    static constexpr int REW = REW_EXP, RADIAL = 1, CMAX = -1;
    static constexpr bool WS_DRAW_P = true;
    static constexpr bool WS_G3 = true;
    enum { C_MGL2, C_DAMP, C_INV_J, C_AMAX };
    __device__ static void calc_consts(const Task&, const float* p, float* c) {
        float g = p[0], m = p[1], l = p[2];
        c[C_MGL2] = m * g * l / 2.0f;            // :104
        c[C_DAMP] = p[3];
        c[C_INV_J] = 1.0f / (m * l * l / 3.0f);  // rod about its end
        c[C_AMAX] = p[4];                        // torque_thold, _create_spaces :73-78
    }
    __device__ static void state_bounds(const float*, float* lo, float* hi) {  // :71
        hi[0] = PI4_F; hi[1] = PI4_F; lo[0] = -PI4_F; lo[1] = -PI4_F;
    }
    __device__ static void act_bounds(const float* c, float* lo, float* hi) { hi[0] = c[C_AMAX]; lo[0] = -c[C_AMAX]; }
    template <class R>
    __device__ static void dynamics(const Task& T, const float* c, R* s, R*, const R* act, const R* tr) {
        R sn, cs;
        if (tr) sn = tr[0];
        else sincos_fast(s[0], &sn, &cs);
        R th_ddot = (act[0] - c[C_MGL2] * sn - c[C_DAMP] * s[1]) * c[C_INV_J];  // :103-106
        s[1] += th_ddot * T.dt;  // symplectic Euler :109-110
        s[0] += s[1] * T.dt;
    }
    template <class R>
    __device__ static void observe(const R* s, R* o) {  // :91-92
        sincos_fast(s[0], &o[0], &o[1]);
        o[2] = s[1];
    }
    static constexpr int TRIG = 2, TRIG_AT = 0;
    template <class R>
    __device__ static void observe_p(const R* s, R* tr) { sincos_fast(s[0], &tr[0], &tr[1]); }
    template <class R>
    __device__ static void observe_c(const R* s, const R* tr, R* o) {
        o[0] = tr[0];
        o[1] = tr[1];
        o[2] = s[1];
    }
    __device__ static void sample_init(const Task& T, const float*, Rng&, float* init) {  // SingularStateSpace :76
        init[0] = T.init_fixed[0];
        init[1] = T.init_fixed[1];
    }
    __device__ static void state_from_init(const float* init, float* s) { s[0] = init[0]; s[1] = init[1]; }
    __device__ static void init_hidden(const Task&, const float*, const float*, const float*, float*, bool) {}
};

// =================================================================================================== QBB
// QBallBalancerKin.__call__ (quanser_ball_balancer.py:375-444, Q8): 100 iterations of torch-fp32 SGD(lr .01, mom .9) on
// the rod tip, gradient written out the way autograd evaluates it (pow_backward g*(2x), sqrt backward g/(2*result))
__device__ inline float qbb_ik(float th, float r, float l) {
    const float d = 0.10f, lr = 0.01f, mom = 0.9f;
    float t0 = r, t1 = l;
    float sn, cs;
    sincosf(th, &sn, &cs);
    float rc = r * cs, rs = r * sn;
    float b0 = 0.f, b1 = 0.f;
    for (int it = 0; it < 100; ++it) {
        float dx = t0 - rc, dy = t1 - rs;
        float rod = sqrtf(dx * dx + dy * dy);
        float ex = t0 - r - l, ey = t1 - d;
        float half = sqrtf(ex * ex + ey * ey);
        float gu1 = (2.0f * (rod - d)) / (2.0f * rod);
        float gu2 = (2.0f * (half - l)) / (2.0f * half);
        float g0 = gu1 * (2.0f * dx) + gu2 * (2.0f * ex);
        float g1 = gu1 * (2.0f * dy) + gu2 * (2.0f * ey);
        if (it == 0) { b0 = g0; b1 = g1; }
        else { b0 = b0 * mom + g0; b1 = b1 * mom + g1; }
        t0 -= lr * b0;
        t1 -= lr * b1;
    }
    return PI_F / 2.0f - atan2f(r + l - t0, t1 - d);
}

// QBallBalancerSim, P/environments/pysim/quanser_ball_balancer.py:49-337
struct Qbb : EnvDefaults<2> {
    static constexpr int S = 8, A = 2, O = 8, H = 2, I = 4, P = 20, K = 19, KS = 17;
    static constexpr int REW = REW_SCALED_EXP, RADIAL = -1;
    // A physics wave more than twice the reward wave and ~400 VGPRs: 64-env workgroups while each of the two waves has a SIMD
    // of its own (up to 128 envs per compute unit), 256-env ones (capped at 256 VGPRs) up to 256; with the reward wave
    // pre-processing the actions (WS_PREP_C) 62 against k_rollout's 100 us per 100 steps at 32 768 envs, 83 against 107 at 65 536
    static constexpr int WS_SHAPE_FULL = 256, WS_SMALL = 128;
    static constexpr bool WS_MID = false;
    static constexpr bool WS_ALONE = true;
    // three waves per 64 envs since round 3: 65 536 envs 1.075e11 (256-env workgroups, two roles) -> 1.127e11, 4 096 envs
    // 8.26e9 -> 8.44e9, 32 768 envs (BASELINE config 4) unchanged (6.67e10)
    static constexpr bool WS_G3 = true;
    static constexpr int WS_G3_FULL = 64;
    enum { C_AM, C_BEQV, C_JEQ, C_CKIN, C_OFFX, C_OFFY, C_TXP, C_TXN, C_TYP, C_TYN, C_BDR2, C_JBR, C_MR2, C_CKMGR2,
           C_ZETA, C_XMAX, C_CMAX, C_IK_X0, C_IK_Y0 };
    static constexpr int CMAX = C_CMAX;
    __device__ static void calc_consts(const Task& T, const float* p, float* c) {  // _calc_constants :204-223
        float g = p[0], m_ball = p[1], r_ball = p[2], l_plate = p[3], r_arm = p[4], K_g = p[5], eta_g = p[6],
              J_l = p[7], J_m = p[8], k_m = p[9], R_m = p[10], eta_m = p[11], B_eq = p[12], ball_damping = p[13];
        float J_ball = 2.0f / 5 * m_ball * r_ball * r_ball;
        float c_kin = 2.0f * r_arm / l_plate;
        float r2 = r_ball * r_ball;
        c[C_AM] = eta_g * K_g * eta_m * k_m / R_m;
        c[C_BEQV] = eta_g * K_g * K_g * eta_m * k_m * k_m / R_m + B_eq;
        c[C_JEQ] = 1.0f / (eta_g * K_g * K_g * J_m + J_l);  // 1 / J_eq
        c[C_CKIN] = c_kin;
        c[C_OFFX] = p[18]; c[C_OFFY] = p[19];
        c[C_TXP] = p[14]; c[C_TXN] = p[15]; c[C_TYP] = p[16]; c[C_TYN] = p[17];
        c[C_BDR2] = ball_damping * r2;            // friction term :313
        c[C_JBR] = J_ball * r_ball;               // plate influence :314
        c[C_MR2] = m_ball * r2;                   // centripetal :315
        c[C_CKMGR2] = c_kin * m_ball * g * r2;    // gravity :316
        c[C_ZETA] = 1.0f / (m_ball * r2 + J_ball);  // 1 / zeta
        c[C_XMAX] = l_plate / 2.0f;               // _create_spaces :97-107
        float smax[8] = {PI_4_F, PI_4_F, c[C_XMAX], c[C_XMAX], PI5_F, PI5_F, 0.5f, 0.5f};
        float mc = 0.f;
        for (int j = 0; j < 8; ++j) mc += smax[j] * (T.qd[j] * smax[j]);
        float ma = 0.f;
        for (int j = 0; j < 2; ++j) ma += 3.0f * (T.rd[j] * 3.0f);  // MAX_ACT_QBB
        c[C_CMAX] = 9.210340371976182f / (mc + ma);
        // plate angles of the init-space reset (servo angles 0): depend on the params only -> cached (reset :238-242)
        bool simple = (T.flags & 1) != 0;
        c[C_IK_X0] = simple ? 0.f : qbb_ik(0.f + c[C_OFFX], r_arm, l_plate / 2.0f);
        c[C_IK_Y0] = simple ? 0.f : qbb_ik(0.f + c[C_OFFY], r_arm, l_plate / 2.0f);
    }
    __device__ static void state_bounds(const float* c, float* lo, float* hi) {
        hi[0] = PI_4_F; hi[1] = PI_4_F; hi[2] = c[C_XMAX]; hi[3] = c[C_XMAX];
        hi[4] = PI5_F; hi[5] = PI5_F; hi[6] = 0.5f; hi[7] = 0.5f;
        for (int j = 0; j < 8; ++j) lo[j] = -hi[j];
    }
    __device__ static void act_bounds(const float*, float* lo, float* hi) {
        hi[0] = hi[1] = 3.0f; lo[0] = lo[1] = -3.0f;  // MAX_ACT_QBB
    }
    static constexpr int WS_PREP_C = 2;
    template <class R>
    __device__ static void dead_zone(const Task& T, const float* c, R* a) {  // :261-264
        bool simple = (T.flags & 1) != 0;
        if (!simple && c[C_TXN] <= a[0] && a[0] <= c[C_TXP]) a[0] = 0.f;
        if (!simple && c[C_TYN] <= a[1] && a[1] <= c[C_TYP]) a[1] = 0.f;
    }
    template <class R>
    __device__ static void dynamics(const Task& T, const float* c, R* s, R* h, const R* act, const R* tr) {  // :247-330
        R u[2] = {act[0], act[1]};
        dead_zone(T, c, u);
        dynamics_core(T, c, s, h, u, tr);
    }
    // the step behind the dead zones (u: the voltages that reach the servos)
    template <class R>
    __device__ static void dynamics_core(const Task& T, const float* c, R* s, R* h, const R* u, const R*) {
        bool simple = (T.flags & 1) != 0;
        R a0 = u[0], a1 = u[1];
        R th_x = s[0] + c[C_OFFX], th_y = s[1] + c[C_OFFY];
        R x = s[2], y = s[3], th_x_dot = s[4], th_y_dot = s[5], x_dot = s[6], y_dot = s[7];
        R th_x_ddot = (c[C_AM] * a0 - c[C_BEQV] * th_x_dot) * c[C_JEQ];  // C_JEQ holds 1 / J_eq
        R th_y_ddot = (c[C_AM] * a1 - c[C_BEQV] * th_y_dot) * c[C_JEQ];
        R sx, cx, sy, cy, sa, ca, sb, cb;
        sincos_fast(th_x, &sx, &cx);
        sincos_fast(h[0], &sa, &ca);
#ifdef VS_QBB_HALF  // diagnostic builds only: what ONE axis of the ball balancer costs (results are wrong)
        sy = sx, cy = cx, sb = sa, cb = ca;
        th_y_dot = th_x_dot, y = x, y_dot = x_dot, a1 = a0;
#else
        sincos_fast(th_y, &sy, &cy);
        sincos_fast(h[1], &sb, &cb);
#endif
        R ck = c[C_CKIN];
        R inv_ca = rcp_fast(ca), inv_cb = rcp_fast(cb);
        R a_dot = ck * th_x_dot * cx * inv_ca;
        R b_dot = ck * -th_y_dot * cy * inv_cb;  // cos(-th_y) = cos(th_y)
        R x_ddot, y_ddot;
        if (simple) {
            x_ddot = c[C_CKMGR2] * sx * c[C_ZETA];  // C_ZETA holds 1 / zeta
            y_ddot = c[C_CKMGR2] * sy * c[C_ZETA];
        } else {
            R a_ddot = inv_ca * (ck * (th_x_ddot * cx - th_x_dot * th_x_dot * sx) + a_dot * a_dot * sa);
            // -(-th_y_dot)^2 * sin(-th_y) = + th_y_dot^2 * sin(th_y)
            R b_ddot = inv_cb * (ck * (-th_y_ddot * cy + th_y_dot * th_y_dot * sy) + b_dot * b_dot * sb);
            x_ddot = (-c[C_BDR2] * x_dot - c[C_JBR] * a_ddot + c[C_MR2] * x * a_dot * a_dot + c[C_CKMGR2] * sx) * c[C_ZETA];
            y_ddot = (-c[C_BDR2] * y_dot - c[C_JBR] * b_ddot + c[C_MR2] * y * b_dot * b_dot + c[C_CKMGR2] * sy) * c[C_ZETA];
        }
        R dt = T.dt;
        s[4] += th_x_ddot * dt; s[5] += th_y_ddot * dt; s[6] += x_ddot * dt; s[7] += y_ddot * dt;  // symplectic Euler
        s[0] += s[4] * dt; s[1] += s[5] * dt; s[2] += s[6] * dt; s[3] += s[7] * dt;
        h[0] += a_dot * dt;  // forward Euler on the plate angles :330
        h[1] += b_dot * dt;
    }
    template <class R>
    __device__ static void observe(const R* s, R* o) {
        for (int j = 0; j < 8; ++j) o[j] = s[j];
    }
    VS_OBSERVE_C_IS_OBSERVE
    __device__ static void sample_init(const Task&, const float* c, Rng& g, float* init) {  // :108-117, polar.py:108-113
        float l2 = c[C_XMAX];
        float r = g.uniform(0.75f * l2, 0.8f * l2);
        float phi = g.uniform(-PI_F, PI_F);
        float sp, cp;
        sincosf(phi, &sp, &cp);
        init[0] = r * cp;
        init[1] = r * sp;
        init[2] = g.uniform(-0.025f, 0.025f);
        init[3] = g.uniform(-0.025f, 0.025f);
    }
    __device__ static void state_from_init(const float* init, float* s) {  // _state_from_init :225-229
        s[0] = s[1] = s[4] = s[5] = 0.f;
        s[2] = init[0]; s[3] = init[1]; s[6] = init[2]; s[7] = init[3];
    }
    // reset(): plate_angs = IK(th + offset) (:231-245). From the init space the servo angles are 0 -> cached constants;
    // a full-state init with non-zero servo angles runs the IK (needs arm_radius / plate_length from the raw params).
    __device__ static void init_hidden(const Task& T, const float* c, const float* p, const float* s, float* h,
                                       bool full_state) {
        if ((T.flags & 1) != 0) { h[0] = h[1] = 0.f; return; }
        if (!full_state || (s[0] == 0.f && s[1] == 0.f)) { h[0] = c[C_IK_X0]; h[1] = c[C_IK_Y0]; return; }
        h[0] = qbb_ik(s[0] + c[C_OFFX], p[4], p[3] / 2.0f);
        h[1] = qbb_ik(s[1] + c[C_OFFY], p[4], p[3] / 2.0f);
    }
};

}  // namespace vs
