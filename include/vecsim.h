/*
 * vecsim.h -- C-ABI of libvecsim: an MI355X (gfx950) native, batched, device-resident stepper for Pyrado's
 * pure-Python simulated robots (SimPyEnv subclasses).
 *
 * The reference (swami1995/SimuRLacra, Pyrado) has no FFI for this path: the boundary is the Python duck type
 * Env/SimEnv consumed by rollout(), the env wrappers and the samplers.  Each entry point below is the batched
 * (N environments, one per wavefront lane) replacement of one reference method; the host-side mirror of the Python
 * surface lives in simurlacra_amd/ and calls these through ctypes (see INTEGRATION.md for the stub).
 * `P/` = Pyrado/pyrado/ in the reference tree.
 *
 * Conventions
 *   - every function returns an int status: 0 = VS_OK, < 0 = error (vs_last_error() gives the text); nothing throws;
 *   - the library owns all device buffers; callers get raw device pointers (vs_get) and never free them;
 *   - one handle <-> one HIP device + one stream; a handle is not re-entrant, independent handles are thread-safe;
 *   - all per-env arrays are fp32 struct-of-arrays [dim][ld] with row pitch ld = vs_ld() >= n_envs (rows 256-B
 *     aligned); done/err flags are uint8 [ld], step counters int32 [ld];
 *   - pointers passed IN (params, init states, actions, masks) may be host or device memory (detected with
 *     hipPointerGetAttributes) unless stated otherwise; SoA inputs use row pitch n_envs when they come from the
 *     host and an explicit pitch argument when they are device pointers.
 */
#ifndef VECSIM_H
#define VECSIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VS_OK 0
#define VS_ERR_ARG (-1)     /* bad argument (shape, enum, null) -- pyrado.ShapeErr / ValueErr / TypeErr on the host side */
#define VS_ERR_HIP (-2)     /* a HIP runtime call failed (no device, OOM, launch error) */
#define VS_ERR_STATE (-3)   /* call order / handle state */
#define VS_ERR_NAN (-4)     /* NaN seen in an action or a state: BoxSpace.contains raises ValueErr, P/spaces/box.py:142-146 */

/* environment families (the `name` attribute of the reference classes) */
enum vs_env_type {
    VS_ENV_OMO = 0,    /* "omo"    OneMassOscillatorSim  P/environments/pysim/one_mass_oscillator.py:49-121 */
    VS_ENV_BOB = 1,    /* "bob"    BallOnBeamSim         P/environments/pysim/ball_on_beam.py:41-136 */
    VS_ENV_QQ_SU = 2,  /* "qq-su"  QQubeSwingUpSim       P/environments/pysim/quanser_qube.py:41-188 */
    VS_ENV_QCP_SU = 3, /* "qcp-su" QCartPoleSwingUpSim   P/environments/pysim/quanser_cartpole.py:45-230,507-587 */
    VS_ENV_QBB = 4,    /* "qbb"    QBallBalancerSim      P/environments/pysim/quanser_ball_balancer.py:49-337 */
    /* the remaining pysim families, sharing the kernels above */
    VS_ENV_QQ_ST = 5,  /* "qq-st"  QQubeStabSim          P/environments/pysim/quanser_qube.py:191-222 */
    VS_ENV_QCP_ST = 6, /* "qcp-st" QCartPoleStabSim      P/environments/pysim/quanser_cartpole.py:441-504 (flags default to long + simple) */
    VS_ENV_PEND = 7,   /* "pend"   PendulumSim           P/environments/pysim/pendulum.py:43-117 */
    VS_ENV_BOB_D = 8,  /* "bob-d"  BallOnBeamDiscSim     P/environments/pysim/ball_on_beam.py:139-161 */
    VS_ENV_COUNT = 9
};

/* buffers addressable through vs_get / vs_copy_to_host / vs_copy_from_host */
enum vs_buffer {
    VS_STATE = 0,      /* f32 [S][ld]  Env.state                                  P/environments/base.py:65 */
    VS_OBS = 1,        /* f32 [O][ld]  observe(state) after the last step/reset   P/environments/pysim/base.py:241 */
    VS_REW = 2,        /* f32 [ld]     reward of the last step (incl. final reward) base.py:220,237-239 */
    VS_DONE = 3,       /* u8  [ld]     done flag of the last step                 base.py:232-235 */
    VS_HIDDEN = 4,     /* f32 [H][ld]  qcp: _th_ddot, qbb: plate_angs             quanser_cartpole.py:74, quanser_ball_balancer.py:83 */
    VS_STEPCOUNT = 5,  /* i32 [ld]     Env.curr_step */
    VS_ERRFLAG = 6,    /* u8  [ld]     sticky: NaN seen in act/state (reference raises pyrado.ValueErr) */
    VS_RETURNS = 7,    /* f32 [ld]     undiscounted return of the running episode */
    VS_PARAMS = 8,     /* f32 [P][ld]  domain parameters, order = get_nominal_domain_param() */
    VS_CONSTS = 9,     /* f32 [K][ld]  derived constants (_calc_constants, bounds, c_max); layout in DESIGN.md */
    VS_EP_RETURNS = 10,/* f32 [ep_cap] episode log (vs_set_episode_log): completed-episode returns, append order (ring) */
    VS_EP_LENGTHS = 11,/* i32 [ep_cap] completed-episode lengths */
    VS_EP_ENVIDX = 12, /* i32 [ep_cap] env index of each completed episode */
    VS_EP_COUNT = 13,  /* u32 [1]      number of episodes appended since vs_clear_episodes */
    VS_TRAJ_REC = 14,  /* f32 [T][F * ld]: the records of a recording vs_step_random, one per env and step.
                        * record mode 1 (default), F = O + A + 1:
                        *   [obs BEFORE the step (O) | raw (unclipped) action of the policy (A) | reward]
                        * record mode 2 (vs_set_record_mode), F = O + A + 1 + S + A + H: everything rollout() keeps per step
                        * (P/sampling/rollout.py:237-258):
                        *   [... | state BEFORE the step (S) | applied action env.limit_act(act) (A) | hidden state BEFORE
                        *    the step (H; qcp: th_ddot)]
                        * Row t holds the F floats of every env split into planes of 4, 2 and 1 floats per env
                        * (F = 4 nq + 2 h2 + h1):
                        *   plane q < nq : f32 [ld][4] at float offset 4 ld q          <- record[4q .. 4q+3]
                        *   2-wide plane : f32 [ld][2] at 4 ld nq (if h2)              <- record[4nq], record[4nq+1]
                        *   1-wide plane : f32 [ld]    at (4 nq + 2 h2) ld (if h1)     <- record[F-1]
                        * so that a wavefront writes each plane with one dwordx4 / x2 / x1 store per lane, contiguously
                        * (vs_traj_layout reports nq, h2, h1). */
    VS_TRAJ_RESERVED_15 = 15,
    VS_TRAJ_RESERVED_16 = 16,
    VS_TRAJ_DONE = 17, /* u32 [ceil(T / 32)][ld]: done flag of recorded step t of env i = bit (t % 32) of word [t / 32][i].
                        * A lane keeps the running word in a register and a wave stores it once per 32 steps with one
                        * coalesced dword store (a byte per env and step was a 64-B partial-line store per wave and step) */
    VS_FAILED = 18,    /* u8  [ld]     Task.has_failed(state) of the last step   P/tasks/base.py:159-167 */
    VS_EPSTAT_COUNT = 19,  /* u32 [ld]  completed episodes per env since vs_clear_episodes */
    VS_EPSTAT_RETSUM = 20, /* f32 [ld]  sum of their undiscounted returns */
    VS_EPSTAT_LENSUM = 21, /* i32 [ld]  sum of their lengths */
    VS_JAC_STATE = 22,     /* f32 [S][S+A][ld]  d s'_j / d (s, a)_k of the last vs_step_jac */
    VS_JAC_REW = 23,       /* f32 [S+A][ld]     d r / d (s, a)_k */
    VS_JAC_OBS = 24,       /* f32 [O][S+A][ld]  d obs'_j / d (s, a)_k */
    VS_BUFFER_COUNT = 25
};

/* vs_task_cfg.flags */
#define VS_FLAG_SIMPLE_DYNAMICS 1 /* qcp / qbb ctor arg simple_dynamics=True */
#define VS_FLAG_LONG_POLE 2       /* qcp ctor arg long=True (changes the nominal pole only) */
#define VS_FLAG_ACT_NORM 4        /* ActNormWrapper fused into the step: incoming actions live in [-1, 1] and are mapped to
                                     lb + (a + 1) (ub - lb) / 2 before anything else sees them
                                     (P/environment_wrappers/action_normalization.py:66-75) */
#define VS_FLAG_FREEZE_DONE 8     /* vs_set_freeze_done: vs_step leaves lanes alone whose done flag is set (rollout() stops at
                                     done, P/sampling/rollout.py:185); off by default -- env.step() after done keeps stepping */
#define VS_FLAG_LEAN_STEP 16      /* vs_set_lean_step: vs_step returns what SimPyEnv.step returns -- (obs, rew, done),
                                     P/environments/pysim/base.py:217-241 -- and keeps neither the running return VS_RETURNS nor
                                     the VS_FAILED byte */

/* Task / ctor configuration. Zero-initialise and set `use_defaults = 1` to get the reference defaults
 * (_create_task of each env).  Q and R are diagonal (all reference defaults are). */
typedef struct vs_task_cfg {
    int32_t use_defaults; /* 1: ignore state_des/q_diag/r_diag below */
    int32_t flags;        /* VS_FLAG_* */
    int32_t wild_init;    /* qcp: 0 = 'True' (default), 1 = 'False', 2 = anything else   quanser_cartpole.py:552-560 */
    int32_t reserved;
    float state_des[8];   /* task_args['state_des'] */
    float q_diag[8];      /* diag(task_args['Q']) */
    float r_diag[2];      /* diag(task_args['R']) */
    float init_state[8];  /* pend: the fixed initial state of its SingularStateSpace (ctor arg init_state) */
} vs_task_cfg;

/* One randomised domain parameter: DomainParam.sample = distr.sample -> clamp(clip_lo, clip_up)
 * (P/domain_randomization/domain_parameter.py:104-203) */
#define VS_DP_NORMAL 0  /* NormalDomainParam(mean, std)       spread = std */
#define VS_DP_UNIFORM 1 /* UniformDomainParam(mean, halfspan) spread = halfspan */
#define VS_DP_BERNOULLI 2 /* BernoulliDomainParam(val_0, val_1, prob_1) (domain_parameter.py:248-311):
                           * mean = val_0, spread = val_1, aux = prob_1 */
/* MultivariateNormalDomainParam (domain_parameter.py:206-245) of dimension 1 is VS_DP_NORMAL with spread = sqrt(cov);
 * higher dimensions put a vector under ONE parameter name, which no pysim env accepts */
typedef struct vs_dp_spec {
    int32_t param_index; /* row in VS_PARAMS */
    int32_t kind;        /* VS_DP_* */
    float mean;
    float spread;
    float clip_lo;       /* -INFINITY for none */
    float clip_up;       /* +INFINITY for none */
    float aux;           /* VS_DP_BERNOULLI: prob_1 */
    int32_t roundint;    /* DomainParam(roundint=True): round half to even after clipping (domain_parameter.py:127-129) */
} vs_dp_spec;

/* A feed-forward network policy for vs_step_policy: FNN(input, output, hidden_sizes, hidden_nonlin, output_nonlin)
 * of P/policies/feed_back/fnn.py:43-160 (dropout 0).  Hidden layers are at most 64 units wide. */
#define VS_NL_NONE 0
#define VS_NL_TANH 1
#define VS_NL_RELU 2
#define VS_NL_SIGMOID 3
#define VS_FNN_MAX_HIDDEN 4
#define VS_FNN_MAX_WIDTH 64
typedef struct vs_fnn_desc {
    int32_t n_hidden;          /* 1 .. VS_FNN_MAX_HIDDEN hidden layers */
    int32_t hidden[4];         /* their sizes, 1 .. VS_FNN_MAX_WIDTH each */
    int32_t hidden_nonlin[4];  /* VS_NL_* per hidden layer */
    int32_t output_nonlin;     /* VS_NL_* of the output layer */
    int32_t feat;              /* 0: the input is the visible observation; 1: the fork's FNNPolicy.forward featurisation
                                  [o_0, sin o_1, cos o_1, o_2 ..] (fnn.py:219-222; one input more than observation rows) */
    int32_t n_obs;             /* number of observation rows the policy sees; 0 = all of them, in order */
    int32_t obs_idx[8];        /* ... and which (ObsPartialWrapper, P/environment_wrappers/observation_partial.py:36-75) */
    float noise_std[2];        /* exploration: + std * N(0, 1) per action dimension (NormalActNoiseExplStrat); 0 = none */
} vs_fnn_desc;

typedef struct vs_env* vs_handle;

/* ---- static information (no GPU needed) ---- */

/* widths of an env family: state, action, observation, #domain params, hidden state, init-space element, #constants */
int vs_env_dims(int env_type, int* S, int* A, int* O, int* P, int* H, int* I, int* K);
/* the `name` class attribute ("qq-su", ...), NULL for a bad type */
const char* vs_env_name(int env_type);
/* i-th domain parameter name in get_nominal_domain_param() order, NULL when out of range */
const char* vs_param_name(int env_type, int i);
/* nominal domain parameters (get_nominal_domain_param; qcp honours VS_FLAG_LONG_POLE); out has P floats */
int vs_nominal_params(int env_type, int flags, float* out);
/* plane decomposition of a VS_TRAJ_REC row in record mode 1 or 2 (see vs_buffer): F = 4 * nq + 2 * h2 + h1 */
int vs_traj_layout(int env_type, int record_mode, int* F, int* nq, int* h2, int* h1);
/* library / ABI version */
int vs_version(void);

/* ---- lifetime ---- */

/* Replaces the env constructor (SimPyEnv.__init__, P/environments/pysim/base.py:46-80) for n_envs instances:
 * nominal domain params, derived constants, spaces and task.  max_steps <= 0 means pyrado.inf.  cfg may be NULL. */
int vs_create(int env_type, int64_t n_envs, double dt, int64_t max_steps, int device_id, const vs_task_cfg* cfg,
              vs_handle* out);
int vs_destroy(vs_handle h);
/* `env.max_steps = n` / `env.dt = h` (P/environments/base.py:73-104): plain attribute updates in the reference -- state,
 * step counters and running episodes are left alone; the new values apply from the next step.  max_steps <= 0: inf. */
int vs_set_max_steps(vs_handle h, int64_t max_steps);
int vs_set_dt(vs_handle h, double dt);
/* Streams.  A handle launches on its own stream, created as a blocking stream (hipStreamDefault): it is implicitly
 * ordered with the legacy default stream (torch's default "current stream"), so default-stream work may read the
 * handle's buffers (vs_get) after a launch without further synchronisation.  A caller that runs on another,
 * non-blocking stream (a torch side stream, a hipGraph capture stream) passes that hipStream_t here and the handle
 * launches on it; NULL restores the own stream.  For a loop that alternates the caller's default-stream work with vs_step,
 * pass hipStreamLegacy ((hipStream_t)1): on the very same stream there is no hand-over to pay for (the implicit
 * synchronisation between the legacy stream and a blocking stream costs ~25 us per step). */
int vs_set_stream(vs_handle h, void* hip_stream);
int vs_sync(vs_handle h);
int64_t vs_n_envs(vs_handle h);
int64_t vs_ld(vs_handle h);
const char* vs_last_error(vs_handle h); /* h may be NULL: last error of a failed vs_create on this thread */

/* ---- domain parameters: the `domain_param` setter (P/environments/pysim/base.py:112-124) ---- */

/* params_soa: f32 [P][pitch]; pitch = n_envs for host memory. mask (u8 [n_envs], may be NULL = all) selects the envs.
 * Recomputes _calc_constants, the spaces' bounds and the reward scale c_max for the selected envs. */
int vs_set_params(vs_handle h, const float* params_soa, int64_t pitch, const uint8_t* mask);
/* the same parameter vector (P floats, host) for every env; enables the broadcast-constant kernels */
int vs_set_params_uniform(vs_handle h, const float* params);
/* DomainRandomizer.randomize + get_params on device (P/domain_randomization/domain_randomizer.py:123-227):
 * Normal/Uniform draws (Philox4x32-10 keyed by seed), clipped; parameters without a spec keep their value. */
int vs_sample_params(vs_handle h, const vs_dp_spec* specs, int n_specs, uint64_t seed, const uint8_t* mask);
/* DomainRandWrapperBuffer (P/environment_wrappers/domain_randomization.py:151-261): a buffer of n_sets domain-parameter
 * sets, f32 [P][n_sets] (host memory); at each of its resets an env takes the next set (selection 0 = cyclic: lane i starts
 * at set (global index i) mod n_sets and advances by one per reset) or a random one (selection 1, Philox).
 * n_sets = 0 removes the buffer.  Mutually exclusive with vs_set_randomizer. */
int vs_set_param_buffer(vs_handle h, const float* params_soa, int n_sets, int selection);
/* toggle VS_FLAG_ACT_NORM after creation */
int vs_set_act_norm(vs_handle h, int on);
/* GaussianActNoiseWrapper (P/environment_wrappers/action_noise.py:38-79: act + randn * std + mean) and ActDelayWrapper
 * (P/environment_wrappers/action_delay.py:37-112: a queue of `delay` zero actions at reset, push the commanded action,
 * pop the oldest) fused into the step, applied after ActNormWrapper's de-normalisation and before the env's reward and
 * action clipping -- exactly where the wrapped env's step() would see them.
 *   delay              0 .. VS_MAX_ACT_DELAY time steps, the same for every env
 *   noise_mean/std     A floats each (host) or NULL for none
 *   noise_normed       the noise wrapper sits OUTSIDE ActNormWrapper: its draw is scaled by (ub - lb) / 2 of the env
 *   noise_after_delay  the noise wrapper sits INSIDE ActDelayWrapper (noise is added to the popped action)
 *   seed               Philox key of the noise; a draw is a pure function of (seed, global env index, episode, step)
 * vs_step_random draws the policy's action first and then runs it through this pipeline; the record holds the
 * policy's action.  delay = 0 and NULL noise removes the stage. */
#define VS_MAX_ACT_DELAY 64
int vs_set_act_pipeline(vs_handle h, int delay, const float* noise_mean, const float* noise_std, int noise_normed,
                        int noise_after_delay, uint64_t seed);
/* ObsNormWrapper (P/environment_wrappers/observation_normalization.py:41-126: (obs - lb) / (ub - lb) * 2 - 1) and
 * GaussianObsNoiseWrapper (P/environment_wrappers/observation_noise.py:38-73: obs + randn * std + mean), in any stacking
 * order, compose to   obs' = obs * scale + shift + noise_std * z,  z ~ N(0, 1) i.i.d.   (O floats each, host, NULL =
 * identity / none).  VS_OBS, the recorded observations and the observation vs_reset produces are the wrapped ones. */
int vs_set_obs_pipeline(vs_handle h, const float* scale, const float* shift, const float* noise_std, uint64_t seed);
/* DomainRandWrapperLive (P/environment_wrappers/domain_randomization.py:135-148): remember specs and redraw the
 * parameters of an env at each of its resets (vs_reset without explicit params, and auto-reset). n_specs = 0 disables. */
int vs_set_randomizer(vs_handle h, const vs_dp_spec* specs, int n_specs);

/* ---- reset: SimPyEnv.reset (P/environments/pysim/base.py:166-203) ---- */

/* init_state: NULL -> sample the env's init space on device (init_space.sample_uniform, P/spaces/box.py:169-178,
 * polar.py:108-113, compound.py:84-87); else f32 [I or S][pitch] (init_is_full_state selects which, base.py:184-193).
 * mask as above.  seed keys the Philox stream used for sampling (and for live domain randomisation). */
int vs_reset(vs_handle h, const float* init_state, int64_t pitch, int init_is_full_state, const uint8_t* mask,
             uint64_t seed);
/* Global index of lane 0 (default 0).  Every random stream of lane i is keyed by (first_global_index + i): a set of
 * envs gives the same trajectories whether it lives in one handle, is cut into batches, or is sharded over GPUs
 * (rank r of a node passes r * n_envs). */
int vs_set_index_offset(vs_handle h, uint32_t first_global_index);
/* when on, a lane whose episode ended is reset inside the same step kernel (fresh init state, redrawn params when a
 * randomizer is set); VS_OBS then holds the first observation of the new episode, VS_REW/VS_DONE the finished step */
int vs_set_auto_reset(vs_handle h, int on, uint64_t seed);

/* ---- step: SimPyEnv.step (P/environments/pysim/base.py:217-241), fused in one kernel ---- */

/* actions: device f32, element (env i, dim j) at actions[i * env_stride + j * dim_stride]
 * ([A][ld] SoA: env_stride 1, dim_stride ld;  [N][A] row-major policy output: env_stride A, dim_stride 1) */
int vs_step(vs_handle h, const float* actions, int64_t env_stride, int64_t dim_stride);
/* vs_step that also records the step (policy in the loop: rollout() with the caller's policy, rollout.py:185-258): the
 * observation the policy saw (VS_OBS as the previous step / the reset left it), its action, the reward and the done bit --
 * in record mode 2 also the state and hidden state before the step and env.limit_act(act) -- go into row `row` of the
 * VS_TRAJ_* buffers (same layout as the records of vs_step_random).  row < 0: the row comes from a device-side counter of
 * the handle (vs_set_record_row), which a one-thread kernel behind the step advances -- nothing host-side enters the
 * launch, so a captured hipGraph of (policy, vs_step_record) pairs replays correctly.  Rows beyond the capacity are skipped. */
int vs_step_record(vs_handle h, const float* actions, int64_t env_stride, int64_t dim_stride, int row);
int vs_set_record_row(vs_handle h, int row);
/* vs_step plus the step Jacobians d(s', r, obs') / d(s, a) (forward-mode differentiation of the same step code): what the
 * fork computes with torch autograd for its SAC-with-gradients (P/sampling/rollout.py:836-837 around
 * quanser_cartpole.py:233-431), for every family.  The raw action is the differentiation variable (a clipped or dead-zoned
 * action has zero gradient); hidden state is held constant; auto-reset must be off.  Values are bit-identical to vs_step. */
int vs_step_jac(vs_handle h, const float* actions, int64_t env_stride, int64_t dim_stride);
/* rollout() with DummyPolicy (P/sampling/rollout.py:185-239, P/policies/feed_forward/dummy.py:77-84):
 * k_steps env steps in ONE launch with on-device uniform actions in act_space, state kept in registers.
 * record != 0 streams obs/act/rew/done of every step into the VS_TRAJ_* buffers (k_steps <= vs_traj_capacity). */
int vs_step_random(vs_handle h, uint64_t seed, int k_steps, int record);
/* rollout() with a feed-forward network policy evaluated INSIDE the fused kernel (rollout.py:185-258 with act = policy(obs),
 * rollout.py:203-219): vs_set_policy_fnn hands over the network -- `params` is the policy's flat parameter vector in torch
 * order (per layer: weight [out][in] row-major, then bias [out]; hidden layers first, output layer last =
 * parameters_to_vector(FNN.parameters()), fnn.py:104-112), host or device memory, copied -- and vs_step_policy runs k_steps
 * env steps per launch: observation -> network -> (+ exploration noise) -> SimPyEnv.step -> record, exactly as
 * vs_step_random does with its uniform policy (same records, auto-reset and freeze semantics).  The noise stream is
 * Philox(noise_seed; global env index, episode index, step).  desc == NULL removes the network.
 * Not available with a wrapper pipeline on the handle or for the discrete-action family (VS_ERR_STATE / VS_ERR_ARG). */
int vs_set_policy_fnn(vs_handle h, const vs_fnn_desc* desc, const float* params, int64_t n_params);
int vs_step_policy(vs_handle h, int k_steps, int record, uint64_t noise_seed);
/* the shape vs_step_policy evaluates the network in: -1 automatic (default), 0 the network of 64 envs spread over the 8 waves
 * of a 64-env workgroup (vector ALU, lane = hidden unit), 1 the same in 256-env workgroups, 2 256-env workgroups with the
 * hidden layers on the matrix cores (v_mfma_f32_32x32x2_f32: fp32 in, fp32 out; one and two hidden layers -- deeper networks
 * run shape 0 whatever is asked).  No counterpart in the reference (which evaluates the torch module, P/sampling/rollout.py:203-219);
 * the shapes differ in the summation order of a layer only. */
int vs_set_policy_shape(vs_handle h, int shape);
/* The recorded steps of lanes 0 .. n_lanes - 1 (rows 0 .. of VS_TRAJ_REC, vs_set_record_mode's layout) as ROLLOUTS in one row-major
 * matrix rows[total + n_lanes][F] (F = vs_traj_layout's record width, device memory): rollout j = steps 0 .. lengths[j] - 1 of
 * lane j, the rollouts one after the other, starts[j] = lengths[0] + .. + lengths[j - 1] (both int64, device memory).
 *   rows[starts[j] + j + t]          the record of step t: [obs (O) | act (A) | rew | state (S) | act_app (A) | hidden (H)] (the last
 *                                    three in record mode 2 only), t < lengths[j];
 *   rows[starts[j] + j + lengths[j]] the entry behind the last step: final observation / state / hidden state from VS_OBS / VS_STATE /
 *                                    VS_HIDDEN of the lane (frozen at its done), the per-step fields 0.
 * Every field of a rollout is therefore a strided view of lengths[j] (per-step fields) or lengths[j] + 1 rows (observations, states,
 * hidden states: the value before every step and the final one).  Runs on the handle's stream.
 * Replaces: the histories rollout() returns per env -- P/sampling/rollout.py:305-325 (obs_hist / act_hist / rew_hist / state_hist /
 * act_app_hist / th_ddot_hist -> StepSequence) -- and their concatenation over rollouts, P/sampling/step_sequence.py:777-825. */
/* Where the rollouts in the records end: lengths[j] = 1 + the first of rows 0 .. t_steps - 1 of VS_TRAJ_DONE whose done bit is set
 * for lane j (t_steps if none is), done_last[j] = whether one is -- the step at which rollout()'s loop stops
 * (`while not done and env.curr_step < env.max_steps`, P/sampling/rollout.py:185) and StepSequence.done[-1].  Device memory. */
int vs_rollout_lengths(vs_handle h, int n_lanes, int t_steps, int64_t* lengths, uint8_t* done_last);
int vs_pack_traj(vs_handle h, int n_lanes, int t_steps, const int64_t* lengths, const int64_t* starts, float* rows);
/* The action stream of vs_step_random is Philox(seed; global env index, absolute step index); the handle counts the
 * steps it has taken.  vs_seek_random repositions that counter (0 = start of a fresh batch of rollouts). */
int vs_seek_random(vs_handle h, uint64_t step_index);
/* rows of the VS_TRAJ_* buffers (grow-only; a larger capacity replaces and frees the previous buffers) */
int vs_set_traj_capacity(vs_handle h, int t_max);
/* what a recording vs_step_random writes per step: 1 = [obs | act | rew] (default), 2 = + [state | act_app | hidden], the
 * fields rollout() returns in its StepSequence (rollout.py:305-325).  Changing the mode drops the record buffers: set
 * the capacity again afterwards. */
int vs_set_record_mode(vs_handle h, int mode);
int vs_record_mode(vs_handle h);
/* rollout() stops stepping an env at done (rollout.py:185).  With freeze on (and auto-reset off) vs_step skips the lanes
 * whose VS_DONE flag is set: state, observation, step counter and flags stay, VS_REW reads 0, and whatever action such a
 * lane is fed cannot raise its NaN flag.  Off (default): env.step() after done keeps stepping, as in the reference. */
int vs_set_freeze_done(vs_handle h, int on);
/* SimPyEnv.step returns (obs, rew, done, info) and nothing else (P/environments/pysim/base.py:217-241).  vs_step by default also
 * keeps a running undiscounted return per env (VS_RETURNS, what the auto-reset books into VS_EPSTAT_RETSUM when an episode
 * ends) and the VS_FAILED byte: 9 bytes per env step on top of the 117 algorithmic ones of SURVEY.md 8(d).  With lean on,
 * vs_step reads and writes exactly that model: VS_RETURNS / VS_FAILED are left untouched (stale) and episodes that end under
 * auto-reset count into VS_EPSTAT_COUNT / VS_EPSTAT_LENSUM with a return of 0.  vs_step_random / vs_step_policy are not
 * affected (their returns live in registers).  Off by default. */
int vs_set_lean_step(vs_handle h, int on);
/* which kernel the next vs_step_random launches for this handle's configuration: 0 = k_rollout (one wave per 64 envs),
 * 1 = k_rollout_ws in 256-env workgroups, 2 = k_rollout_ws in 64-env workgroups (a physics wave and a reward/record wave
 * per 64 envs; chosen while the plain kernel would leave the SIMDs with a single wave, the small workgroups while the
 * batch cannot give every compute unit a large one).  Results are bit-identical. */
int vs_rollout_variant(vs_handle h);
/* pin the choice: -1 automatic (default), 0 k_rollout, 1 / 2 k_rollout_ws where the configuration allows it (no wrapper
 * pipeline, no state-and-time dependent final reward; a live randomizer / parameter buffer only for the families with
 * fixed action bounds and an unscaled reward: qq-*, qcp-su), else k_rollout */
int vs_set_rollout_variant(vs_handle h, int variant);
/* first row of the VS_TRAJ_* buffers written by the next recording vs_step_random (default 0): consecutive launches can
 * fill one long trajectory buffer, t0 + k_steps <= capacity */
int vs_set_traj_offset(vs_handle h, int t0);
/* Episode bookkeeping.  Always on: per-env accumulators VS_EPSTAT_* (plain per-lane adds, no atomics) -- what the
 * RCCL return gather reads.  Opt-in (vs_set_episode_log): every finished episode is also appended as (return, length,
 * env index) to the VS_EP_* ring, compacted with a wavefront ballot and one atomic per wave; that atomic is a shared
 * counter (about 90 appends/us chip-wide), so leave the log off on throughput runs with short episodes. */
int vs_set_episode_log(vs_handle h, int on);
int vs_clear_episodes(vs_handle h);

/* ---- mixed batches (BASELINE config 5): several env families stepped by ONE launch ----
 * A mixed handle groups up to 5 ordinary handles on one device (lanes sorted by type: each handle is one contiguous
 * segment).  A workgroup belongs to one segment, so the env-type dispatch is uniform per workgroup / wavefront.
 * Parameters, resets and data access go through the member handles; the mixed handle only fuses the launches.  All
 * members share the stream of the first one and must agree on auto-reset. */
typedef struct vs_mixed* vs_mixed_handle;
int vs_mixed_create(const vs_handle* handles, int n, vs_mixed_handle* out);
int vs_mixed_destroy(vs_mixed_handle m);
const char* vs_mixed_last_error(vs_mixed_handle m);
/* vs_step_random / vs_step for every member in one kernel; actions[q] etc. are the arguments of vs_step for member q */
int vs_mixed_step_random(vs_mixed_handle m, uint64_t seed, int k_steps, int record);
int vs_mixed_step(vs_mixed_handle m, const float* const* actions, const int64_t* env_strides, const int64_t* dim_strides);
int vs_mixed_time_random(vs_mixed_handle m, uint64_t seed, int k_steps, int record, int iters, float* avg_ms);

/* ---- data access ---- */

void* vs_get(vs_handle h, int which);                       /* device pointer, NULL on error */
int vs_copy_to_host(vs_handle h, int which, void* dst);     /* whole buffer, pitch ld (synchronises the stream) */
int vs_copy_from_host(vs_handle h, int which, const void* src); /* VS_STATE / VS_HIDDEN / VS_STEPCOUNT: `state` setter */
/* number of lanes with the sticky error flag set (synchronises); the host shim raises pyrado.ValueErr when > 0 */
int64_t vs_error_count(vs_handle h);

/* ---- measurement ---- */

/* average device time [ms] of the step kernel over `iters` launches, measured with hipEvents on the handle's stream
 * (mode 0: vs_step with the given device actions, mode 1: vs_step_random(k_steps, record); recording launches rotate
 * through the record buffer in k_steps-row slots, so a capacity beyond the 256 MiB Infinity Cache makes it an HBM stream) */
int vs_time_step_kernel(vs_handle h, int mode, const float* actions, int64_t env_stride, int64_t dim_stride,
                        int k_steps, int record, int iters, float* avg_ms);
/* HIP-event stopwatch on the handle's stream: vs_timer_start records an event where the stream stands, vs_timer_stop
 * records a second one, waits for it and returns the device time between the two [ms] -- the time of exactly the launches
 * issued in between (bench.py brackets its timed region with it: kernel time and wall time of the same launches) */
int vs_timer_start(vs_handle h);
int vs_timer_stop(vs_handle h, float* ms);
/* streaming copy kernel (float4, 4 independent 16-B accesses per thread, non-temporal, one-shot grid, own stream) over
 * `bytes` of device memory: achieved GB/s read + write (in-repo HBM reference point) */
int vs_membw_probe(int device_id, int64_t bytes, int iters, float* gbps);
/* the same for a pure write stream (float4 stores): the ceiling of the record stream of vs_step_random */
int vs_memwrite_probe(int device_id, int64_t bytes, int iters, float* gbps);

#ifdef __cplusplus
}
#endif
#endif /* VECSIM_H */
