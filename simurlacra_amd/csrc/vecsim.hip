// vecsim.hip -- libvecsim: the C-ABI of include/vecsim.h (host side).  The kernels live in vecsim_kernels.h and are
// compiled per env family (vecsim_family.hip) and for the mixed batches (vecsim_mixed.hip); see build.py.
#include "vecsim_kernels.h"

namespace vs {

__global__ void k_bump_row(int* row) { *row += 1; }

// vs_rollout_lengths: per lane, the first recorded step whose done bit is set (words [t / 32][ld], bit t % 32)
__global__ __launch_bounds__(256) void k_rollout_lengths(const uint32_t* __restrict__ words, size_t ld, int n, int t_steps,
                                                         long long* __restrict__ lengths, uint8_t* __restrict__ done_last) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int nw = (t_steps + 31) / 32;
    int first = -1;
    for (int w = 0; w < nw && first < 0; ++w) {
        uint32_t bits = words[(size_t)w * ld + i];
        if (w == nw - 1 && (t_steps & 31)) bits &= (1u << (t_steps & 31)) - 1u;  // rows beyond t_steps are not part of it
        if (bits) first = w * 32 + (__ffs((int)bits) - 1);
    }
    lengths[i] = first < 0 ? (long long)t_steps : (long long)first + 1;
    done_last[i] = first >= 0;
}

__global__ void k_count_err(const uint8_t* err, int n, unsigned long long* out) {
    int i = blockIdx.x * BLOCK + threadIdx.x;
    bool e = i < n && err[i] != 0;
    unsigned long long m = __builtin_amdgcn_ballot_w64(e);
    if (__lane_id() == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}

// HBM reference points next to the roofline: a streaming float4 copy and a pure float4 write stream.
// One-shot grids (a block owns PROBE_V * 256 consecutive float4: 16 KiB), PROBE_V independent 16-B accesses per thread in
// flight, non-temporal (the data is touched once).
constexpr int PROBE_V = 4;
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_fill4(v4f* __restrict__ dst, size_t n4, float v) {
    size_t base = (size_t)blockIdx.x * (256 * PROBE_V) + threadIdx.x;
    const v4f x = {v, v + 1.f, v + 2.f, v + 3.f};
#pragma unroll
    for (int k = 0; k < PROBE_V; ++k) {
        size_t i = base + (size_t)k * 256;
        if (i < n4) __builtin_nontemporal_store(x, &dst[i]);
    }
}

__global__ __launch_bounds__(256) void k_copy4(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n4) {
    size_t base = (size_t)blockIdx.x * (256 * PROBE_V) + threadIdx.x;
    v4f x[PROBE_V];
#pragma unroll
    for (int k = 0; k < PROBE_V; ++k) {
        size_t i = base + (size_t)k * 256;
        if (i < n4) x[k] = __builtin_nontemporal_load(&src[i]);
    }
#pragma unroll
    for (int k = 0; k < PROBE_V; ++k) {
        size_t i = base + (size_t)k * 256;
        if (i < n4) __builtin_nontemporal_store(x[k], &dst[i]);
    }
}

}  // namespace vs

// ====================================================================================================================
// host side
// ====================================================================================================================
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

struct vs_mixed {
    int n = 0;
    vs_handle sub[MAX_SEG]{};
    Segs host{};
    Segs* dev = nullptr;
    int total_blocks = 0;
    std::string err;
};

// some member redraws domain parameters at a reset inside the launch (live randomizer / parameter buffer)
static bool mixed_redraws(const vs_mixed* m) {
    for (int q = 0; q < m->n; ++q)
        if (m->sub[q]->d.dr_n > 0 || m->sub[q]->d.pbuf_n > 0) return true;
    return false;
}

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

int vs_version(void) { return 304; }

static int record_width(int t, int mode) {
    const EnvInfo& e = ENV_INFO[t];
    return mode == 2 ? e.O + e.A + 1 + e.S + e.A + e.H : e.O + e.A + 1;
}

int vs_traj_layout(int t, int mode, int* F, int* nq, int* h2, int* h1) {
    if (t < 0 || t >= VS_ENV_COUNT || mode < 1 || mode > 2) return VS_ERR_ARG;
    const int f = record_width(t, mode);
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
    CK(dalloc(h, &d.rec_row, (size_t)1));
#ifdef VS_WS_STAMP
    CK(dalloc(h, &d.dbg, (size_t)(ld / 64) * 12));
#endif

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
    if (h->fnn.w) (void)hipFree((void*)h->fnn.w);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
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
    DISPATCH_ENV(h->type, Launch<E>::set_params(h, src, sp, 0, m));
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
    DISPATCH_ENV(h->type, Launch<E>::set_params(h, (const float*)h->stage, 0L, 1, (const uint8_t*)nullptr));
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
    DISPATCH_ENV(h->type, Launch<E>::sample_params(h, seed, m));
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
    DISPATCH_ENV(h->type, Launch<E>::observe(h));
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
    DISPATCH_ENV(h->type, Launch<E>::reset(h, src, sp, full, m, seed));
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
    DISPATCH_ENV(h->type, Launch<E>::step(h, actions, (long)env_stride, (long)dim_stride));
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_step_record(vs_handle h, const float* actions, int64_t env_stride, int64_t dim_stride, int row) {
    if (!h || !actions) return fail(h, VS_ERR_ARG, "vs_step_record: NULL argument");
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &st) != hipSuccess) (void)hipGetLastError();
    if (st == hipStreamCaptureStatusNone) {
        if (!is_device_ptr(actions)) return fail(h, VS_ERR_ARG, "vs_step_record: actions must be device memory");
        HIPCHK(h, hipSetDevice(h->device));
    }
    if (h->traj_cap <= 0) return fail(h, VS_ERR_STATE, "vs_step_record: set the record capacity first (vs_set_traj_capacity)");
    if (row >= h->traj_cap) return fail(h, VS_ERR_STATE, "vs_step_record: row exceeds vs_set_traj_capacity");
    DISPATCH_ENV(h->type, Launch<E>::step(h, actions, (long)env_stride, (long)dim_stride, h->record_mode, row < 0 ? -1 : row));
    if (row < 0) hipLaunchKernelGGL(k_bump_row, dim3(1), dim3(1), 0, h->stream, h->d.rec_row);
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_set_record_row(vs_handle h, int row) {
    if (!h || row < 0) return fail(h, VS_ERR_ARG, "vs_set_record_row: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemcpyAsync(h->d.rec_row, &row, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // `row` lives on this stack frame
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
    DISPATCH_ENV(h->type, Launch<E>::jac(h, actions, (long)env_stride, (long)dim_stride));
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

// the record buffers of a superseded capacity / mode are released at once (a sampler that resizes per call must not grow)
static int free_traj(vs_handle h) {
    Dev& d = h->d;
    void* old[2] = {d.traj_rec, d.traj_done};
    for (void* q : old) {
        if (!q) continue;
        for (size_t k = 0; k < h->allocs.size(); ++k)
            if (h->allocs[k] == q) { h->allocs.erase(h->allocs.begin() + (long)k); break; }
        HIPCHK(h, hipFree(q));
    }
    d.traj_rec = nullptr;
    d.traj_done = nullptr;
    h->traj_cap = 0;
    d.traj_rows = 0;
    return VS_OK;
}

int vs_set_traj_capacity(vs_handle h, int t_max) {
    if (!h || t_max < 0) return fail(h, VS_ERR_ARG, "vs_set_traj_capacity: bad argument");
    if (t_max <= h->traj_cap) return VS_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    Dev& d = h->d;
    size_t ld = d.ld;
    int rc;
    if ((rc = free_traj(h))) return rc;
    if ((rc = dalloc(h, &d.traj_rec, (size_t)t_max * record_width(h->type, h->record_mode) * ld))) return rc;
    if ((rc = dalloc(h, &d.traj_done, (size_t)((t_max + 31) / 32) * ld))) return rc;
    h->traj_cap = t_max;
    d.traj_rows = t_max;
    return VS_OK;
}

int vs_set_record_mode(vs_handle h, int mode) {
    if (!h || mode < 1 || mode > 2) return fail(h, VS_ERR_ARG, "vs_set_record_mode: 1 (obs | act | rew) or 2 (+ state | act_app | hidden)");
    if (mode == h->record_mode) return VS_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int rc = free_traj(h);  // the row width changes: the capacity has to be set again
    if (rc) return rc;
    h->record_mode = mode;
    h->d.traj_t0 = 0;
    return VS_OK;
}

int vs_record_mode(vs_handle h) { return h ? h->record_mode : VS_ERR_ARG; }

int vs_set_freeze_done(vs_handle h, int on) {
    if (!h) return VS_ERR_ARG;
    if (on) h->task.flags |= VS_FLAG_FREEZE_DONE;
    else h->task.flags &= ~VS_FLAG_FREEZE_DONE;
    return VS_OK;
}

int vs_set_lean_step(vs_handle h, int on) {
    if (!h) return VS_ERR_ARG;
    if (on) h->task.flags |= VS_FLAG_LEAN_STEP;
    else h->task.flags &= ~VS_FLAG_LEAN_STEP;
    return VS_OK;
}

int vs_set_rollout_variant(vs_handle h, int variant) {
    if (!h || variant < -1 || variant > 4) return fail(h, VS_ERR_ARG, "vs_set_rollout_variant: -1 (automatic) or 0 .. 4");
    h->rollout_variant = variant;
    return VS_OK;
}

int vs_set_policy_shape(vs_handle h, int shape) {
    if (!h || shape < -1 || shape > 2) return fail(h, VS_ERR_ARG, "vs_set_policy_shape: -1 (automatic) or 0 .. 2");
    h->policy_shape = shape;
    return VS_OK;
}

int vs_rollout_variant(vs_handle h) {
    if (!h) return VS_ERR_ARG;
    int var = 0;
    DISPATCH_ENV(h->type, var = Launch<E>::variant(h));
    return var;
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
    DISPATCH_ENV(h->type, Launch<E>::rollout(h, k_steps, seed, ep, record ? h->record_mode : 0));
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_set_policy_fnn(vs_handle h, const vs_fnn_desc* desc, const float* params, int64_t n_params) {
    if (!h) return VS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->fnn.w) { HIPCHK(h, hipFree((void*)h->fnn.w)); }
    h->fnn = Fnn{};
    if (!desc) return VS_OK;
    const EnvInfo& ei = ENV_INFO[h->type];
    if (h->type == VS_ENV_BOB_D) return fail(h, VS_ERR_ARG, "vs_set_policy_fnn: the discrete-action family takes no network policy");
    if (!params || desc->n_hidden < 1 || desc->n_hidden > FNN_MAXH) return fail(h, VS_ERR_ARG, "vs_set_policy_fnn: 1 .. 4 hidden layers and a parameter vector");
    Fnn f{};
    f.n_hidden = desc->n_hidden;
    f.n_vis = desc->n_obs > 0 ? desc->n_obs : ei.O;
    if (f.n_vis > ei.O) return fail(h, VS_ERR_ARG, "vs_set_policy_fnn: more visible observation rows than the env has");
    f.ident = 1;
    for (int k = 0; k < f.n_vis; ++k) {
        f.obs_idx[k] = desc->n_obs > 0 ? desc->obs_idx[k] : k;
        if (f.obs_idx[k] < 0 || f.obs_idx[k] >= ei.O) return fail(h, VS_ERR_ARG, "vs_set_policy_fnn: obs_idx out of range");
        if (f.obs_idx[k] != k) f.ident = 0;
    }
    if (f.n_vis != ei.O) f.ident = 0;
    f.feat = desc->feat != 0;
    if (f.feat && f.n_vis < 2) return fail(h, VS_ERR_ARG, "vs_set_policy_fnn: the sin / cos featurisation needs two observation rows");
    f.in_dim = f.n_vis + f.feat;
    f.out_dim = ei.A;
    f.out_nonlin = desc->output_nonlin;
    int64_t need = 0;
    int off = 0, last = f.in_dim;
    for (int l = 0; l < f.n_hidden; ++l) {
        f.hidden[l] = desc->hidden[l];
        f.hid_nonlin[l] = desc->hidden_nonlin[l];
        if (f.hidden[l] < 1 || f.hidden[l] > FNN_W) return fail(h, VS_ERR_ARG, "vs_set_policy_fnn: hidden layers are 1 .. 64 units wide");
        if (f.hid_nonlin[l] < 0 || f.hid_nonlin[l] > FNN_SIGMOID) return fail(h, VS_ERR_ARG, "vs_set_policy_fnn: unknown nonlinearity");
        need += (int64_t)f.hidden[l] * last + f.hidden[l];
        f.off_w[l] = off;
        off += (l == 0 ? last : FNN_W) * FNN_W;  // layers behind the first read all 64 (zero-padded) input rows
        f.off_b[l] = off;
        off += FNN_W;
        last = f.hidden[l];
    }
    if (f.out_nonlin < 0 || f.out_nonlin > FNN_SIGMOID) return fail(h, VS_ERR_ARG, "vs_set_policy_fnn: unknown nonlinearity");
    need += (int64_t)ei.A * last + ei.A;
    f.off_w[f.n_hidden] = off;
    off += ei.A * FNN_W;
    f.off_b[f.n_hidden] = off;
    off += 8;
    if (n_params != need) return fail(h, VS_ERR_ARG, "vs_set_policy_fnn: parameter count does not match the layer sizes");
    for (int j = 0; j < ei.A; ++j) {
        f.noise_std[j] = desc->noise_std[j];
        if (!(f.noise_std[j] >= 0.f)) return fail(h, VS_ERR_ARG, "vs_set_policy_fnn: noise_std must be >= 0");
        if (f.noise_std[j] > 0.f) f.noisy = 1;
    }
    // torch layout -> transposed, zero-padded rows: unit j of layer l reads Wt_l[k][j], contiguous over j (scalar-load friendly)
    std::vector<float> src((size_t)need), pk((size_t)off, 0.f);
    HIPCHK(h, hipMemcpy(src.data(), params, (size_t)need * sizeof(float), is_device_ptr(params) ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
    size_t q = 0;
    last = f.in_dim;
    for (int l = 0; l < f.n_hidden; ++l) {
        for (int j = 0; j < f.hidden[l]; ++j)
            for (int k = 0; k < last; ++k) pk[(size_t)f.off_w[l] + (size_t)k * FNN_W + j] = src[q++];
        for (int j = 0; j < f.hidden[l]; ++j) pk[(size_t)f.off_b[l] + j] = src[q++];
        last = f.hidden[l];
    }
    for (int j = 0; j < ei.A; ++j)
        for (int k = 0; k < last; ++k) pk[(size_t)f.off_w[f.n_hidden] + (size_t)j * FNN_W + k] = src[q++];
    for (int j = 0; j < ei.A; ++j) pk[(size_t)f.off_b[f.n_hidden] + j] = src[q++];
    float* dw = nullptr;
    HIPCHK(h, hipMalloc((void**)&dw, pk.size() * sizeof(float)));
    hipError_t e = hipMemcpy(dw, pk.data(), pk.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(dw); return fail(h, VS_ERR_HIP, "vs_set_policy_fnn: upload", e); }
    f.w = dw;
    h->fnn = f;
    return VS_OK;
}

int vs_step_policy(vs_handle h, int k_steps, int record, uint64_t noise_seed) {
    if (!h || k_steps < 1) return fail(h, VS_ERR_ARG, "vs_step_policy: bad argument");
    if (!h->fnn.w) return fail(h, VS_ERR_STATE, "vs_step_policy: no network set (vs_set_policy_fnn)");
    if (h->d.pipe.act_on || h->d.pipe.obs_on) return fail(h, VS_ERR_STATE, "vs_step_policy: not available with a wrapper pipeline on the handle");
    if (record && h->d.traj_t0 + k_steps > h->traj_cap) return fail(h, VS_ERR_STATE, "vs_step_policy: traj offset + k_steps exceeds vs_set_traj_capacity");
    HIPCHK(h, hipSetDevice(h->device));
    DISPATCH_ENV(h->type, Launch<E>::rollout_fnn(h, k_steps, record ? h->record_mode : 0, noise_seed));
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_rollout_lengths(vs_handle h, int n_lanes, int t_steps, int64_t* lengths, uint8_t* done_last) {
    if (!h || n_lanes < 1 || n_lanes > h->d.n || t_steps < 1 || !lengths || !done_last)
        return fail(h, VS_ERR_ARG, "vs_rollout_lengths: bad argument");
    if (!h->d.traj_done || t_steps > h->traj_cap) return fail(h, VS_ERR_STATE, "vs_rollout_lengths: more steps than vs_set_traj_capacity holds");
    HIPCHK(h, hipSetDevice(h->device));
    hipLaunchKernelGGL(k_rollout_lengths, dim3((unsigned)((n_lanes + 255) / 256)), dim3(256), 0, h->stream, (const uint32_t*)h->d.traj_done,
                       (size_t)h->d.ld, n_lanes, t_steps, (long long*)lengths, done_last);
    HIPCHK(h, hipGetLastError());
    return VS_OK;
}

int vs_pack_traj(vs_handle h, int n_lanes, int t_steps, const int64_t* lengths, const int64_t* starts, float* rows) {
    if (!h || n_lanes < 1 || n_lanes > h->d.n || t_steps < 1 || !lengths || !starts || !rows)
        return fail(h, VS_ERR_ARG, "vs_pack_traj: bad argument");
    if (!h->d.traj_rec || t_steps > h->traj_cap) return fail(h, VS_ERR_STATE, "vs_pack_traj: more steps than vs_set_traj_capacity holds");
    if (((uintptr_t)rows & 3u) != 0) return fail(h, VS_ERR_ARG, "vs_pack_traj: the destination must be 4-byte aligned");
    HIPCHK(h, hipSetDevice(h->device));
    DISPATCH_ENV(h->type, Launch<E>::pack_traj(h, n_lanes, t_steps, (const long long*)lengths, (const long long*)starts, rows));
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
        if (m->sub[q]->record_mode != m->sub[0]->record_mode) { m->err = "vs_mixed_step_random: segments differ in record mode"; return VS_ERR_STATE; }
    }
    if (hipSetDevice(m->sub[0]->device) != hipSuccess) return VS_ERR_HIP;
    int rc = mixed_upload(m, nullptr, nullptr, nullptr, k_steps);
    if (rc) return rc;
    launch_rollout_mixed((const Segs*)m->dev, m->total_blocks, m->sub[0]->stream, m->sub[0]->auto_reset,
                         record ? m->sub[0]->record_mode : 0, k_steps, seed, mixed_redraws(m));
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
    launch_step_mixed((const Segs*)m->dev, m->total_blocks, m->sub[0]->stream, m->sub[0]->auto_reset, mixed_redraws(m));
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
        case VS_TRAJ_REC: *p = d.traj_rec; *bytes = (size_t)h->traj_cap * record_width(h->type, h->record_mode) * ld * 4; return true;
        case VS_TRAJ_DONE: *p = d.traj_done; *bytes = (size_t)((h->traj_cap + 31) / 32) * ld * 4; return true;
        case VS_FAILED: *p = d.failed; *bytes = ld; return true;
        case VS_EPSTAT_COUNT: *p = d.es_count; *bytes = ld * 4; return true;
        case VS_EPSTAT_RETSUM: *p = d.es_retsum; *bytes = ld * 4; return true;
        case VS_EPSTAT_LENSUM: *p = d.es_lensum; *bytes = ld * 4; return true;
        case VS_JAC_STATE: *p = d.jac_s; *bytes = d.jac_s ? (size_t)ei.S * (ei.S + ei.A) * ld * 4 : 0; return true;
        case VS_JAC_REW: *p = d.jac_r; *bytes = d.jac_r ? (size_t)(ei.S + ei.A) * ld * 4 : 0; return true;
        case VS_JAC_OBS: *p = d.jac_o; *bytes = d.jac_o ? (size_t)ei.O * (ei.S + ei.A) * ld * 4 : 0; return true;
#ifdef VS_WS_STAMP
        case 99: *p = d.dbg; *bytes = (size_t)(ld / 64) * 12 * 8; return true;
#endif

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
        DISPATCH_ENV(h->type, Launch<E>::observe(h));
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
    // recording launches rotate through the record buffer (as many k_steps-row slots as its capacity holds), so that a
    // capacity of several times the 256 MiB Infinity Cache makes the record stream a real HBM stream
    const int t0_saved = h->d.traj_t0;
    const int slots = (mode == 1 && record && k_steps > 0) ? (h->traj_cap / k_steps > 0 ? h->traj_cap / k_steps : 1) : 1;
    auto one = [&](int it) {
        if (mode == 0) return vs_step(h, actions, env_stride, dim_stride);
        if (record) h->d.traj_t0 = (it % slots) * k_steps;
        return vs_step_random(h, 1234, k_steps, record);
    };
    // events on the stream the kernels are launched on; one warm launch first
    rc = one(0);
    if (rc == VS_OK) {
        (void)hipEventRecord(e0, h->stream);
        for (int it = 0; it < iters && rc == VS_OK; ++it) rc = one(it + 1);
        (void)hipEventRecord(e1, h->stream);
        hipError_t e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) rc = fail(h, VS_ERR_HIP, "vs_time_step_kernel: event timing", e);
        *avg_ms = ms / (float)iters;
    }
    h->d.traj_t0 = t0_saved;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int vs_timer_start(vs_handle h) {
    if (!h) return VS_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->ev0) {
        HIPCHK(h, hipEventCreate(&h->ev0));
        HIPCHK(h, hipEventCreate(&h->ev1));
    }
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    return VS_OK;
}

int vs_timer_stop(vs_handle h, float* ms) {
    if (!h || !ms) return fail(h, VS_ERR_ARG, "vs_timer_stop: NULL argument");
    if (!h->ev0) return fail(h, VS_ERR_STATE, "vs_timer_stop: vs_timer_start has not been called");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return VS_OK;
}

// mode 0: copy (read + write of `bytes`), mode 1: pure write stream
static int mem_probe(int device_id, int64_t bytes, int iters, float* gbps, int mode) {
    if (!gbps || bytes < (1 << 20) || iters < 1) return VS_ERR_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return VS_ERR_HIP;
    void *a = nullptr, *b = nullptr;
    const size_t n4 = (size_t)bytes / 16;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return VS_ERR_HIP;
    if (hipMalloc(&a, n4 * 16) != hipSuccess) { (void)hipStreamDestroy(st); return VS_ERR_HIP; }
    if (mode == 0 && hipMalloc(&b, n4 * 16) != hipSuccess) { (void)hipFree(a); (void)hipStreamDestroy(st); return VS_ERR_HIP; }
    (void)hipMemsetAsync(a, 1, n4 * 16, st);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const dim3 g((unsigned)((n4 + 256 * PROBE_V - 1) / (256 * PROBE_V))), blk(256);
    auto one = [&](int i) {
        if (mode == 0) hipLaunchKernelGGL(k_copy4, g, blk, 0, st, (const v4f*)a, (v4f*)b, n4);
        else hipLaunchKernelGGL(k_fill4, g, blk, 0, st, (v4f*)a, n4, (float)i);
    };
    one(0);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) one(i + 1);
    (void)hipEventRecord(e1, st);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    if (b) (void)hipFree(b);
    (void)hipStreamDestroy(st);
    if (e != hipSuccess || ms <= 0.f) return VS_ERR_HIP;
    *gbps = (float)((mode == 0 ? 2.0 : 1.0) * (double)(n4 * 16) * iters / (ms * 1e-3) / 1e9);
    return VS_OK;
}

int vs_membw_probe(int device_id, int64_t bytes, int iters, float* gbps) { return mem_probe(device_id, bytes, iters, gbps, 0); }

int vs_memwrite_probe(int device_id, int64_t bytes, int iters, float* gbps) { return mem_probe(device_id, bytes, iters, gbps, 1); }

}  // extern "C"
