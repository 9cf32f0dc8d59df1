/* A plain C host program against include/vecsim.h: what a non-Python caller (or a cgo / JNI shim) links.
 * Built and run by tests/test_capi_symbols.py; prints one line per check, exits non-zero on the first mismatch.
 * Without a GPU vs_create must fail loudly (no CPU fallback); with one it runs a short fused rollout. */
#include <stdio.h>
#include <string.h>

#include "vecsim.h"

#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main(void) {
    int S, A, O, P, H, I, K, F, nq, h2, h1;
    CHECK(vs_version() >= 200);
    CHECK(vs_env_dims(VS_ENV_QQ_SU, &S, &A, &O, &P, &H, &I, &K) == VS_OK);
    CHECK(S == 4 && A == 1 && O == 6 && P == 11 && H == 0 && I == 4);
    CHECK(strcmp(vs_env_name(VS_ENV_QQ_SU), "qq-su") == 0);
    CHECK(strcmp(vs_param_name(VS_ENV_QQ_SU, 0), "gravity_const") == 0);
    CHECK(vs_traj_layout(VS_ENV_QQ_SU, 1, &F, &nq, &h2, &h1) == VS_OK && F == 8 && nq == 2 && h2 == 0 && h1 == 0);
    CHECK(vs_traj_layout(VS_ENV_QBB, 1, &F, &nq, &h2, &h1) == VS_OK && F == 11 && nq == 2 && h2 == 1 && h1 == 1);
    CHECK(vs_traj_layout(VS_ENV_QCP_SU, 2, &F, &nq, &h2, &h1) == VS_OK && F == 13 && nq == 3 && h2 == 0 && h1 == 1);
    CHECK(vs_traj_layout(VS_ENV_QQ_SU, 3, &F, &nq, &h2, &h1) == VS_ERR_ARG);
    float nominal[32];
    CHECK(vs_nominal_params(VS_ENV_QQ_SU, 0, nominal) == VS_OK && nominal[0] > 9.8f && nominal[0] < 9.82f);
    printf("static tables ok\n");

    vs_handle h = NULL;
    int rc = vs_create(VS_ENV_QQ_SU, 4096, 0.004, 4000, 0, NULL, &h);
    if (rc != VS_OK) {
        CHECK(rc == VS_ERR_HIP && h == NULL);
        CHECK(strstr(vs_last_error(NULL), "no HIP device") != NULL);
        printf("no GPU: vs_create failed loudly: %s\n", vs_last_error(NULL));
        return 0;
    }
    CHECK(vs_n_envs(h) == 4096 && vs_ld(h) == 4096);
    CHECK(vs_set_auto_reset(h, 1, 7) == VS_OK);
    CHECK(vs_reset(h, NULL, 0, 0, NULL, 3) == VS_OK);
    CHECK(vs_step_random(h, 5, 100, 0) == VS_OK);
    CHECK(vs_sync(h) == VS_OK);
    static float state[4 * 4096];
    CHECK(vs_copy_to_host(h, VS_STATE, state) == VS_OK);
    CHECK(vs_error_count(h) == 0);
    double sum = 0.0;
    for (int i = 0; i < 4 * 4096; ++i) sum += state[i] < 0 ? -state[i] : state[i];
    CHECK(sum > 0.0);
    printf("GPU: 4096 envs x 100 fused steps ok, variant %d\n", vs_rollout_variant(h));
    CHECK(vs_destroy(h) == VS_OK);
    return 0;
}
