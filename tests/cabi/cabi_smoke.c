/* Plain-C client of include/nbody_amd.h: proves the drop-in boundary needs nothing but the C-ABI.
 * Built and run by tests/test_gpu_parity.py::test_c_client_of_the_cabi on the GPU box:
 *     gcc -O2 -I include tests/cabi/cabi_smoke.c -o cabi_smoke -L <pkg> -lnbody_amd -Wl,-rpath,<pkg> -lm
 * Prints:  n  kinetic  potential  sum(x)  sum(v)   after `steps` leapfrog steps in FLOAT64 mode. */
#include <stdio.h>
#include <stdlib.h>

#include "nbody_amd.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        int rc_ = (call);                                                      \
        if (rc_ != NB_OK) {                                                    \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nb_last_error());    \
            return 1;                                                          \
        }                                                                      \
    } while (0)

static double lcg(unsigned long long *s)
{
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)((*s >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 53);
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 3000;
    const int steps = argc > 2 ? atoi(argv[2]) : 5;
    unsigned long long seed = 12345;
    double *pos = malloc(sizeof(double) * 2 * n), *vel = malloc(sizeof(double) * 2 * n), *mass = malloc(sizeof(double) * n);
    for (int i = 0; i < n; ++i) {
        pos[2 * i] = 20.0 * lcg(&seed) - 10.0;
        pos[2 * i + 1] = 20.0 * lcg(&seed) - 10.0;
        vel[2 * i] = 0.2 * lcg(&seed) - 0.1;
        vel[2 * i + 1] = 0.2 * lcg(&seed) - 0.1;
        mass[i] = 0.5 + lcg(&seed);
    }
    nb_config cfg = {0};
    cfg.n = n; cfg.dim = 2; cfg.mode = NB_FLOAT64; cfg.levels = 0;
    cfg.G = 0.001; cfg.softening_sq = 0.1 * 0.1; cfg.dt = 0.01;
    cfg.device = 0; cfg.rank = 0; cfg.nranks = 1; cfg.flags = 0;
    nb_sim *sim = NULL;
    CHECK(nb_create(&sim, &cfg));
    CHECK(nb_set_state(sim, pos, vel, mass, NB_F64, 0));
    CHECK(nb_compute_accelerations(sim));
    CHECK(nb_step(sim, steps));
    double ke, pe;
    CHECK(nb_energy(sim, &ke, &pe));
    int32_t dts[4];
    CHECK(nb_state_dtypes(sim, dts));
    if (dts[0] != NB_F64 || dts[3] != NB_F64) { fprintf(stderr, "unexpected dtypes\n"); return 1; }
    CHECK(nb_get_state(sim, pos, vel, NULL, NULL, 0));
    double sx = 0, sv = 0;
    for (int i = 0; i < 2 * n; ++i) { sx += pos[i]; sv += vel[i]; }
    printf("%d %.17g %.17g %.17g %.17g\n", n, ke, pe, sx, sv);
    CHECK(nb_destroy(sim));
    free(pos); free(vel); free(mass);
    return 0;
}
