// ref_pso_harness.cpp -- TEST INFRASTRUCTURE.  Drives the reference's own
// PsoSolver (compiled from /root/reference/TMVS/pso/*.cpp, see Makefile) with
// an injected deterministic uniform stream, so that oracle/pais_oracle.c's
// restatement po_pso_run() can be pinned against the real thing.
//
// rand()/srand()/time() below shadow libc inside this shared object only
// (-Wl,-Bsymbolic): PsoSolver::random() (psosolver.cpp:66-68) then returns
// r_k / RAND_MAX with r_k the injected 31-bit draws, and setRandomSeed()
// (psosolver.cpp:60-64) becomes a no-op.  OpenMP is forced to one thread so the
// draw order is the serial order of SURVEY 8a-P2.
#include <omp.h>
#include <stdint.h>
#include <time.h>
#include "pso/psosolver.h"

typedef uint32_t (*rnd_fn)(void *);
typedef double (*fit_fn)(const double *pos, void *obj);

static rnd_fn g_rnd = 0;
static void *g_rndObj = 0;

extern "C" int rand(void) throw() { return g_rnd ? (int)g_rnd(g_rndObj) : 0; }
extern "C" void srand(unsigned int) throw() {}
extern "C" time_t time(time_t *t) throw() { if (t) *t = 0; return 0; }

struct Adapter { fit_fn fn; void *obj; };
static double adapter(const PAIS::Particle &p, void *obj)
{
    Adapter *a = (Adapter *)obj;
    return a->fn(p.pos, a->obj);
}

extern "C" int ref_pso_rand_max(void) { return RAND_MAX; }

// Same call sequence as Patch::psoOptimization (patch.cpp:190-213):
// new PsoSolver(dim, L, U, fn, obj, maxIt, N); setParticle(init); run(true).
extern "C" int ref_pso_run(int dim, const double *L, const double *U, fit_fn fn, void *obj,
                           int maxIteration, int particleNum, const double *init,
                           rnd_fn rnd, void *rndObj, int enableGLN,
                           double *gBest, double *gBestFitness, int *iterations)
{
    omp_set_num_threads(1);
    g_rnd = rnd;
    g_rndObj = rndObj;
    Adapter a = {fn, obj};
    PAIS::PsoSolver *solver = new PAIS::PsoSolver(dim, L, U, adapter, &a, maxIteration, particleNum);
    if (init) solver->setParticle(init);
    solver->run(enableGLN != 0);
    const double *g = solver->getGbest();
    for (int d = 0; d < dim; ++d) gBest[d] = g[d];
    *gBestFitness = solver->getGbestFitness();
    *iterations = solver->getIteration();
    delete solver;
    g_rnd = 0;
    g_rndObj = 0;
    return 0;
}
