// host_dev_shim.cpp -- TEST ONLY.  Compiles pais_mvs_amd/csrc/pais_dev.hpp (the
// lane-local arithmetic that is inlined into the gfx950 kernels) for the host so
// that it can be checked against the oracle without a GPU.  Not part of the
// product; nothing in the product links this.
#include <math.h>
#include <string.h>
#include "../pais_mvs_amd/csrc/pais_dev.hpp"

extern "C" {
double shim_exp(double x) { return pais::det_exp(x); }
double shim_exp_bf(double x) { return pais::det_exp_bf(x); }
// number of inputs where the branch-free statement differs in any bit from det_exp (NaN payloads compared as bits)
long shim_exp_bf_mismatches(const double *x, long n)
{
    long bad = 0;
    for (long i = 0; i < n; ++i) bad += pais::d2u(pais::det_exp_bf(x[i])) != pais::d2u(pais::det_exp(x[i]));
    return bad;
}
double shim_exp_poly(double x) { return pais::det_exp_poly(x); }
// max |ulp| error of det_exp_poly against the platform exp over an array (inputs where exp is finite and normal)
double shim_exp_poly_max_ulp(const double *x, long n)
{
    double worst = 0;
    for (long i = 0; i < n; ++i) {
        const double a = pais::det_exp_poly(x[i]), b = exp(x[i]);
        if (!(b > 1e-300) || !(b < 1e300)) continue;
        const double u = fabs(a - b) / (nextafter(b, INFINITY) - b);
        if (u > worst) worst = u;
    }
    return worst;
}
double shim_sin(double x) { return pais::det_sin(x); }
double shim_cos(double x) { return pais::det_cos(x); }
unsigned shim_rand31(unsigned long long seed, unsigned long long key, unsigned run, unsigned k) { return pais::rand31(seed, key, run, k); }
unsigned long long shim_child_key(unsigned long long pk, int cam, int cx, int cy) { return pais::child_key(pk, cam, cx, cy); }
double shim_region_ratio(double ptx, double pty, int r, const double *H) { return pais::region_ratio(ptx, pty, r, H); }
void shim_inv3(const double *m, double *o) { pais::inv3(m, o); }
void shim_plane_h(double d, double s, const double *KRr, const double *KTr, const double *KR, const double *KT, const double *n, double *H)
{
    double Mr[9], inv[9], M[9];
    pais::plane_matrix(d, s, KRr, KTr, n, Mr);
    pais::inv3(Mr, inv);
    pais::plane_matrix(d, s, KR, KT, n, M);
    pais::mul33(M, inv, H);
}
void shim_project(const double *R, const double *T, const double *f, const double *pp, double sc, const double *X, double *out)
{
    pais::project_raw(R, T, f, pp, sc, X, out);
}
double shim_bilinear(const unsigned char *img, int stride, double ix, double iy) { return pais::bilinear(img, stride, ix, iy); }
void shim_s2n(double t, double p, double *n) { pais::spherical2normal(t, p, n); }
// one moveParticles() pass over the whole swarm; draws start at index k0 (4 per particle)
void shim_pso_move_all(int N, int localK, double iw, unsigned long long seed, unsigned long long key, unsigned run, unsigned k0,
                       double *pos, double *vec, const double *pBest, double *nBest, const double *fit,
                       const double *pBestFit, int gIdx, const double *rangeL, const double *rangeU)
{
    unsigned long long sb = pais::stream_base(seed, key);
    double gB[3] = {pBest[gIdx * 3], pBest[gIdx * 3 + 1], pBest[gIdx * 3 + 2]};
    for (int i = 0; i < N; ++i) {
        double u[4];
        for (int q = 0; q < 4; ++q) u[q] = pais::uniform_from(sb, run, k0 + 4 * i + q);
        pais::pso_move_particle(i, N, localK, iw, u, (double(*)[3])pos, (double(*)[3])vec, (const double(*)[3])pBest,
                                (double(*)[3])nBest, fit, pBestFit, gB, rangeL, rangeU);
    }
}
}
