// LSMR(preconditioner!, P) with ANY preconditioner P that supports ldiv! (types.jl:82-86; README.md:47 of the reference: "The
// preconditioner can be any type that supports A_ldiv_B!(x, P, y)").  The fused device-resident recurrence of lsq_lsmr.hip
// folds a DIAGONAL P into its epilogues; a general P is a black box the library can only call, so this is the reference's own
// structure restated at the operator level: lsmr! (lsmr.jl:53-238) on PreconditionedMatrix(DampenedMatrix(J, sqrt(damp)), P)
// (iterative_lsmr.jl:12-51, 61-109, 179-198, 238-259), every vector operation a launch, the scalars on the host, P applied by
// the caller's callback between launches (the stream is drained first).  The slow path here as it is in the reference.
#include "lsq_solver.h"

#include <algorithm>
#include <cmath>

__global__ void __launch_bounds__(LSQ_NT) k_gen_sqrt(int n, double *__restrict__ d) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT) d[i] = sqrt(d[i]);   // map!(sqrt, damp, damp)
}
// b.x = z + alpha * x * y with b.x already scaled by beta (iterative_lsmr.jl:92, :107): out[i] = beta*out[i] + alpha * a[i] * dg[i]
__global__ void __launch_bounds__(LSQ_NT)
k_gen_muladd(int n, double beta, double *__restrict__ out, double alpha, const double *__restrict__ a, const double *__restrict__ dg) {
    for (int i = blockIdx.x * LSQ_NT + threadIdx.x; i < n; i += gridDim.x * LSQ_NT) {
        const double z = beta == 1.0 ? out[i] : (beta == 0.0 ? 0.0 : out[i] * beta);
        out[i] = z + alpha * a[i] * dg[i];
    }
}

static int gen_grid(const lsq_ctx *c, int n) { return std::max(1, std::min(lsq_div_up(n > 0 ? n : 1, LSQ_NT), c->num_cus * 4)); }

extern "C" int lsq_solver_set_general_preconditioner(lsq_solver *s, lsq_precond_update_callback update, lsq_precond_ldiv_callback ldiv,
                                                     void *user) {
    if (!s || s->kind != LSQ_LSMR) {
        lsq_set_error("a preconditioner belongs to LSMR()");
        return LSQ_EARG;
    }
    s->gen_update = update;
    s->gen_ldiv = ldiv;
    s->gen_user = user;
    return LSQ_OK;
}

int lsq_lsmr_general_solve(lsq_solver *s, lsq_mat *J, const double *d_y, double *d_damp, double *d_x, int *nmul) {
    lsq_ctx *c = s->ctx;
    const int m = J->m, n = J->n, gn = gen_grid(c, n);
    const bool damped = d_damp != nullptr;
    double *u = s->d_u, *ux = s->d_ux, *v = s->d_v, *h = s->d_h, *hbar = s->d_hbar, *tmp = s->d_t, *tmp2 = s->d_P, *x = s->d_rhs;
    const double *dg = d_damp;    // after map!(sqrt, damp, damp)
    auto P_ldiv = [&](double *out, const double *in) -> int {     // ldiv!(out, P, in)
        LSQ_HIP(hipStreamSynchronize(c->stream));
        if (s->gen_ldiv(out, in, s->gen_user) != 0) {
            lsq_set_error("preconditioner ldiv! callback reported failure");
            return LSQ_ECALLBACK;
        }
        return LSQ_OK;
    };
    auto norm_u = [&](double *out) -> int {                       // norm(::DampenedVector), iterative_lsmr.jl:72
        double a = 0.0, b = 0.0;
        LSQ_TRY(lsq_sumsq(c, m, u, &a));
        if (damped) LSQ_TRY(lsq_sumsq(c, n, ux, &b));
        const double ny = std::sqrt(a), nx = std::sqrt(b);
        *out = damped ? std::sqrt(ny * ny + nx * nx) : ny;
        return LSQ_OK;
    };
    auto scal_u = [&](double f) -> int {
        LSQ_TRY(lsq_scal(c, m, f, u));
        if (damped) LSQ_TRY(lsq_scal(c, n, f, ux));
        return LSQ_OK;
    };
    // mul!(v, A', u, 1, bv) for A = PreconditionedMatrix(DampenedMatrix(J, dg), P): iterative_lsmr.jl:36-51 over :95-109
    auto At_u = [&](double bv) -> int {
        LSQ_TRY(lsq_mul(J, 1, 1.0, u, 0.0, tmp));                 // fill!(tmp, 0); mul!(tmp, J', u.y, 1, 1)
        if (damped) LSQ_LAUNCH(k_gen_muladd, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, 1.0, tmp, 1.0, ux, dg);
        LSQ_TRY(P_ldiv(tmp2, tmp));
        if (bv == 0.0) LSQ_TRY(lsq_fill(c, n, 0.0, v));
        else if (bv != 1.0) LSQ_TRY(lsq_scal(c, n, bv, v));
        return lsq_axpy(c, n, 1.0, tmp2, v);
    };
    // mul!(u, A, v, 1, bu): ldiv!(tmp, P, v); rmul!(u, bu); u.y += J tmp; u.x += tmp .* dg   (:30-34 over :87-94)
    auto A_v = [&](double bu) -> int {
        LSQ_TRY(P_ldiv(tmp, v));
        LSQ_TRY(lsq_mul(J, 0, 1.0, tmp, bu, u));
        if (damped) LSQ_LAUNCH(k_gen_muladd, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, bu, ux, 1.0, (const double *)tmp, dg);
        LSQ_HIP(hipGetLastError());
        return LSQ_OK;
    };

    // ldiv! glue: iterative_lsmr.jl:179-198 / :238-259
    LSQ_TRY(lsq_fill(c, n, 0.0, x));
    LSQ_TRY(lsq_copy(c, m, d_y, u));
    if (damped) LSQ_TRY(lsq_fill(c, n, 0.0, ux));
    LSQ_TRY(lsq_fill(c, n, 0.0, tmp));
    LSQ_HIP(hipStreamSynchronize(c->stream));
    if (s->gen_update && s->gen_update(J, d_damp, s->gen_user) != 0) {      // preconditioner!(P, x, J, damp): damp un-rooted
        lsq_set_error("preconditioner! callback reported failure");
        return LSQ_ECALLBACK;
    }
    if (damped) LSQ_LAUNCH(k_gen_sqrt, dim3(gn), dim3(LSQ_NT), 0, c->stream, n, d_damp);
    const double atol = 1e-6, btol = damped ? 0.5 : 1e-6, ctol = 1e-8;
    const long long rows = (s->row_cb ? s->global_rows : (long long)m) + (damped ? n : 0);
    const int maxiter = (int)std::max<long long>(rows, n);

    // lsmr.jl:73-78 (x = 0: u = b)
    double beta, alpha;
    LSQ_TRY(norm_u(&beta));
    if (beta > 0) LSQ_TRY(scal_u(1.0 / beta));
    LSQ_TRY(At_u(0.0));
    LSQ_TRY(lsq_nrm2(c, n, v, &alpha));
    if (alpha > 0) LSQ_TRY(lsq_scal(c, n, 1.0 / alpha, v));
    double zetabar = alpha * beta, alphabar = alpha, rho = 1, rhobar = 1, cbar = 1, sbar = 0;
    LSQ_TRY(lsq_copy(c, n, v, h));
    LSQ_TRY(lsq_fill(c, n, 0.0, hbar));
    double betadd = beta, betad = 0, rhodold = 1, tautildeold = 0, thetatilde = 0, zeta = 0, d = 0;
    double normA2 = alpha * alpha, maxrbar = 0, minrbar = 1e100;
    const double normb = beta;
    double normAr = alpha * beta;
    int istop = 0, iter = 0;
    if (normAr != 0) {
        while (iter < maxiter) {
            iter += 1;
            LSQ_TRY(A_v(-alpha));
            LSQ_TRY(norm_u(&beta));
            if (beta > 0) {
                LSQ_TRY(scal_u(1.0 / beta));
                LSQ_TRY(At_u(-beta));
                LSQ_TRY(lsq_nrm2(c, n, v, &alpha));
                if (alpha > 0) LSQ_TRY(lsq_scal(c, n, 1.0 / alpha, v));
            }
            const double alphahat = alphabar, chat = 1.0, shat = 0.0;     // lambda = 0 (:127-130)
            const double rhoold = rho;
            rho = std::sqrt(alphahat * alphahat + beta * beta);
            const double cc = alphahat / rho, ss = beta / rho;
            const double thetanew = ss * alpha;
            alphabar = cc * alpha;
            const double rhobarold = rhobar, zetaold = zeta;
            const double thetabar = sbar * rho, rhotemp = cbar * rho;
            rhobar = std::sqrt((cbar * rho) * (cbar * rho) + thetanew * thetanew);
            cbar = cbar * rho / rhobar;
            sbar = thetanew / rhobar;
            zeta = cbar * zetabar;
            zetabar = -sbar * zetabar;
            LSQ_TRY(lsq_scal(c, n, -thetabar * rho / (rhoold * rhobarold), hbar));     // :152-156
            LSQ_TRY(lsq_axpy(c, n, 1.0, h, hbar));
            LSQ_TRY(lsq_axpy(c, n, zeta / (rho * rhobar), hbar, x));
            LSQ_TRY(lsq_scal(c, n, -thetanew / rho, h));
            LSQ_TRY(lsq_axpy(c, n, 1.0, v, h));
            const double betaacute = chat * betadd, betacheck = -shat * betadd;    // :164-184
            const double betahat = cc * betaacute;
            betadd = -ss * betaacute;
            const double thetatildeold = thetatilde;
            const double rhotildeold = std::sqrt(rhodold * rhodold + thetabar * thetabar);
            const double ctildeold = rhodold / rhotildeold, stildeold = thetabar / rhotildeold;
            thetatilde = stildeold * rhobar;
            rhodold = ctildeold * rhobar;
            betad = -stildeold * betad + ctildeold * betahat;
            tautildeold = (zetaold - thetatildeold * tautildeold) / rhotildeold;
            const double taud = (zeta - thetatilde * tautildeold) / rhodold;
            d = d + betacheck * betacheck;
            const double normr = std::sqrt(d + (betad - taud) * (betad - taud) + betadd * betadd);
            normA2 = normA2 + beta * beta;                                          // :187-189
            const double normA = std::sqrt(normA2);
            normA2 = normA2 + alpha * alpha;
            maxrbar = std::max(maxrbar, rhobarold);                                 // :192-196
            if (iter > 1) minrbar = std::min(minrbar, rhobarold);
            const double condA = std::max(maxrbar, rhotemp) / std::min(minrbar, rhotemp);
            normAr = std::fabs(zetabar);                                            // :205-206
            double normx;
            LSQ_TRY(lsq_nrm2(c, n, x, &normx));
            const double test1 = normr / normb, test2 = normAr / (normA * normr), test3 = 1.0 / condA;
            const double t1 = test1 / (1.0 + normA * normx / normb);
            const double rtol = btol + atol * normA * normx / normb;
            if (iter >= maxiter) { istop = 7; break; }                              // :224-231, first hit wins
            if (1.0 + test3 <= 1.0) { istop = 6; break; }
            if (1.0 + test2 <= 1.0) { istop = 5; break; }
            if (1.0 + t1 <= 1.0) { istop = 4; break; }
            if (test3 <= ctol) { istop = 3; break; }
            if (test2 <= atol) { istop = 2; break; }
            if (test1 <= rtol) { istop = 1; break; }
        }
    }
    LSQ_TRY(P_ldiv(tmp, x));                      // ldiv!(tmp, P, x); copyto!(x, tmp)  (iterative_lsmr.jl:195-196, 256-257)
    LSQ_TRY(lsq_copy(c, n, tmp, d_x));
    s->last_iter = iter;
    s->last_istop = istop;
    if (nmul) *nmul = 2 * iter;                   // lsmr.jl:236
    return LSQ_OK;
}
