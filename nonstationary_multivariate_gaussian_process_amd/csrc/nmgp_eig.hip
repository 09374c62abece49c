// Separable / stationary objectives, Kronecker-structured primitives and deterministic prediction.
// (entry points declared in include/nmgp.h)
#include "nmgp_internal.h"

using namespace nmgpk;

#define NMGP_TODO(ctx, name) return nmgp_fail(ctx, NMGP_E_UNSUPPORTED, name " is not implemented yet")

extern "C" int nmgp_logpos_sep(nmgp_ctx* c, const double*, const double[9], int, double[6], double*) {
    if (!c) return NMGP_E_NULL;
    NMGP_TODO(c, "nmgp_logpos_sep");
}
extern "C" int nmgp_logpos_sta(nmgp_ctx* c, const double*, const double[5], int, double[5], double*) {
    if (!c) return NMGP_E_NULL;
    NMGP_TODO(c, "nmgp_logpos_sta");
}
extern "C" int nmgp_kron_mv(nmgp_ctx* c, const double*, int, int, const double*, int, int, const double*, double*) {
    if (!c) return NMGP_E_NULL;
    NMGP_TODO(c, "nmgp_kron_mv");
}
extern "C" int nmgp_mvn_logpdf_kron(nmgp_ctx* c, const double*, const double*, const double*, int, const double*, int,
                                    double, double*) {
    if (!c) return NMGP_E_NULL;
    NMGP_TODO(c, "nmgp_mvn_logpdf_kron");
}
extern "C" int nmgp_mvn_logpdf_dense(nmgp_ctx* c, const double*, const double*, const double*, int, const double*, int,
                                     double, double*) {
    if (!c) return NMGP_E_NULL;
    NMGP_TODO(c, "nmgp_mvn_logpdf_dense");
}
extern "C" int nmgp_kron_inv_logdet(nmgp_ctx* c, double, const double*, int, const double*, int, double*, double*) {
    if (!c) return NMGP_E_NULL;
    NMGP_TODO(c, "nmgp_kron_inv_logdet");
}
extern "C" int nmgp_predict_svc(nmgp_ctx* c, const double*, const double[8], const double*, int, double*, double*,
                                double*) {
    if (!c) return NMGP_E_NULL;
    NMGP_TODO(c, "nmgp_predict_svc");
}
extern "C" int nmgp_predict_sep(nmgp_ctx* c, const double*, const double[9], const double*, int, double*, double*) {
    if (!c) return NMGP_E_NULL;
    NMGP_TODO(c, "nmgp_predict_sep");
}
extern "C" int nmgp_predict_sta(nmgp_ctx* c, const double*, const double*, int, double*, double*) {
    if (!c) return NMGP_E_NULL;
    NMGP_TODO(c, "nmgp_predict_sta");
}
extern "C" int nmgp_mvn_logpdf(nmgp_ctx* c, const double*, const double*, double, const double*, int, double*) {
    if (!c) return NMGP_E_NULL;
    NMGP_TODO(c, "nmgp_mvn_logpdf");
}
