/* A plain-C caller of the host-pointer entry point: proves that include/gpbo.h is C (not C++) and that the library
 * can be driven without Python.  Built and run by tests/test_c_abi.py.
 * Problem: N = 4 observations on a line, M = 9 candidates, LCB with explore = 4; prints the selected index, the best
 * acquisition value and mu / sigma of every candidate. */
#include <stdio.h>
#include <stdlib.h>

#include "gpbo.h"

int main(void) {
    const double X[4] = {0.0, 1.0, 2.5, 4.0};
    const double y[4] = {1.0, -0.5, 0.25, 2.0};
    const double ls[1] = {0.8};
    double Xs[9], mu[9], sigma[9], acq[9];
    gpbo_result res;
    int32_t info = -1;
    int i, rc;
    for (i = 0; i < 9; ++i) Xs[i] = 0.5 * i;
    if (gpbo_version() < 110) return 2;
    rc = gpbo_select_next_host_f64(X, y, 4, 1, ls, 1e-4, 1e-6, Xs, 9, GPBO_ACQ_LCB, 4.0, 0.0, 0.0, 0, mu, sigma, acq, NULL,
                                   &res, &info);
    if (rc != GPBO_OK) {
        fprintf(stderr, "gpbo_select_next_host_f64: %s\n", gpbo_strerror(rc));
        return 1;
    }
    printf("%d %lld %lld %.17g\n", (int)info, (long long)res.best_idx, (long long)res.nan_count, res.best_val);
    for (i = 0; i < 9; ++i) printf("%.17g %.17g %.17g\n", mu[i], sigma[i], acq[i]);
    return 0;
}
