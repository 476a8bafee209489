/* TEST ORACLE (C part).  Not linked into, loaded by, or shipped with the product.
 *
 * lsap_f64: restatement of scipy.optimize.linear_sum_assignment (SciPy >= 1.6,
 * scipy/optimize/rectangular_lsap/rectangular_lsap.cpp; third-party dependency of
 * reference deep_sort/sort/linear_assignment.py:4,56; image has scipy 1.15.3).
 * Published algorithm: Crouse, "On implementing 2D rectangular assignment
 * algorithms", IEEE TAES 52(4), 2016 - shortest augmenting paths with dual
 * variables.  Tie-break details that matter for bit-exact assignments:
 *   - tall matrices (nr > nc) are solved transposed, result sorted by row;
 *   - `remaining` is filled in reverse (nc-1 .. 0) and shrunk by swap-with-last;
 *   - among equal minimum reduced costs the scan keeps the first minimum but
 *     moves to any later column of equal cost that is unassigned.
 * Pinned in tests/test_oracle_lsap.py against real scipy on tie-heavy matrices.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int64_t augmenting_path(int64_t nc, const double *cost, double *u, double *v,
                               int64_t *path, int64_t *row4col, double *spc, int64_t i,
                               char *SR, char *SC, int64_t *remaining, double *p_min)
{
    double minVal = 0;
    int64_t num_remaining = nc;
    for (int64_t it = 0; it < nc; it++) remaining[it] = nc - it - 1;
    memset(SC, 0, (size_t)nc);
    for (int64_t j = 0; j < nc; j++) spc[j] = INFINITY;
    int64_t sink = -1;
    while (sink == -1) {
        int64_t index = -1;
        double lowest = INFINITY;
        SR[i] = 1;
        for (int64_t it = 0; it < num_remaining; it++) {
            int64_t j = remaining[it];
            double r = minVal + cost[i * nc + j] - u[i] - v[j];
            if (r < spc[j]) { path[j] = i; spc[j] = r; }
            if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                lowest = spc[j];
                index = it;
            }
        }
        minVal = lowest;
        if (minVal == INFINITY) return -1;
        int64_t j = remaining[index];
        if (row4col[j] == -1) sink = j; else i = row4col[j];
        SC[j] = 1;
        remaining[index] = remaining[--num_remaining];
    }
    *p_min = minVal;
    return sink;
}

/* cost: nr x nc row-major doubles.  rows/cols: min(nr,nc) outputs.  returns 0 / -1 infeasible */
int lsap_f64(int64_t nr, int64_t nc, const double *cost_in, int64_t *rows, int64_t *cols)
{
    if (nr == 0 || nc == 0) return 0;
    int transpose = nc < nr;
    double *temp = NULL;
    const double *cost = cost_in;
    if (transpose) {
        temp = (double *)malloc(sizeof(double) * (size_t)(nr * nc));
        for (int64_t i = 0; i < nr; i++)
            for (int64_t j = 0; j < nc; j++) temp[j * nr + i] = cost_in[i * nc + j];
        int64_t t = nr; nr = nc; nc = t;
        cost = temp;
    }
    double *u = (double *)calloc((size_t)nr, sizeof(double));
    double *v = (double *)calloc((size_t)nc, sizeof(double));
    double *spc = (double *)malloc(sizeof(double) * (size_t)nc);
    int64_t *path = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int64_t *col4row = (int64_t *)malloc(sizeof(int64_t) * (size_t)nr);
    int64_t *row4col = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    int64_t *remaining = (int64_t *)malloc(sizeof(int64_t) * (size_t)nc);
    char *SR = (char *)malloc((size_t)nr), *SC = (char *)malloc((size_t)nc);
    for (int64_t j = 0; j < nc; j++) { path[j] = -1; row4col[j] = -1; }
    for (int64_t i = 0; i < nr; i++) col4row[i] = -1;
    int rc = 0;
    for (int64_t cur = 0; cur < nr; cur++) {
        double minVal;
        memset(SR, 0, (size_t)nr);
        int64_t sink = augmenting_path(nc, cost, u, v, path, row4col, spc, cur, SR, SC, remaining, &minVal);
        if (sink < 0) { rc = -1; break; }
        u[cur] += minVal;
        for (int64_t i = 0; i < nr; i++)
            if (SR[i] && i != cur) u[i] += minVal - spc[col4row[i]];
        for (int64_t j = 0; j < nc; j++)
            if (SC[j]) v[j] -= minVal - spc[j];
        int64_t j = sink;
        while (1) {
            int64_t i = path[j];
            row4col[j] = i;
            int64_t t = col4row[i]; col4row[i] = j; j = t;
            if (i == cur) break;
        }
    }
    if (rc == 0) {
        if (transpose) {
            /* rows of the original = col4row values; emit sorted by original row */
            int64_t k = 0;
            for (int64_t r = 0; r < nc; r++) {       /* nc (swapped) = original nr */
                int64_t who = row4col[r];
                if (who >= 0) { rows[k] = r; cols[k] = who; k++; }
            }
        } else {
            for (int64_t i = 0; i < nr; i++) { rows[i] = i; cols[i] = col4row[i]; }
        }
    }
    free(temp); free(u); free(v); free(spc); free(path); free(col4row); free(row4col);
    free(remaining); free(SR); free(SC);
    return rc;
}
