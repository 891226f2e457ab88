/* pcb_eval.c -- plain C caller of the libpcx_hip C ABI (include/pcx.h): load a .pcb file
 * written by PyChebyshev (or by this package) straight into a device handle and evaluate it.
 * The MI355X counterpart of the reference's examples/binary_reader/reader.c, which walks the
 * same file on the CPU.
 *
 *   gcc -O2 -std=c99 -Iinclude examples/pcb_eval.c -o pcb_eval \
 *       -Lpychebyshev_amd -lpcx_hip -Wl,-rpath,$PWD/pychebyshev_amd
 *   ./pcb_eval tests/golden/approx_5d_bs.pcb 0.1 -0.2 0.3 0.4 -0.5
 *
 * Prints the value and, for every dimension, the first partial derivative at the point.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pcx.h"

static int die(const char *what) {
    fprintf(stderr, "%s: %s\n", what, pcx_last_error());
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s model.pcb x0 [x1 ...]\n", argv[0]);
        return 2;
    }
    int ndev = 0;
    if (pcx_device_count(&ndev) != 0 || ndev < 1) return die("no HIP device");
    pcx_bary *h = NULL;
    if (pcx_bary_create_from_pcb(0, argv[1], &h) != 0) return die("pcx_bary_create_from_pcb");
    int32_t d = 0, n_nodes[PCX_MAX_DIMS];
    if (pcx_bary_shape(h, &d, n_nodes) != 0) return die("pcx_bary_shape");
    if (argc - 2 != d) {
        fprintf(stderr, "model has %d dimensions, %d coordinates given\n", (int)d, argc - 2);
        pcx_bary_destroy(h);
        return 2;
    }
    double point[PCX_MAX_DIMS], out[1 + PCX_MAX_DIMS];
    int32_t specs[(1 + PCX_MAX_DIMS) * PCX_MAX_DIMS];
    memset(specs, 0, sizeof specs);
    for (int k = 0; k < d; ++k) {
        point[k] = atof(argv[2 + k]);
        specs[(1 + k) * d + k] = 1;                 /* spec 1+k: d/dx_k */
    }
    /* one launch for the value and the d first partials (vectorized_eval_multi) */
    if (pcx_bary_eval_multi_batch(h, point, 1, specs, 1 + d, out) != 0) return die("pcx_bary_eval_multi_batch");
    printf("value = %.17g\n", out[0]);
    for (int k = 0; k < d; ++k) printf("d/dx%d  = %.17g\n", k, out[1 + k]);
    pcx_bary_destroy(h);
    return 0;
}
