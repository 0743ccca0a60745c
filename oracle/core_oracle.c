/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the numba kernels of the reference's multigrid
 * inner loop, emg3d/core.py (reference @ /root/reference). Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the emg3d_amd product path never does.
 *
 * Parity status: PINNED. Checked in the build container against the reference
 * itself (imported un-jitted, tools/make_golden.py / tools/check_oracle_vs_reference.py)
 * and against the committed golden vectors in tests/golden/ (tests/test_oracle.py).
 *
 * Build: see oracle/Makefile  (gcc -O2 -std=c99, no -ffast-math).
 */
#include <complex.h>
#include <stdlib.h>
#include <stddef.h>

/* Sequence in which a FORWARD four-colour sweep visits the colour classes (backward =
 * reversed). Default 0,2,3,1 = the order of the HIP kernels (emg3d_amd/csrc/launch.h; chosen
 * because it converges fastest of the 4!/4 distinct sequences -- as does its mirror image
 * under x <-> y, 0,1,3,2 --); settable for experiments. */
int oracle_colour_order[4] = {0, 2, 3, 1};
void oracle_set_colour_order(int a, int b, int c, int d)
{
    oracle_colour_order[0] = a; oracle_colour_order[1] = b;
    oracle_colour_order[2] = c; oracle_colour_order[3] = d;
}

/* experiments only: a backward sequence that is not the reverse of the forward one */
int oracle_colour_order_b[4] = {1, 3, 2, 0};
int oracle_backward_custom = 0;
void oracle_set_colour_order_backward(int on, int a, int b, int c, int d)
{
    oracle_backward_custom = on;
    oracle_colour_order_b[0] = a; oracle_colour_order_b[1] = b;
    oracle_colour_order_b[2] = c; oracle_colour_order_b[3] = d;
}

/* orders 1 / 2 of the POINT smoother: the node colours in the same sequence in every sweep (default, = the
 * HIP kernels since round 3), or mirrored in backward sweeps (0) */
int oracle_point_repeat = 1;
void oracle_set_point_repeat(int on) { oracle_point_repeat = on; }

/* threads of the four-colour / tiled orders (classes of independent nodes, lines, tiles: results do not
 * depend on it); 1 = serial. The reference order (0) is sequential by definition and never threaded. */
int oracle_threads = 1;
void oracle_set_threads(int n) { oracle_threads = n > 1 ? n : 1; }

/* order 1 of the LINE smoothers: cyclic pass sequence (default, = the HIP kernels) or the mirrored sweeps */
int oracle_line_cyclic = 1;
int oracle_line_cycle[4] = {1, 2, 3, 0};
void oracle_set_line_order(int cyclic, int a, int b, int c, int d)
{
    oracle_line_cyclic = cyclic;
    oracle_line_cycle[0] = a; oracle_line_cycle[1] = b; oracle_line_cycle[2] = c; oracle_line_cycle[3] = d;
}

/* order 2 of the point smoother: tile extents in nodes and the sequence of the eight tile
 * colours of a FORWARD sweep (backward = reversed). */
int oracle_tile[3] = {32, 4, 6};
int oracle_tile_order[8] = {0, 7, 1, 6, 2, 5, 3, 4};   /* complementary colours next to each other (launch.h) */
/* experiment: backward sweeps visit the tile colours in the forward order too */
int oracle_tile_repeat = 0;
void oracle_set_tile_repeat(int on) { oracle_tile_repeat = on; }
void oracle_set_tile(int bx, int by, int bz) { oracle_tile[0] = bx; oracle_tile[1] = by; oracle_tile[2] = bz; }
void oracle_set_tile_order(const int *o) { int i; for (i = 0; i < 8; i++) oracle_tile_order[i] = o[i]; }

/* Over-relaxation INSIDE the colour passes / sweeps (an experiment of the oracle, not in the
 * reference, which notes the possibility at emg3d/core.py:774): every solved edge value is written as
 * old + omega (new - old). 1.0 = the reference's plain Gauss-Seidel update. */
double oracle_omega = 1.0;
void oracle_set_omega(double w) { oracle_omega = w; }

#define PASTE_(a, b) a##b
#define PASTE(a, b) PASTE_(a, b)

/* real instantiation:  *_d */
#define T double
#define FN(name) PASTE(name, _d)
#include "core_generic.h"
#undef T
#undef FN

/* complex instantiation:  *_z */
#define T double complex
#define FN(name) PASTE(name, _z)
#include "core_generic.h"
#undef T
#undef FN

/* core.restrict_weights -- reference emg3d/core.py:2004-2076.
 * n = len(cnodes); outputs wl, w0, wr of length n. d has n+1 entries. */
void restrict_weights(const double *nodes, const double *cell_centers, const double *h, int nh,
                      const double *cnodes, const double *ccell_centers, const double *ch, int n,
                      double *wl, double *w0, double *wr)
{
    double *d = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    int i;
    int nch = n - 1;           /* number of coarse cells */
    int nnodes = nh + 1;       /* number of fine nodes   */

    /* dual grid cell widths (core.py:2055-2059) */
    d[0] = h[0] / 2;
    d[n] = h[nh - 1] / 2;
    for (i = 1; i < n; i++) d[i] = (h[2 * i - 2] + h[2 * i - 1]) / 2.;

    /* left weight (core.py:2062-2065) */
    for (i = 0; i < n; i++) wl[i] = 1 / d[i];
    wl[0] *= (nodes[0] - h[0] / 2) - (cnodes[0] - ch[0] / 2);
    for (i = 1; i < n; i++) wl[i] *= cell_centers[2 * i - 1] - ccell_centers[i - 1];

    /* central weight (core.py:2068) */
    for (i = 0; i < n; i++) w0[i] = 1.0;

    /* right weight (core.py:2071-2074) */
    for (i = 0; i < n; i++) wr[i] = 1 / d[i + 1];
    wr[n - 1] *= (cnodes[n - 1] + ch[nch - 1] / 2) - (nodes[nnodes - 1] + h[nh - 1] / 2);
    for (i = 0; i < n - 1; i++) wr[i] *= ccell_centers[i] - cell_centers[2 * i];

    free(d);
}
