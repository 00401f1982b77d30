/* computeFlow from plain C through the C ABI of libb2f.so (include/b2f.h) -- the harness a maintainer
 * without Lua or Python would write; mirrors README.md:45-59 of the reference (init, computeFlow, save).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/compute_flow.c -o compute_flow \
 *       -Lback2future_amd -lb2f -Wl,-rpath,$PWD/back2future_amd -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64
 *   ./compute_flow <model: name | path.t7 | random:hard|soft[:seed[:gain]]> <in.raw> <H> <W> <out.raw>
 *
 * in.raw : three 3 x H x W planar RGB float32 images in [0,1] (im1, im2, im3), little endian
 * out.raw: flow 2 x H x W float64, then fwd_occ H x W bytes, then bwd_occ H x W bytes
 */
#include <stdio.h>
#include <stdlib.h>

#include "b2f.h"

static int die(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, b2f_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc != 6) {
        fprintf(stderr, "usage: %s <model> <in.raw> <H> <W> <out.raw>\n", argv[0]);
        return 2;
    }
    const int H = atoi(argv[3]), W = atoi(argv[4]);
    if (H <= 0 || W <= 0) return 2;
    const size_t hw = (size_t)H * (size_t)W;
    float *im = (float *)malloc(9 * hw * sizeof(float));
    double *flow = (double *)malloc(2 * hw * sizeof(double));
    unsigned char *occ = (unsigned char *)malloc(2 * hw);
    if (!im || !flow || !occ) return 3;
    FILE *f = fopen(argv[2], "rb");
    if (!f || fread(im, sizeof(float), 9 * hw, f) != 9 * hw) {
        fprintf(stderr, "cannot read %s\n", argv[2]);
        return 3;
    }
    fclose(f);

    b2f_ctx *ctx = NULL;
    if (b2f_init(argv[1], 0, &ctx)) return die("b2f_init");
    int levels, win, past_flow, n_outputs;
    long long n_params;
    if (b2f_info(ctx, &levels, &win, &past_flow, &n_outputs, &n_params)) return die("b2f_info");
    fprintf(stderr, "model %s: %d levels, window %d, %s, %lld parameters\n", argv[1], levels, win,
            past_flow ? "Soft (past-flow decoders)" : "Hard", n_params);
    if (b2f_compute_flow(ctx, im, im + 3 * hw, im + 6 * hw, H, W, flow, occ, occ + hw)) return die("b2f_compute_flow");
    b2f_destroy(ctx);

    f = fopen(argv[5], "wb");
    if (!f || fwrite(flow, sizeof(double), 2 * hw, f) != 2 * hw || fwrite(occ, 1, 2 * hw, f) != 2 * hw) {
        fprintf(stderr, "cannot write %s\n", argv[5]);
        return 3;
    }
    fclose(f);
    free(im); free(flow); free(occ);
    return 0;
}
