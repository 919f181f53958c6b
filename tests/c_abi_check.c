/* A plain-C client of libpysp_hip.so: the header must compile as C99 and the entry points must link.
 * Built and run by tests/test_abi_cpu.py::test_header_is_plain_c_and_links; needs no GPU (it checks that the
 * library says so loudly), on a GPU box it also creates a context and runs one tiny demosaic. */
#include <stdio.h>
#include <string.h>
#include "../include/pysp_hip.h"

int main(void) {
    static float dec[321 * 4], cb[257 * 4];
    if (pysp_abi_version() != PYSP_ABI_VERSION) { printf("abi mismatch\n"); return 1; }
    if (pysp_lab_tables(dec, cb) != PYSP_OK) { printf("lab tables: %s\n", pysp_last_error()); return 1; }
    if (!(dec[320 * 4] > 0.999f && dec[320 * 4] < 1.001f)) { printf("decode(1) = %g\n", dec[320 * 4]); return 1; }
    if (pysp_lab_tables(NULL, cb) != PYSP_EBADARG || !strstr(pysp_last_error(), "null")) { printf("bad-arg path\n"); return 1; }
    int n = pysp_device_count();
    pysp_ctx *ctx = pysp_ctx_create(0, NULL);
    if (n <= 0) {
        if (ctx != NULL || !strstr(pysp_last_error(), "no CPU fallback")) { printf("expected a loud failure without a GPU\n"); return 1; }
        printf("ok (no GPU: %s)\n", pysp_last_error());
        return 0;
    }
    if (!ctx) { printf("ctx: %s\n", pysp_last_error()); return 1; }
    float bayer[4 * 4], rgb[4 * 4 * 3];
    const float wb[3] = {2.0f, 1.0f, 1.4285715f};
    const double M[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 16; i++) bayer[i] = 0.25f;
    int rc = pysp_demosaic_f32(ctx, bayer, 4, 4, wb, M, PYSP_QUALITY_DRAFT, 0, 0, rgb);
    if (rc != PYSP_OK) { printf("demosaic: %s\n", pysp_last_error()); return 1; }
    if (pysp_demosaic_f32(ctx, bayer, 3, 4, wb, M, PYSP_QUALITY_DRAFT, 0, 0, rgb) != PYSP_EBADARG) { printf("odd dims accepted\n"); return 1; }
    if (pysp_demosaic_f32(ctx, bayer, 4, 4, wb, M, 7, 0, 0, rgb) != PYSP_ENOTIMPL) { printf("unknown quality accepted\n"); return 1; }
    pysp_ctx_destroy(ctx);
    printf("ok (GPU: rgb[0] = %g %g %g)\n", rgb[0], rgb[1], rgb[2]);
    return 0;
}
