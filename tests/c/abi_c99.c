/* The boundary is a C ABI: this file is compiled as C99 (not C++) against include/pixell_hip.h and linked
 * with libpixell_hip.so.  It only takes the no-GPU paths (version, argument validation, error text). */
#include <stdio.h>
#include <string.h>
#include "pixell_hip.h"

int main(void) {
    char msg[256];
    pxl_car_wcs w = { { -1.0, 1.0 }, { 180.5, 91.0 }, { 0.5, 0.0 }, 0.017453292519943295 };
    int64_t shape[2] = { 360, 181 };
    if (pxl_version() != PXL_VERSION) return 1;
    if (sizeof(pxl_car_wcs) != 56) return 2;                                   /* 7 doubles, car_proj.jl:7-12 */
    if (pxl_pix2sky_car_f64(NULL, 0, NULL, NULL, PXL_WRAP_NONE, NULL) != PXL_EINVAL) return 3;
    pxl_last_error(msg, sizeof msg);
    if (!strstr(msg, "WCS")) return 4;
    if (pxl_pix2sky_car_f64(&w, 0, NULL, NULL, PXL_WRAP_UNWIND, NULL) != PXL_OK) return 5;   /* empty batch */
    if (pxl_sky2pix_car_f64(&w, shape, -1, NULL, NULL, 1, PXL_FORM_RECIP, NULL) != PXL_EINVAL) return 6;
    if (pxl_posmap_car_f64(&w, shape, 0, 0, NULL, NULL, 1, NULL) != PXL_OK) return 7;
    if (pxl_rewind_f64(NULL, 0, 6.283185307179586, 0.0, NULL) != PXL_OK) return 8;
    printf("abi_c99 ok (libpixell_hip %d)\n", pxl_version());
    return 0;
}
