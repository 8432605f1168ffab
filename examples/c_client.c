/* Plain-C client of the drop-in boundary (include/aprilgrid_amd.h): what a Rust / C / Go binding
 * does, without Python or torch in the process.  Build:
 *   gcc -std=c99 -Iinclude examples/c_client.c -Laprilgrid-rs_amd -laprilgrid_amd \
 *       -Wl,-rpath,$PWD/aprilgrid-rs_amd -o c_client
 * Usage: c_client <raw L8 file> <width> <height>   (prints saddle and tag counts)
 * Mirrors: TagDetector::new(&TagFamily::T36H11, None) -> refined_saddle_points / detect. */
#include "aprilgrid_amd.h"

#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
    if (argc != 4) {
        fprintf(stderr, "usage: %s <raw L8 file> <width> <height>\n", argv[0]);
        return 2;
    }
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    unsigned char *img = (unsigned char *)malloc((size_t)w * h);
    FILE *f = fopen(argv[1], "rb");
    if (!img || !f || fread(img, 1, (size_t)w * h, f) != (size_t)w * h) {
        fprintf(stderr, "cannot read %dx%d bytes from %s\n", w, h, argv[1]);
        return 2;
    }
    fclose(f);

    int family = 0;
    if (agx_family_from_str("t36h11", &family) != AGX_OK) return 1;
    agx_params params;
    agx_default_params(&params);
    agx_detector *det = NULL;
    int st = agx_detector_create(family, &params, 0, &det);
    if (st != AGX_OK) {
        fprintf(stderr, "agx_detector_create: %s (%s)\n", agx_status_string(st), agx_last_error(NULL));
        return 1;
    }
    static agx_saddle saddles[16384];
    uint32_t n_saddles = 0;
    st = agx_refined_saddle_points(det, img, w, h, (size_t)w, AGX_L8, saddles, 16384, &n_saddles);
    if (st != AGX_OK) {
        fprintf(stderr, "agx_refined_saddle_points: %s (%s)\n", agx_status_string(st), agx_last_error(det));
        return 1;
    }
    static agx_tag tags[1024];
    uint32_t n_tags = 0;
    st = agx_detect(det, img, w, h, (size_t)w, AGX_L8, tags, 1024, &n_tags);
    if (st != AGX_OK) {
        fprintf(stderr, "agx_detect: %s (%s)\n", agx_status_string(st), agx_last_error(det));
        return 1;
    }
    printf("abi %d: %u saddles, %u tags", agx_abi_version(), n_saddles, n_tags);
    if (n_saddles) printf("; first saddle (%.3f, %.3f) k=%.5f", saddles[0].x, saddles[0].y, saddles[0].k);
    if (n_tags) printf("; first tag id %u", tags[0].id);
    printf("\n");
    agx_detector_destroy(det);
    free(img);
    return 0;
}
