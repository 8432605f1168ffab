/* Plain-C client of the drop-in boundary (include/aprilgrid_amd.h): what a Rust / C / Go binding
 * does, without Python or torch in the process.  Build:
 *   gcc -std=c99 -Iinclude examples/c_client.c -Laprilgrid-rs_amd -laprilgrid_amd \
 *       -Wl,-rpath,$PWD/aprilgrid-rs_amd -o c_client
 * Usage: c_client <raw L8 file> <width> <height> [n_frames]   (prints saddle and tag counts of the first frame; with
 *        n_frames > 1 the file holds that many frames back to back and every frame also goes through agx_detect_batch)
 * Mirrors: TagDetector::new(&TagFamily::T36H11, None) -> refined_saddle_points / detect. */
#include "aprilgrid_amd.h"

#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv)
{
    if (argc != 4 && argc != 5) {
        fprintf(stderr, "usage: %s <raw L8 file> <width> <height> [n_frames]\n", argv[0]);
        return 2;
    }
    const int w = atoi(argv[2]), h = atoi(argv[3]), n_frames = argc == 5 ? atoi(argv[4]) : 1;
    if (w <= 0 || h <= 0 || n_frames <= 0) return 2;
    unsigned char *img = (unsigned char *)malloc((size_t)w * h * (size_t)n_frames);
    FILE *f = fopen(argv[1], "rb");
    if (!img || !f || fread(img, 1, (size_t)w * h * (size_t)n_frames, f) != (size_t)w * h * (size_t)n_frames) {
        fprintf(stderr, "cannot read %d x %dx%d bytes from %s\n", n_frames, w, h, argv[1]);
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
    if (n_frames > 1) {
        /* detect over the whole file: chain chunk by chunk on the device, board search + decode on every CPU the process is granted */
        enum { CAP = 128 };
        agx_tag *all = (agx_tag *)malloc(sizeof(agx_tag) * CAP * (size_t)n_frames);
        uint32_t *counts = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n_frames);
        int *status = (int *)malloc(sizeof(int) * (size_t)n_frames);
        if (!all || !counts || !status) return 1;
        st = agx_detect_batch(det, img, NULL, n_frames, w, h, (size_t)w, (size_t)w * h, AGX_L8, all, CAP, counts, status, 0);
        if (st != AGX_OK) {
            fprintf(stderr, "agx_detect_batch: %s (%s)\n", agx_status_string(st), agx_last_error(det));
            return 1;
        }
        printf("batch of %d frames on %d host threads:", n_frames, agx_host_parallelism());
        for (int i = 0; i < n_frames; ++i) printf(" %u", counts[i]);
        printf(" tags; frame 0's first tag id %u\n", counts[0] ? all[0].id : 0u);
        if (counts[0] != n_tags) {
            fprintf(stderr, "agx_detect_batch and agx_detect disagree on frame 0: %u vs %u tags\n", counts[0], n_tags);
            return 1;
        }
        free(all);
        free(counts);
        free(status);
    }
    agx_detector_destroy(det);
    free(img);
    return 0;
}
